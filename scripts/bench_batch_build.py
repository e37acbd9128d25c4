"""Batch construction cost: per-sample Python collate (what sample.py:177-183 + the transform chain
do, modelled by cbgbench_amd/synthetic.make_batch) vs the vectorised builder of cbgbench_amd/priors.py.
    python scripts/bench_batch_build.py [--device cuda]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbgbench_amd import priors, synthetic  # noqa: E402


def timeit(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args()
    out = {}
    for P, S in ((10, 10), (100, 10)):
        rng = np.random.default_rng(0)
        pk = [synthetic.make_pocket(rng, int(rng.integers(350, 651))) for _ in range(P)]
        n_lig = rng.integers(10, 46, size=(P, S))

        def loop():
            b = synthetic.make_batch([pk[p] for p in range(P) for _ in range(S)], n_lig.reshape(-1), rng, 13)
            return synthetic.batch_to(b, args.device)

        ps = priors.PocketSet(pk, device=args.device, center=False)

        def vec():
            return priors.build_sampling_batch(ps, S, 13, n_lig=n_lig, rng=rng)

        out[f"{P}x{S}"] = {"per_sample_collate_ms": round(timeit(loop), 3), "vectorised_ms": round(timeit(vec), 3),
                          "pocket_set_once_ms": round(timeit(lambda: priors.PocketSet(pk, device=args.device, center=False), 2), 3)}
    print(json.dumps({"device": args.device, "batch_build": out}))


if __name__ == "__main__":
    main()
