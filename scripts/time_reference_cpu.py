#!/usr/bin/env python
"""Time the UNMODIFIED reference (repo/models/diffusion/targetdiff.py::TargetDiff.sample, imported from /root/reference behind
the import shims of oracle/ref_shim.py) on this host's CPU cores, next to the oracle port on the same inputs.

Only runs where /root/reference exists (the build container; not the GPU box).  BASELINE.md section 2 records the output:
    python scripts/time_reference_cpu.py > profiles/reference_cpu_r02.json

A T-step model is built so that `model.sample(batch)` is exactly T iterations of the reference's own loop
(targetdiff.py:150-182: embed, compose, denoiser, posterior sampling, per-step .cpu()).  Same synthetic pockets
(cbgbench_amd/synthetic.py), same synthetic weights (oracle/weights.py) as bench.py's GPU run and cpu_baseline leg.
Caveat printed with the numbers: torch_scatter / torch_cluster are the pure-torch stand-ins of oracle/ref_shim.py (the C++
wheels are not installed here); they are < 3 % of a step."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cbgbench_amd import synthetic  # noqa: E402
from oracle import ref_shim, targetdiff as OT, weights as W  # noqa: E402


def reference_model(T):
    M = ref_shim.load_reference()
    cfg = ref_shim.targetdiff_config(13, 9)
    cfg.generator.num_diffusion_timesteps = T
    model = M.get_model(cfg).eval()
    model.load_state_dict(W.synthetic_state_dict(13, 9, seed=0, num_timesteps=T), strict=True)
    return model


def time_reference(batch, T, threads):
    torch.set_num_threads(threads)
    model = reference_model(T)
    torch.manual_seed(2024)                       # sample.py:106
    with torch.no_grad():
        t0 = time.perf_counter()
        model.sample(batch)
        return (time.perf_counter() - t0) / T


def time_oracle(batch, T, threads):
    torch.set_num_threads(threads)
    sd = W.synthetic_state_dict(13, 9, seed=0, num_timesteps=1000)
    x = batch["ligand_pos"]
    c = torch.nn.functional.one_hot(batch["ligand_atom_type"], 13).float()
    g = torch.Generator().manual_seed(2024)
    n = x.shape[0]
    with torch.no_grad():
        t0 = time.perf_counter()
        for s in range(T):
            x, c = OT.denoise_step(sd, batch, x, c, 999 - s, torch.randn(n, 3, generator=g), torch.rand(n, 13, generator=g), 13)
        return (time.perf_counter() - t0) / T


def main():
    ncpu = os.cpu_count() or 1
    rng = np.random.default_rng(1000)
    pocket = synthetic.make_pocket(rng, 450)
    cases = {
        "configs[0]: 1 pocket x 1 sample, N = 450 + 25": (synthetic.make_batch([pocket], [25], rng, 13), 1, 10),
        "configs[1] batch of sample.py:177-183: 1 pocket x 10 samples, N = 10 x (450 + ~27)":
            (synthetic.make_batch([pocket] * 10, [int(rng.integers(10, 46)) for _ in range(10)], rng, 13), 10, 3),
    }
    out = {"host_cores": ncpu, "torch": torch.__version__,
           "caveat": "torch_scatter / torch_cluster = pure-torch stand-ins (oracle/ref_shim.py); reference otherwise unmodified",
           "rows": []}
    for name, (batch, graphs, T) in cases.items():
        for threads in sorted({min(8, ncpu), 1}, reverse=True):
            if threads == 1 and graphs > 1:
                continue
            time_reference(batch, 2, threads)      # warm-up (T >= 2: the reference indexes posterior_var[1]); (allocator, oneDNN primitives)
            ref_s = time_reference(batch, T, threads)
            ora_s = time_oracle(batch, T, threads)
            out["rows"].append({"case": name, "threads": threads, "steps_timed": T,
                                "reference_ms_per_step": round(1e3 * ref_s, 1),
                                "reference_graph_steps_per_s": round(graphs / ref_s, 3),
                                "oracle_port_ms_per_step": round(1e3 * ora_s, 1),
                                "oracle_port_graph_steps_per_s": round(graphs / ora_s, 3)})
            print(json.dumps(out["rows"][-1]), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
