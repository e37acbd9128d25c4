"""bench.py as the driver runs it: the JSON contract, and the N-rank path on the hardware at hand (two ranks sharing
the one GPU of a gpurun box over gloo -- RCCL needs one GPU per rank; the 8-GPU run is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e,
                       timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_small_job():
    out = _run(["--steps", "2", "--warmup", "1", "--pockets", "4", "--graphs-per-batch", "20", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["dtype"] == "f32"
    cfg = out["config"]
    assert cfg["graphs_per_gpu"] == 40 and cfg["denoising_steps_per_bench_step"] == 5 and cfg["ranks_seen"] == 1
    # value is consistent with ms_per_step: graphs x 5 denoising steps per bench step
    assert abs(out["value"] - 40 * 5 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1.5 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4


def test_bench_two_ranks_on_one_gpu_gloo():
    """`python bench.py --gpus 2` launches two ranks itself; both take part (ranks_seen from an all-reduce), the value is
    the whole-job aggregate.  gloo because both ranks share this box's single GPU."""
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pockets", "2", "--graphs-per-batch", "20",
                "--no-cpu-baseline", "--no-roofline"], env={"CBGX_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == 2
    assert abs(out["value"] - 2 * 20 * 5 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]


def test_bench_train_two_ranks_on_one_gpu_gloo():
    out = _run(["--gpus", "2", "--workload", "train", "--steps", "2", "--warmup", "1", "--pockets", "4",
                "--no-cpu-baseline", "--no-roofline"], env={"CBGX_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == 2
    assert out["config"]["allreduce_ms_per_step"] > 0
