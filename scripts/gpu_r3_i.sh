#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bench.py -q -m gpu -x 2>&1 | tail -6 | cut -c1-400
timeout 600 python bench.py > $OUT/bench_r03y.json 2> $OUT/bench_r03y.err; tail -3 $OUT/bench_r03y.err | cut -c1-300
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/bench_r03y.json") if l.startswith("{")][0])
print(d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
for k, v in d["secondary"].items():
    if isinstance(v, dict): print(k, v.get("value"), v.get("error"), v.get("measured_in_s"))
    else: print(k, v)
P
