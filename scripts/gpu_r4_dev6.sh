#!/bin/bash
# Round-4 development call: all GPU tests, the driver's line with its secondary block (no CPU leg), forward A/B over ab_libs/*.so
TAG=${1:-r04f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1700 python -m pytest tests -q -m gpu --maxfail=12 -p no:faulthandler -s 2>&1 | grep -v "^$" | grep -E "passed|failed|Error|error|assert|roll-out|ReLU flip|worst relative|FAILED|pocket frame|^\| " | tail -60 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
echo "== bench (driver line, no CPU leg) =="
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$TAG.json; wc -c $OUT/bench_$TAG.json; python - <<PY
import json
d = json.load(open("$OUT/bench_$TAG.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["per_kernel_us_avg_and_launches"])
for k, v in d["secondary"].items():
    print(k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a in ("value", "ms_per_step", "frac", "error", "wall_s")})
PY
echo "== A/B forward =="
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
