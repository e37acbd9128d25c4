#!/bin/bash
# last check of a round: every GPU test, smoke, the driver's line without the CPU leg
TAG=${1:-r04y}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -q -m gpu -p no:faulthandler 2>&1 | grep -v "^$" | tail -6 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke_$TAG.log
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$TAG.json; wc -c $OUT/bench_$TAG.json; python - <<PY
import json
d = json.load(open("$OUT/bench_$TAG.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["config"].get("devices_by_uuid"))
print({k: v.get("value") if isinstance(v, dict) else v for k, v in d["secondary"].items()})
PY
