#!/bin/bash
TAG=${1:-r03f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -q -m gpu -x --durations=3 2>&1 | grep -v "^$" | tail -12
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_quick_$TAG.json | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print(o['value'], o['ms_per_step'], r['frac'], r['avg_launch_us'], o['config']['streams'])
for k,v in o['secondary'].items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('error'))"
