#!/bin/bash
TAG=${1:-r03g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=3 2>&1 | grep -v "^$" | tail -8
timeout 600 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('train', o['value'], o['ms_per_step'])"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_quick_$TAG.json | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print(o['value'], o['ms_per_step'], r['frac'], r['avg_launch_us'])
for k,v in o['secondary'].items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('launches_per_denoising_step'), v.get('error'))"
