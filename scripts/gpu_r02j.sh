#!/bin/bash
TAG=${1:-r02j}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" | tail -8 | tee $OUT/pytest_gpu_$TAG.log
for m in diffbp diffsbdd; do
  timeout 600 python bench.py --model $m --steps 6 --warmup 2 2>&1 | tail -1 | tee $OUT/bench_${m}_$TAG.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$m value', d['value'], 'ms/denoise-step', d['config']['ms_per_denoising_step_of_the_job'], 'x2h us', r['avg_launch_us'], 'frac', r['frac'], 'cpu', d['cpu_baseline']['value'])
print({k: round(v['us_avg'],1) for k,v in r['per_kernel'].items() if v['launches']}, r['launches_per_denoising_step'])"
done
