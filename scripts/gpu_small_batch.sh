#!/bin/bash
# small-batch rows of the sampling bench: 1 / 10 / 40 graphs per step (profile mode prints the per-kernel averages)
TAG=${1:-sb}
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "1 1" "1 10" "4 10"; do
  set -- $cfg
  python bench.py --pockets $1 --samples $2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/bench_p${1}s${2}_$TAG.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('pockets $1 samples $2:', d['value'], d['unit'], 'ms/denoise-step', d['config']['ms_per_denoising_step_of_the_job'])
print('   ', {k: round(v['us_avg'],1) for k,v in {n: {'us_avg': v[0], 'launches': v[1]} for n, v in r['per_kernel_us_avg_and_launches'].items()}.items() if v['launches']})"
done
