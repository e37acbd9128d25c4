#!/bin/bash
TAG=${1:-r02m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" | tail -12 | tee $OUT/pytest_gpu_$TAG.log
for p in "1 10" "1 1" "4 10"; do set -- $p
timeout 300 python bench.py --pockets $1 --samples $2 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_p$1s$2_$TAG.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('pockets $1 samples $2 value', d['value'], 'ms/denoise-step', d['config']['ms_per_denoising_step_of_the_job'])
print({k: round(v['us_avg'],1) for k,v in r['per_kernel'].items() if v['launches']}, r['launches_per_denoising_step'])"
done
timeout 400 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('train value', d['value'], 'ms/step', d['ms_per_step'])
print({k: (round(v['us_avg'],1), v['launches']) for k,v in r['per_kernel'].items() if v['launches']})"
