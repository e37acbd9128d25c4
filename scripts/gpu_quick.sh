#!/bin/bash
# quick A/B call: GPU tests + the driver bench line without the CPU leg.
# Usage: bash scripts/gpu_quick.sh <tag> [pytest -k expr | "all"]
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
if [ "$2" = "all" ]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" | tail -15 | tee $OUT/pytest_gpu_$TAG.log
else
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sized.py -q -x -k "${2:-not training}" 2>&1 | grep -v "^$" | tail -12 | tee $OUT/pytest_quick_$TAG.log
fi
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_quick_$TAG.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', d['value'], 'ms/denoise-step', d['config']['ms_per_denoising_step_of_the_job'], 'x2h us', r['avg_launch_us'], 'frac', r['frac'])
print({k: round(v['us_avg'],1) for k,v in {n: {'us_avg': v[0], 'launches': v[1]} for n, v in r['per_kernel_us_avg_and_launches'].items()}.items() if v['launches']})"
