#!/bin/bash
# DiffBP with the denoiser's h' on the CoM stack's source rows only: parity + config-sized tests, the sampler rows
TAG=${1:-r04k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sized.py tests/test_gpu_bench.py -q -m gpu --maxfail=12 -p no:faulthandler 2>&1 | grep -v "^$" | tail -12 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
for m in diffbp diffsbdd targetdiff; do timeout 300 python bench.py --model $m --pockets 20 --graphs-per-batch 200 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m', d['value'], d['roofline']['per_kernel_us_avg_and_launches'])"; done | tee $OUT/samplers_$TAG.log
