#!/bin/bash
# Round-4 development call: parity tests, stage microbenchmark and forward A/B of ab_libs/*.so
TAG=${1:-r04i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest (parity, range, config-sized sampling) =="
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_range.py tests/test_gpu_training.py -q -m gpu --maxfail=12 -p no:faulthandler 2>&1 | grep -v "^$" | tail -12 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
echo "== stage microbenchmark =="
for rep in 1 2; do for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 10 2>&1 | tail -1; CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 10 x2h 2>&1 | tail -1; done; done | tee $OUT/ubench_stage_$TAG.log
echo "== A/B forward =="
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
