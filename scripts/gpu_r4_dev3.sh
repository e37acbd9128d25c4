#!/bin/bash
# Round-4 development call: GPU tests (stop at 12 failures), stage microbenchmark and the forward A/B over ab_libs/*.so
TAG=${1:-r04c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 -p no:faulthandler 2>&1 | grep -v "^$" | tail -70 | cut -c1-400 | tee $OUT/pytest_gpu_$TAG.log
echo "== stage microbenchmark =="
for rep in 1 2; do for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 2>&1 | tail -1; done; done | tee $OUT/ubench_stage_$TAG.log
echo "== A/B forward =="
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
