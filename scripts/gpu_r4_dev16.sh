#!/bin/bash
# node_proj with rows two tiles ahead (-DCBGX_NPROJ_DEPTH=2) against the default: stage microbenchmark, forward A/B, forward parity
TAG=${1:-r04v}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 10 x2h 2>&1 | tail -1; done; done | tee $OUT/ubench_stage_$TAG.log
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
CBGX_LIBRARY=$(pwd)/ab_libs/deep.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | tee $OUT/parity_deep_$TAG.log
