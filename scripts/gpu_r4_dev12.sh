#!/bin/bash
# node_proj with two column chunks per workgroup (-DCBGX_NPROJ_CPW=2) against the default: stage microbenchmark, forward A/B and the
# forward parity tests on the variant
TAG=${1:-r04m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 10 x2h 2>&1 | tail -1; done; done | tee $OUT/ubench_stage_$TAG.log
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
CBGX_LIBRARY=$(pwd)/ab_libs/cpw2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/parity_cpw2_$TAG.log
