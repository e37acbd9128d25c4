#!/bin/bash
# node_proj timing ablations (libcbgx_ablate-style variants in ab_libs/, WRONG results by design): which resource bounds the kernel
TAG=${1:-r04h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 10 x2h 2>&1 | tail -1; done; done | tee $OUT/ubench_nproj_$TAG.log
