#!/bin/bash
# node_proj occupancy variant (four workgroups per CU, no register prefetch) against the default: stage microbenchmark + forward A/B
TAG=${1:-r04l}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 10 x2h 2>&1 | tail -1; CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 10 2>&1 | tail -1; done; done | tee $OUT/ubench_stage_$TAG.log
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
