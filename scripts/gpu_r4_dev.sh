#!/bin/bash
# Round-4 development call (one gpurun): all GPU tests on the in-tree library, the forward A/B over ab_libs/*.so (two interleaved
# repetitions), and the per-launch kernel trace of one denoising step WITHOUT the two-stream overlap (CBGX_OVERLAP=0, one caller
# stream), so that every launch's duration is its own.  Usage (repo root on the GPU box): bash scripts/gpu_r4_dev.sh [tag] [notest]
TAG=${1:-r04a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
if [ "$2" != "notest" ]; then
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 --durations=5 2>&1 | grep -v "^$" | tail -40 | tee $OUT/pytest_gpu_$TAG.log
fi
echo "== A/B forward =="
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
echo "== trace, no overlap =="
CBGX_OVERLAP=0 bash scripts/gpu_trace_sizes.sh ${TAG}_serial 2>&1 | tail -45 | cut -c1-400
