#!/bin/bash
# The driver's GPU test command, N times back to back on ONE box: python -m pytest tests/ -x -q -m gpu
# Usage (repo root on the GPU box): bash scripts/gpu_suite3x.sh [tag] [times]
TAG=${1:-r06a}; N=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
: > $OUT/pytest_gpu_3x_$TAG.log
for i in $(seq 1 $N); do
  echo "== run $i of $N: python -m pytest tests/ -x -q -m gpu ==" | tee -a $OUT/pytest_gpu_3x_$TAG.log
  timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=4 -p no:faulthandler 2>&1 | grep -v "^$" | tail -12 | cut -c1-300 | tee -a $OUT/pytest_gpu_3x_$TAG.log
done
grep -c " passed" $OUT/pytest_gpu_3x_$TAG.log
