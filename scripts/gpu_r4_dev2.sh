#!/bin/bash
# Round-4 development call: GPU tests, then the stage microbenchmark over every ab_libs/*.so (two interleaved repetitions).
TAG=${1:-r04b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
if [ "$2" != "notest" ]; then
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 --durations=5 -p no:faulthandler 2>&1 | grep -v "^$" | tail -60 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
fi
echo "== stage microbenchmark =="
for rep in 1 2; do for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 120 python scripts/ubench_stage.py 2>&1 | tail -1; done; done | tee $OUT/ubench_stage_$TAG.log
