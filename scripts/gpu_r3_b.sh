#!/bin/bash
# Round-3 call B: the tests added this round + the full default driver line (secondary block, CPU legs).
TAG=${1:-r03b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest (new tests) =="
timeout 1500 python -m pytest tests/test_gpu_range.py tests/test_gpu_bench.py tests/test_gpu_config_sized.py -q -m gpu --durations=8 2>&1 | grep -v "^$" | tail -40 | tee $OUT/pytest_gpu_$TAG.log
echo "== bench: default driver command =="
( time timeout 1200 python bench.py ) 2>&1 | tail -5 | tee $OUT/bench_$TAG.json | cut -c1-6000
