#!/bin/bash
# round-2 call b: whole GPU suite without -x, gradient-error diagnostic at the configs[4] shape, pipe micro-benchmarks
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== ubench pipes =="
timeout 300 scripts/ubench/pipes 2>&1 | tee $OUT/ubench_pipes_$TAG.log
echo "== pytest -m gpu (all) =="
timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | grep -v "^$" | tail -40 | tee $OUT/pytest_gpu_$TAG.log
echo "== gradient error diagnostic =="
timeout 900 python scripts/grad_err_config_sized.py 32 404 2>&1 | tail -60 | tee $OUT/grad_err_$TAG.log
timeout 600 python scripts/grad_err_config_sized.py 8 404 2>&1 | tail -60 | tee $OUT/grad_err_b8_$TAG.log
