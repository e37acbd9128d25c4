#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 300 python scripts/dev_dbg.py 2>&1 | tail -12 | cut -c1-300
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tee $OUT/pytest_gpu_r03y.log | tail -8 | cut -c1-250
