#!/bin/bash
# h2x edge backward with a dynamic remainder; one / two / three dynamic rounds in the x2h edge backward: block-level parity on the
# default build, then the training line on every ab_libs/*.so
TAG=${1:-r04q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_loss.py tests/test_gpu_training.py -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_train_$TAG.log
bash scripts/ab_train.sh 2>&1 | tee $OUT/ab_train_$TAG.log
