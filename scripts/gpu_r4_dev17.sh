#!/bin/bash
# classifier-head backward on the ligand rows + wider loss kernel: parity tests on the default build, then the training line on
# ab_libs/a_head.so against ab_libs/b_cls.so
TAG=${1:-r04x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_loss.py tests/test_gpu_training.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|^E |FAILED" | cut -c1-500 | head -20 | tee $OUT/pytest_train_$TAG.log
timeout 600 python -m pytest tests/test_gpu_config_sized.py -m gpu -q -k "training_gradients_at_config5_shape" 2>&1 | grep -E "passed|failed|Error|^E |FAILED" | cut -c1-500 | head -10 | tee -a $OUT/pytest_train_$TAG.log
bash scripts/ab_train.sh 2>&1 | tee $OUT/ab_train_$TAG.log
