#!/bin/bash
# fused MFMA gate backward + static rounds / dynamic remainder in the x2h edge backward: parity tests on the new build, then the
# training line on ab_libs/base.so (HEAD) against ab_libs/new.so
TAG=${1:-r04p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_loss.py tests/test_gpu_training.py -m gpu -q 2>&1 | tail -40 | tee $OUT/pytest_train_$TAG.log
timeout 600 python -m pytest tests/test_gpu_config_sized.py -m gpu -q -k "test_training_gradients_at_config5_shape or test_diffbp_training" 2>&1 | tail -5 | tee -a $OUT/pytest_train_$TAG.log
bash scripts/ab_train.sh 2>&1 | tee $OUT/ab_train_$TAG.log
CBGX_LIBRARY=$(pwd)/ab_libs/new.so timeout 300 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_train_$TAG.json
