#!/bin/bash
# training-step launch diet (merged slab folds / reduce-and-store, no per-block fills, fused noising + losses): parity tests, then the
# training line with and without the fused tensor-side kernels
TAG=${1:-r04o}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_loss.py tests/test_gpu_training.py -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest_train_$TAG.log
timeout 600 python -m pytest tests/test_gpu_config_sized.py -m gpu -q -k "training_gradients_at_config5_shape" 2>&1 | tail -5 | tee -a $OUT/pytest_train_$TAG.log
for rep in 1 2; do for f in 1 0; do
CBGX_FUSED_TRAINING_OPS=$f timeout 300 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fused=$f train', d['value'], 'graph-steps/s', d['ms_per_step'], 'ms; x2h backward', d['roofline']['avg_launch_us'], 'us')"
done; done | tee $OUT/ab_train_$TAG.log
