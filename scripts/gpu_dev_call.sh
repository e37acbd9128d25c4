#!/bin/bash
# One development call on the GPU box (the parametrised form of the per-call scripts of rounds 4 and 5): pytest selection, smoke, then an
# interleaved A/B of every ab_libs/*.so (built by scripts/build_variant.py; `base` = the commit before) on one of the bench lines.
# Usage (repo root on the GPU box):
#   bash scripts/gpu_dev_call.sh <tag> <tests> <ab> [extra]
#     tests : all | train | fwd | none | a pytest -k expression          (all = the driver's command: python -m pytest tests/ -x -q -m gpu)
#     ab    : train | fwd | small | none                                  (scripts/ab_train.sh, scripts/ab_fwd.sh, the 1 / 10-graph rows)
#     extra : a command run at the end (its output goes to gpurun_out/extra_<tag>.log)
# CBGX_TEST_LIBS="a b": the gradient tests once more per named variant (CBGX_LIBRARY=ab_libs/<name>.so) -- parity of an experimental build.
TAG=${1:-dev}; TESTS=${2:-all}; AB=${3:-none}; EXTRA=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
PYT="timeout 1500 python -m pytest -x -q -m gpu -p no:faulthandler"
filt() { grep -v "^$" | tail -${1:-8} | cut -c1-300; }
case "$TESTS" in
  none) ;;
  all)   echo "== pytest: all GPU tests =="; $PYT tests/ --durations=4 2>&1 | filt 12 | tee $OUT/pytest_gpu_$TAG.log ;;
  train) echo "== pytest: gradient parity =="; $PYT tests/test_gpu_training.py tests/test_gpu_train_loss.py tests/test_gpu_config_sized.py 2>&1 | filt | tee $OUT/pytest_gpu_$TAG.log ;;
  fwd)   echo "== pytest: forward parity =="; $PYT tests/test_gpu_parity.py tests/test_gpu_config_sized.py tests/test_gpu_range.py 2>&1 | filt | tee $OUT/pytest_gpu_$TAG.log ;;
  *)     echo "== pytest -k '$TESTS' =="; $PYT tests/ -k "$TESTS" 2>&1 | filt | tee $OUT/pytest_gpu_$TAG.log ;;
esac
for v in $CBGX_TEST_LIBS; do
  echo "== gradient + forward stage parity of ab_libs/$v.so =="
  CBGX_LIBRARY=$ROOT/ab_libs/$v.so $PYT tests/test_gpu_training.py tests/test_gpu_parity.py -k "training or stages or full_denoiser or backward or grad" 2>&1 | filt 6 | tee $OUT/pytest_${v}_$TAG.log
done
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke_$TAG.log
case "$AB" in
  train) echo "== A/B training line =="; bash scripts/ab_train.sh 2>&1 | tee $OUT/ab_train_$TAG.log ;;
  fwd)   echo "== A/B sampling line =="; bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log ;;
  small) echo "== small inputs =="; bash scripts/gpu_small_batch.sh $TAG 2>&1 | tail -20 | tee $OUT/small_$TAG.log ;;
esac
if [ -n "$EXTRA" ]; then echo "== extra: $EXTRA =="; bash -c "$EXTRA" 2>&1 | tail -40 | tee $OUT/extra_$TAG.log; fi
