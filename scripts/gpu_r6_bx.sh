#!/bin/bash
# Round 6 development call for the x2h edge backward: gradient parity of the tree's library, then A/B of every ab_libs/*.so on the
# training line (scripts/ab_train.sh: each twice, interleaved).  Usage (repo root on the GPU box): bash scripts/gpu_r6_bx.sh [tag]
TAG=${1:-r06b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== gradient parity (tree library) =="
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_loss.py tests/test_gpu_config_sized.py -x -q -m gpu -p no:faulthandler 2>&1 | grep -v "^$" | tail -15 | cut -c1-300 | tee $OUT/pytest_train_$TAG.log
echo "== new parity tests of round 6 (inference mode, 16 oracle graphs at config size) =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:faulthandler -k "inference_mode or config2 or linker_256_graphs_runs" 2>&1 | grep -v "^$" | tail -6 | cut -c1-300 | tee $OUT/pytest_new_$TAG.log
echo "== gradient parity of ab_libs/bx2.so (4x4x1 tile) =="
CBGX_LIBRARY=$ROOT/ab_libs/bx2.so timeout 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -p no:faulthandler 2>&1 | grep -v "^$" | tail -6 | cut -c1-300 | tee $OUT/pytest_train_bx2_$TAG.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke_$TAG.log
echo "== A/B training line =="
bash scripts/ab_train.sh 2>&1 | tee $OUT/ab_train_$TAG.log
