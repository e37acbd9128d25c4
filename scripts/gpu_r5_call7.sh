#!/bin/bash
# Round 5, seventh GPU call: the training backward with the weight-gradient work of every block on the auxiliary stream
# (CBGX_TRAIN_OVERLAP, default on): every gradient test, then the training line with / without it, twice, interleaved
TAG=${1:-r05g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu (training files) =="
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_loss.py tests/test_train_cli.py -q -m gpu -x -p no:faulthandler 2>&1 | grep -v "^$" | tail -6 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
timeout 900 python -m pytest tests/test_gpu_config_sized.py -q -m gpu -x -k "train or grad" -p no:faulthandler 2>&1 | grep -v "^$" | tail -4 | cut -c1-300 | tee -a $OUT/pytest_gpu_$TAG.log
echo "== training line A/B =="
for rep in 1 2; do for ov in 1 0; do CBGX_TRAIN_OVERLAP=$ov timeout 200 python bench.py --workload train --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('overlap=$ov train', d['value'], 'graph-steps/s; ms/step', d['ms_per_step'], '; x2h backward', d['roofline']['avg_launch_us'], 'us')"; done; done | tee $OUT/ab_train_$TAG.log
for m in diffbp diffsbdd; do for ov in 1 0; do CBGX_TRAIN_OVERLAP=$ov timeout 200 python bench.py --workload train --model $m --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('overlap=$ov train $m', d['value'], 'graph-steps/s')"; done; done | tee -a $OUT/ab_train_$TAG.log
