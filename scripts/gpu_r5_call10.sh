#!/bin/bash
# Round 5, tenth GPU call: persistent kNN-merge grid (CBGX_KNN_MERGE_GRID: 4096 default / 1000000 = the old one-workgroup-per-four-nodes
# grid) and a re-sweep of batches in flight on the round's final kernels; parity tests of the cached graph first
TAG=${1:-r05j}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sized.py -q -m gpu -x -k "static or cache or diffbp or sampl or rollout" -p no:faulthandler 2>&1 | grep -v "^$" | tail -4 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$1', d['value'], {n: v for n, v in k.items() if v[1]})"; }
for rep in 1 2; do for g in 4096 2048 1000000; do CBGX_KNN_MERGE_GRID=$g timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | line "knn_grid=$g headline"; done; done | tee $OUT/ab_fwd_$TAG.log
for cfg in "340 3" "250 4" "250 3" "500 2" "340 2" "200 5"; do set -- $cfg
timeout 300 python bench.py --steps 6 --warmup 2 --graphs-per-batch $1 --streams $2 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graphs_per_batch=$1 streams=$2', d['value'])"; done | tee $OUT/sweep_$TAG.log
timeout 120 python bench.py --model diffbp --pockets 20 --samples 10 --graphs-per-batch 200 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | line "diffbp 200 graphs" | tee -a $OUT/ab_fwd_$TAG.log
