#!/bin/bash
# Final measurement call of round 2 (one gpurun): GPU parity tests, smoke, driver bench line with the CPU leg, the 2-rank path,
# model rows (diffbp / diffsbdd / linker), training, small batches, rocprofv3 kernel stats (sampling + training) and the PMC
# passes of the dominant forward kernel (instruction mix, HBM traffic) and of the x2h backward.
# Usage (from repo root on the GPU box): bash scripts/gpu_round2_final.sh [tag]
TAG=${1:-r02z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== host =="; rocminfo | grep -m2 -E "gfx|Compute Unit"; nproc; free -g | sed -n 2p
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -q -m gpu --durations=6 2>&1 | grep -v "^$" | tail -14 | tee $OUT/pytest_gpu_$TAG.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke_$TAG.log
echo "== bench (driver line: whole configs[1] job, CPU leg included) =="
timeout 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench_$TAG.json | cut -c1-400
echo "== 2 ranks on this GPU (gloo): sampling, training =="
CBGX_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --pockets 20 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | tee $OUT/bench_2rank_gloo_$TAG.json | cut -c1-300
CBGX_DIST_BACKEND=gloo timeout 400 python bench.py --workload train --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | tee $OUT/bench_train_2rank_gloo_$TAG.json | cut -c1-300
echo "== model rows =="
for m in diffbp diffsbdd; do timeout 600 python bench.py --model $m 2>&1 | tail -1 | tee $OUT/bench_${m}_$TAG.json | cut -c1-300; done
timeout 300 python bench.py --workload linker --graphs-per-batch 256 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_linker_$TAG.json | cut -c1-300
echo "== training (configs[4] shape) =="
timeout 600 python bench.py --workload train --steps 10 --warmup 3 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json | cut -c1-400
echo "== small batches =="
bash scripts/gpu_small_batch.sh $TAG
echo "== rocprofv3 kernel stats: driver line command (fewer steps), training =="
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1 ; tail -1 $OUT/rocprof_$TAG.log | cut -c1-200 )
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160 && cp "$f" $OUT/kernel_stats_$TAG.csv
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$TAG -o tr -- python $ROOT/bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/rocprof_train_$TAG.log 2>&1 ; tail -1 $OUT/rocprof_train_$TAG.log | cut -c1-200 )
f=$(find $OUT/prof_train_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160 && cp "$f" $OUT/kernel_stats_train_$TAG.csv
find $OUT/prof_$TAG $OUT/prof_train_$TAG -name "*kernel_trace.csv" -delete; find $OUT/prof_$TAG $OUT/prof_train_$TAG -name "*.db" -delete 2>/dev/null
echo "== PMC: forward x2h (instruction mix, traffic), x2h backward =="
bash scripts/gpu_pmc_x2h.sh $TAG 2>&1 | tail -30
bash scripts/gpu_pmc_traffic.sh $TAG 2>&1 | tail -12
bash scripts/gpu_pmc_bwd.sh $TAG 2>&1 | grep -A24 "edge_backward_x2h"
du -sh $OUT | tail -1
