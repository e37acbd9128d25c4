#!/bin/bash
# One gpurun call of round 2: GPU parity tests, smoke, the driver's bench line (whole configs[1] job), the N-rank path on
# the one GPU at hand (2 ranks over gloo; an RCCL attempt is recorded too), training bench, small-batch rows, rocprofv3.
# Usage (from repo root on the GPU box): bash scripts/gpu_round2.sh [tag]
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== host =="; rocminfo | grep -m2 -E "gfx|Compute Unit"; nproc; free -g | sed -n 2p
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 -s 2>&1 | grep -v "^$" | tail -40 | tee $OUT/pytest_gpu_$TAG.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $OUT/smoke_$TAG.log
echo "== bench (driver line: whole configs[1] job) =="
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -2 | tee $OUT/bench_$TAG.json
echo "== bench --gpus 2 launched by bench.py itself, both ranks on this GPU (gloo) =="
CBGX_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --pockets 20 --no-cpu-baseline --no-roofline 2>&1 | tail -2 | tee $OUT/bench_2rank_gloo_$TAG.json
echo "== same over RCCL (expected to be refused: two ranks on one device) =="
timeout 180 python bench.py --gpus 2 --steps 2 --warmup 1 --pockets 20 --no-cpu-baseline --no-roofline 2>&1 | tail -4 | cut -c1-300 | tee $OUT/bench_2rank_rccl_$TAG.log
echo "== training bench (configs[4] shape) =="
timeout 400 python bench.py --workload train --steps 10 --warmup 3 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json
CBGX_DIST_BACKEND=gloo timeout 400 python bench.py --workload train --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | tee $OUT/bench_train_2rank_gloo_$TAG.json
echo "== small batches: 1 pocket x 10 samples (sample.py's own batch), 1 x 1 =="
timeout 300 python bench.py --pockets 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_p1_$TAG.json
timeout 300 python bench.py --pockets 1 --samples 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_p1s1_$TAG.json
echo "== linker (configs[2]) =="
timeout 300 python bench.py --workload linker --graphs-per-batch 256 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_linker_$TAG.json
echo "== rocprofv3 kernel trace of the driver line's command (fewer steps) =="
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1 ; tail -1 $OUT/rocprof_$TAG.log | cut -c1-300 )
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160 && cp "$f" $OUT/kernel_stats_$TAG.csv
find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
find $OUT/prof_$TAG -name "*.db" -delete 2>/dev/null
du -sh $OUT | tail -1
