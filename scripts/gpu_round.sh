#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench (two batch shapes), rocprofv3 kernel trace.
# Usage (from repo root on the GPU box): bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
echo "== rocminfo =="; rocminfo | grep -m3 -E "gfx|Compute Unit" ; nproc
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 | tee $OUT/pytest_gpu_$TAG.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke_$TAG.log
echo "== bench default (20 pockets x 10 samples) =="
timeout 600 python bench.py 2>&1 | tail -3 | tee $OUT/bench_$TAG.json
echo "== bench 10 pockets x 10 samples (the batch most earlier rows were measured on) =="
timeout 300 python bench.py --pockets 10 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_p10_$TAG.json
echo "== bench 1 pocket x 10 samples =="
timeout 300 python bench.py --pockets 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_p1_$TAG.json
echo "== bench 1 pocket x 1 sample (config 1 shape) =="
timeout 300 python bench.py --pockets 1 --samples 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_p1s1_$TAG.json
echo "== rocprofv3 kernel trace =="
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python ${GRAFT_REPO_ROOT:-.}/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1 ; tail -2 $OUT/rocprof_$TAG.log )
find $OUT/prof_$TAG -name "*kernel_stats*" | head -3
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" && cp "$f" $OUT/kernel_stats_$TAG.csv
# keep the merged-back payload small
find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
echo "== training bench (configs[4] shape) =="
timeout 300 python bench.py --workload train --steps 10 --warmup 3 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json
