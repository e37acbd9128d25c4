#!/bin/bash
# Round-3 correctness + quick numbers in one gpurun call: GPU tests, smoke, sampling line (no CPU leg), training line.
# Usage (repo root on the GPU box): bash scripts/gpu_r3_check.sh [tag] [pytest -k filter]
TAG=${1:-r03a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu =="
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout 1500 python -m pytest tests -q -m gpu --durations=5 "${K[@]}" 2>&1 | grep -v "^$" | tail -40 | tee $OUT/pytest_gpu_$TAG.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke_$TAG.log
echo "== bench (no CPU leg) =="
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_quick_$TAG.json | cut -c1-1500
echo "== training =="
timeout 600 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json | cut -c1-1200
