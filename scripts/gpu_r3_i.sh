#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_range.py -q -m gpu -x 2>&1 | tail -3 | cut -c1-300
bash scripts/ab_fwd.sh 2>&1 | tail -6
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-secondary "$@" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$*', '->', round(d['value']), 'graph-steps/s')"; }
run --steps 6 --warmup 2 --graphs-per-batch 100 --streams 3
run --steps 6 --warmup 2 --graphs-per-batch 100 --streams 6
run --steps 6 --warmup 2 --graphs-per-batch 50 --streams 8
run --steps 6 --warmup 2 --graphs-per-batch 200 --streams 5
run --steps 6 --warmup 2 --graphs-per-batch 340 --streams 3
