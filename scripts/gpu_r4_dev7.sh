#!/bin/bash
# Round-4 development call: GPU tests (parity, config-sized, host-facing), forward A/B, serial trace
TAG=${1:-r04g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1700 python -m pytest tests -q -m gpu --maxfail=12 -p no:faulthandler -s 2>&1 | grep -v "^$" | grep -E "passed|failed|Error|error|assert|roll-out|ReLU flip|worst relative|FAILED|pocket frame" | tail -40 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
echo "== A/B forward =="
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
echo "== small batches =="
for a in "--pockets 1 --samples 10 --graphs-per-batch 10" "--pockets 1 --samples 1 --graphs-per-batch 1"; do for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', '$a', d['value'])"; done; done | tee $OUT/small_$TAG.log
echo "== trace, no overlap =="
CBGX_OVERLAP=0 bash scripts/gpu_trace_sizes.sh ${TAG}_serial 2>&1 | tail -22 | cut -c1-250
