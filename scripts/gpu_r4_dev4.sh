#!/bin/bash
# Round-4 development call: GPU tests, forward A/B over ab_libs/*.so, DiffBP / DiffSBDD sampler rows, serial per-launch trace
TAG=${1:-r04d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1700 python -m pytest tests -q -m gpu --maxfail=12 -p no:faulthandler --durations=8 -s 2>&1 | grep -v "^$" | grep -E "passed|failed|Error|error|assert|roll-out|ReLU flip|worst relative|FAILED|^[0-9.]+s " | tail -70 | cut -c1-400 | tee $OUT/pytest_gpu_$TAG.log
echo "== A/B forward =="
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_$TAG.log
echo "== diffbp / diffsbdd samplers, 200 graphs =="
for m in diffbp diffsbdd; do timeout 300 python bench.py --model $m --pockets 20 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m', d['value'], d['roofline']['per_kernel_us_avg_and_launches'])"; done | tee $OUT/samplers_$TAG.log
echo "== trace, no overlap =="
CBGX_OVERLAP=0 bash scripts/gpu_trace_sizes.sh ${TAG}_serial 2>&1 | tail -32 | cut -c1-330
