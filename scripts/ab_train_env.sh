#!/bin/bash
# interleaved A/B of the training lines under one environment knob of the in-tree library / host code (value 1 vs 0), three repetitions of
# 30 steps per model class.  Usage (on the box): bash scripts/ab_train_env.sh CBGX_FUSED_EMBED ["targetdiff diffbp diffsbdd"]
cd ${GRAFT_REPO_ROOT:-/root/repo}
KNOB=${1:-CBGX_FUSED_EMBED}
for rep in 1 2 3; do for model in ${2:-targetdiff diffbp diffsbdd}; do for v in 1 0; do
env $KNOB=$v python bench.py --workload train --model $model --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$model [$KNOB=$v]', d['value'])"; done; done; done
