#!/bin/bash
# Round 5, ninth GPU call: does the x2h edge backward on fewer than 256 workgroups pay, now that the weight-gradient kernels wait on
# the auxiliary stream for free compute units (CBGX_BX_GRID)?  And hipGraph replay of the 35-launch one-graph step.
TAG=${1:-r05i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for rep in 1 2; do for g in 256 248 240 224; do CBGX_BX_GRID=$g timeout 200 python bench.py --workload train --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bx_grid=$g train', d['value'], 'graph-steps/s; ms/step', d['ms_per_step'])"; done; done | tee $OUT/ab_train_$TAG.log
for gr in off on; do for cfg in "1 1" "1 10"; do set -- $cfg
timeout 120 python bench.py --pockets $1 --samples $2 --steps 30 --warmup 5 --graph $gr --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph=$gr $1 x $2:', d['value'])"; done; done | tee $OUT/graph_$TAG.log
