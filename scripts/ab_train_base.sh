#!/bin/bash
# careful A/B of the training line: ab_libs/base.so (the commit before) against ab_libs/new.so under the training path's per-call
# schedule knobs; three repetitions, interleaved, 30 steps each.  Usage (on the box): bash scripts/ab_train_base.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env $2 CBGX_LIBRARY=$(pwd)/ab_libs/$1.so python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 [$2]', d['value'])"; }
for rep in 1 2 3; do
run base ""
run new ""
run new "CBGX_TRAIN_FWD_OVERLAP=0 CBGX_TRAIN_ZERO_ROWS=0"
run new "CBGX_BX_EDGE_ROWS=1"
done
