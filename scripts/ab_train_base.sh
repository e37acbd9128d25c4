#!/bin/bash
# careful A/B of the training line over every ab_libs/*.so (scripts/build_variant.py; `base` = the commit before): three repetitions,
# interleaved, 30 steps each.  Usage (on the box): [MODEL=diffbp] bash scripts/ab_train_base.sh ["ENV=v ..."]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for lib in ab_libs/*.so; do
env $1 CBGX_LIBRARY=$(pwd)/$lib python bench.py --workload train --model ${MODEL:-targetdiff} --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib [$1]', d['value'])"; done; done
