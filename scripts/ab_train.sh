#!/bin/bash
# A/B of library variants on the training line: every ab_libs/*.so twice, interleaved.  Usage (on the box): bash scripts/ab_train.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for lib in ab_libs/*.so; do CBGX_LIBRARY=$(pwd)/$lib python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib train', d['value'], 'graph-steps/s; x2h backward', d['roofline']['avg_launch_us'], 'us')"; done; done
