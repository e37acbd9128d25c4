#!/bin/bash
# resident batches in flight: sweep of streams x edge workgroups on the sampling line (no CPU leg, no secondary)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${CFGS:-"1 0" "3 256" "5 256" "3 240" "5 240" "5 224" "1 0" "5 256"}; do
  set -- $cfg
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --streams $1 --edge-workgroups $2 $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $1 edge_wgs $2', 'value', d['value'], 'ms/step', d['ms_per_step'])"
done
