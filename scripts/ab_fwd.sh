#!/bin/bash
# A/B of forward-path library variants on ONE box: every ab_libs/*.so runs the sampling line (no CPU leg, no secondary) twice,
# interleaved.  Usage (on the box): bash scripts/ab_fwd.sh [extra bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for lib in ab_libs/*.so; do
  CBGX_LIBRARY=$(pwd)/$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={n: {'us_avg': v[0], 'launches': v[1]} for n, v in d['roofline']['per_kernel_us_avg_and_launches'].items()}
print('$lib', 'value', d['value'], 'x2h us', round(k['edge_x2h']['us_avg'],1), 'listed', round(k['edge_x2h_listed']['us_avg'],1), 'node_q', round(k['node_query']['us_avg'],1), 'node_g', round(k['node_gemm']['us_avg'],1), 'h2x', round(k['edge_h2x']['us_avg'],1))"
done; done
