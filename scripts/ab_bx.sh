#!/bin/bash
# A/B of x2h-backward variants on ONE box: every gpurun_out/ab/*.so (copies of libcbgx.so built from different sources) runs the
# training bench twice, interleaved.  Usage (on the box): bash scripts/ab_bx.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for lib in ab_libs/*.so; do
  CBGX_LIBRARY=$(pwd)/$lib python bench.py --workload train --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={n: {'us_avg': v[0], 'launches': v[1]} for n, v in d['roofline']['per_kernel_us_avg_and_launches'].items()}
print('$lib', 'x2h_bwd us', round(k['edge_x2h_bwd']['us_avg'],1), 'listed', round(k['edge_x2h_bwd_listed']['us_avg'],1), 'ms/step', d['ms_per_step'])"
done; done
