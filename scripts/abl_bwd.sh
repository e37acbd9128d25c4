#!/bin/bash
# timing ablations of the edge backward kernels.  CBGX_BWD_ABL bits, h2x (train_bwd_mfma.hip): 1 no neighbour atomics, 2 no
# neighbour gathers, 4 no rbf-column / rbf gradient MFMAs, 8 no softmax, 16 no T/S stores; x2h (train_bwd_x2h.hip): 1 no
# projection-row atomics, 2 no d rbf pass, 4 no rbf-column MFMAs, 8 no backward at all (three forwards only), 16 no rbf
# pre-activation MFMAs, 32 no third phase (key recompute + backward).  Results are WRONG by design, so the
# switch exists only in a separate build: libcbgx_ablate.so (-DCBGX_ABLATE), selected through CBGX_LIBRARY.
python -c "from cbgbench_amd.build import build_native; print(build_native(ablate=True))" || exit 1
export CBGX_LIBRARY=$(pwd)/cbgbench_amd/lib/libcbgx_ablate.so
for a in ${ABLS:-0 1 2 4 8 16 32 24}; do
  CBGX_BWD_ABL=$a python bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={n: {'us_avg': v[0], 'launches': v[1]} for n, v in d['roofline']['per_kernel_us_avg_and_launches'].items()}
print('abl=$a', 'x2h_bwd us', k['edge_x2h_bwd']['us_avg'], 'h2x_bwd us', k['edge_h2x_bwd']['us_avg'], 'ms/step', d['ms_per_step'])"
done
