#!/usr/bin/env python
"""Per-class kernel times of ONE x2h and ONE h2x attention block on a bench-sized batch (200 graphs, ~100 k nodes), timed with the
library's own HIP-event profiler (cbgx_profile_*: kernels back to back, one stream).  A ~15 s process per library build:
    CBGX_LIBRARY=$PWD/ab_libs/<name>.so python scripts/ubench_stage.py [reps]
Used for A/B of node-stage and edge kernels without paying for a whole bench run per variant."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cbgbench_amd import _native, stages, synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only_x2h = len(sys.argv) > 2 and sys.argv[2] == "x2h"      # node_gemm class = the full projection launch alone
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
st = model.begin_sampling(synthetic.batch_to(bench.build_batch(20, 10, seed=1000), dev), keep_trajectory=False)
model.denoise_step(st, 500)                      # fills st["x"], st["h"] with a mid-trajectory state
x, h = st["x"].clone(), st["h"].clone()
lig, gen = st["lig_flag"].to(torch.uint8).contiguous(), st["gen_flag"].to(torch.uint8).contiguous()
gptr = st["graph_ptr"].to(torch.int32).contiguous()
packed = model.denoiser.packed_weights(dev)
nbr, deg = stages.knn_graph(x, gptr)
e_w = stages.edge_gate(packed, x, nbr, deg)
lib = _native.lib()
names = _native.PROFILE_CLASSES
for _ in range(2):
    stages.x2h_attention(packed, 3, x, h, nbr, deg, lig, e_w)
    stages.h2x_attention(packed, 3, x, h, nbr, deg, lig, gen, e_w)
torch.cuda.synchronize()
_native.check(lib.cbgx_profile_begin(16 * reps + 64), "cbgx_profile_begin")
for _ in range(reps):
    stages.x2h_attention(packed, 3, x, h, nbr, deg, lig, e_w)
    if not only_x2h:
        stages.h2x_attention(packed, 3, x, h, nbr, deg, lig, gen, e_w)
n = len(names)
ms = (ctypes.c_double * n)(); cnt = (ctypes.c_int * n)()
_native.check(lib.cbgx_profile_end(ms, cnt, n), "cbgx_profile_end")
print(os.path.basename(os.environ.get("CBGX_LIBRARY", "libcbgx.so")), "N", x.shape[0],
      " ".join(f"{nm}:{1e3 * ms[i] / cnt[i]:.1f}us/x{cnt[i] // reps}" for i, nm in enumerate(names) if cnt[i]))
