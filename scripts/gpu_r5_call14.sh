#!/bin/bash
# Round 5, fourteenth GPU call: node_qfold_kernel with a tile's query pairs requested together (one wait, none inside the head loop),
# the first tile's list entry and queries around the LDS fill, rows by ds_bpermute.  ab_libs/base.so = the commit before.
TAG=${1:-r05n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -p no:faulthandler 2>&1 | grep -v "^$" | tail -4 | cut -c1-400 | tee $OUT/pytest_gpu_$TAG.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$1', d['value'], {n: v for n, v in k.items() if v[1]})"; }
for lib in tree base tree base; do
  if [ $lib = tree ]; then L=$ROOT/cbgbench_amd/lib/libcbgx.so; else L=$ROOT/ab_libs/$lib.so; fi
  CBGX_LIBRARY=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | line "$lib headline"
done | tee $OUT/ab_fwd_$TAG.log
for lib in tree base; do
  if [ $lib = tree ]; then L=$ROOT/cbgbench_amd/lib/libcbgx.so; else L=$ROOT/ab_libs/$lib.so; fi
  CBGX_LIBRARY=$L timeout 200 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib train', d['value'], {n: v[0] for n, v in k.items() if v[1] and n in ('node_query','node_gemm')})"
  CBGX_LIBRARY=$L timeout 90 python bench.py --pockets 1 --samples 10 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | line "$lib 10 graphs"
done | tee $OUT/ab_train_$TAG.log
