#!/bin/bash
# Round 5, thirteenth GPU call: the first item's header of every edge launch requested behind the LDS fill's loads (and h2x's copy of the
# fixed nodes behind them too), node stage phase-2 operands ahead of the LayerNorm, cached kNN scan with 8 slots per lane and 64-bit
# compares, graph_lists with batched LDS reads.  ab_libs: base = the commit before; lateheader (-DCBGX_EDGE_EARLY_HEADER=0) and p2late
# (-DCBGX_NS_P2_EARLY=0) = the tree without one change each.
TAG=${1:-r05m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -p no:faulthandler 2>&1 | grep -v "^$" | tail -6 | cut -c1-400 | tee $OUT/pytest_gpu_$TAG.log
small() {  # label pockets samples [env...]
  local lab=$1 p=$2 s=$3; shift 3
  env "$@" timeout 90 python bench.py --pockets $p --samples $s --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lab', '$p x $s graphs:', d['value'], {n: v[0] for n, v in k.items() if v[1]})"
}
for cfg in "1 1" "1 10"; do set -- $cfg
  for rep in 1 2; do
    small tree $1 $2 A=1
    small base $1 $2 CBGX_LIBRARY=$ROOT/ab_libs/base.so
  done
  small lateheader $1 $2 CBGX_LIBRARY=$ROOT/ab_libs/lateheader.so
  small p2late $1 $2 CBGX_LIBRARY=$ROOT/ab_libs/p2late.so
done 2>&1 | tee $OUT/small_$TAG.log
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
print('$lib train', d['value'], {n: v[0] for n, v in k.items() if v[1] and n in ('gate','knn')})"; done | tee $OUT/ab_train_$TAG.log
du -sh $OUT | tail -1
