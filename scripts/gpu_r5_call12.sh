#!/bin/bash
# Round 5, twelfth GPU call: kNN merge / gate / scan kernels with a wave-uniform centre index (scalar graph search), one 64-bit compare
# per key pair, centre-only loads ahead of the search; graph_cache_begin's proximity walk four atoms per step; node_stage_kernel with
# 8 waves everywhere (and 4 as a forced variant).  ab_libs/base.so = the commit before (git stash; build_variant.py base; git stash pop).
TAG=${1:-r05l}
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
  small w4 $1 $2 CBGX_NODE_STAGE_WAVES=4
  small w16 $1 $2 CBGX_NODE_STAGE_WAVES=16
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
