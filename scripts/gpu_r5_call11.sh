#!/bin/bash
# Round 5, eleventh GPU call: (1) node_stage_kernel -- one index load per tile (rows via ds_bpermute, predicates via ballots), a
# unit's sixteen operand loads in flight together, no scratch; 8-wave variant (two workgroups per CU) for inputs above 160 row tiles;
# (2) gate kernels at four waves per SIMD (LDS image re-read per tile instead of hoisted into 330 registers).
# Libraries: the tree (both changes), ab_libs/base.so (HEAD before them), gate1 / gate2 (tree with 1 / 2 waves per SIMD in the gate
# kernels), noloadsfirst (tree without the sched_barrier of phase 1).  Before the call:
#   git stash; python scripts/build_variant.py base; git stash pop
#   python scripts/build_variant.py gate1 -DCBGX_GATE_WAVES_PER_EU=1; python scripts/build_variant.py gate2 -DCBGX_GATE_WAVES_PER_EU=2
#   python scripts/build_variant.py noloadsfirst -DCBGX_NS_LOADS_FIRST=0
TAG=${1:-r05k}
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
  small gate1 $1 $2 CBGX_LIBRARY=$ROOT/ab_libs/gate1.so
  small w16 $1 $2 CBGX_NODE_STAGE_WAVES=16
  small w8 $1 $2 CBGX_NODE_STAGE_WAVES=8
  small noloadsfirst $1 $2 CBGX_LIBRARY=$ROOT/ab_libs/noloadsfirst.so
done 2>&1 | tee $OUT/small_$TAG.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$1', d['value'], {n: v for n, v in k.items() if v[1]})"; }
for lib in tree base gate1 tree base; do
  if [ $lib = tree ]; then L=$ROOT/cbgbench_amd/lib/libcbgx.so; else L=$ROOT/ab_libs/$lib.so; fi
  CBGX_LIBRARY=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | line "$lib headline"
done | tee $OUT/ab_fwd_$TAG.log
for lib in tree gate1; do
  if [ $lib = tree ]; then L=$ROOT/cbgbench_amd/lib/libcbgx.so; else L=$ROOT/ab_libs/$lib.so; fi
  CBGX_LIBRARY=$L timeout 200 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib train', d['value'], {n: v[0] for n, v in k.items() if v[1] and n in ('gate','knn')})"; done | tee $OUT/ab_train_$TAG.log
du -sh $OUT | tail -1
