#!/bin/bash
# experiment (measured neutral, not in the tree): the first item's loads of the edge kernels requested before the LDS image
# (-DCBGX_EDGE_EARLY_FIRST=1 variant of a working copy).  Small-batch rows and the headline on ab_libs/base.so (HEAD) against ab_libs/early.so
TAG=${1:-r04z1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for lib in ab_libs/base.so ab_libs/early.so ab_libs/base.so ab_libs/early.so; do for cfg in "1 1" "1 10"; do set -- $cfg
CBGX_LIBRARY=$(pwd)/$lib timeout 60 python bench.py --pockets $1 --samples $2 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib', '$1 x $2 graphs:', d['value'], {n: v[0] for n, v in k.items() if n.startswith('edge')})"
done; done | tee $OUT/small_$TAG.log
for lib in ab_libs/base.so ab_libs/early.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 60 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib headline', d['value'], d['roofline']['avg_launch_us'])"; done | tee -a $OUT/small_$TAG.log
