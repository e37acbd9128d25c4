#!/bin/bash
# timing probe of a small edge launch (wrong results by design, variants of a working copy): p1 = launch + LDS image only,
# p2 = no image loads; 1 graph and 10 graphs
TAG=${1:-r04z2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for lib in ab_libs/*.so; do for cfg in "1 1" "1 10"; do set -- $cfg
CBGX_LIBRARY=$(pwd)/$lib timeout 60 python bench.py --pockets $1 --samples $2 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib', '$1 x $2 graphs:', d['value'], {n: v[0] for n, v in k.items()})"
done; done | tee $OUT/probe_$TAG.log
