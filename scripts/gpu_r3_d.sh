#!/bin/bash
# quick A/B helper: selected tests + the sampling line without CPU leg / secondary block
TAG=${1:-r03d}; KEXPR=${2:-fold}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "$KEXPR" 2>&1 | grep -v "^$" | tail -5
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee $OUT/bench_quick_$TAG.json | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print(o['value'], o['ms_per_step'], r['frac'], r['avg_launch_us']); print({k:(v['us_avg'],v['launches']) for k,v in r['per_kernel'].items() if v['launches']})"
