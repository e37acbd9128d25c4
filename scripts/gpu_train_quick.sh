#!/bin/bash
# quick call for the training path: stage + gradient tests, then the training bench row.
# Usage: bash scripts/gpu_train_quick.sh <tag> [pytest -k expr]
TAG=${1:-t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py -q -x -m gpu ${2:+-k "$2"} 2>&1 | grep -v "^$" | tail -25 | tee $OUT/pytest_train_$TAG.log
timeout 600 python bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('value', d['value'], d['unit'], 'ms/step', d['ms_per_step'])
print({k: (round(v['us_avg'],1), v['launches']) for k,v in {n: {'us_avg': v[0], 'launches': v[1]} for n, v in (r.get('per_kernel_us_avg_and_launches') or {}).items()}.items() if v['launches']})"
