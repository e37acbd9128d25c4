#!/bin/bash
# Round 5, eighth GPU call: the cached-graph gate on the new ranks only (large inputs: two launches, CBGX_MERGE_GATE=1) against the
# round-4 pair (=0): static-context / parity tests, then the 40-graph row and the headline twice each, interleaved
TAG=${1:-r05h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu (parity + config-sized samplers) =="
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sized.py -q -m gpu -x -k "not train and not grad" -p no:faulthandler 2>&1 | grep -v "^$" | tail -5 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
echo "== A/B =="
for rep in 1 2; do for mg in 1 0; do
CBGX_MERGE_GATE=$mg timeout 90 python bench.py --pockets 4 --samples 10 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('merge_gate=$mg 4 x 10 graphs:', d['value'], {n: v for n, v in k.items() if v[1]})"
CBGX_MERGE_GATE=$mg timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('merge_gate=$mg headline', d['value'], {n: v for n, v in k.items() if v[1]})"
done; done | tee $OUT/ab_fwd_$TAG.log
