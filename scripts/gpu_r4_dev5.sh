#!/bin/bash
# Round-4 development call: GPU tests of the files touched, batch-size / stream sweeps of the driver line, PMC of node_proj
TAG=${1:-r04e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu (config-sized + parity) =="
timeout 1700 python -m pytest tests/test_gpu_config_sized.py tests/test_gpu_parity.py tests/test_gpu_training.py -q -m gpu --maxfail=12 -p no:faulthandler -s 2>&1 | grep -v "^$" | grep -E "passed|failed|Error|error|assert|roll-out|ReLU flip|worst relative|FAILED|pocket frame" | tail -50 | cut -c1-400 | tee $OUT/pytest_gpu_$TAG.log
echo "== sweeps =="
run() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['config']['ms_per_denoising_step_of_the_job'])"; }
for rep in 1 2; do
run --graphs-per-batch 200 --streams 3
run --graphs-per-batch 250 --streams 3
run --graphs-per-batch 340 --streams 3
run --graphs-per-batch 500 --streams 2
run --graphs-per-batch 200 --streams 2
run --graphs-per-batch 200 --streams 4
done | tee $OUT/sweep_$TAG.log
echo "== PMC node_proj =="
bash scripts/gpu_pmc_fwd_kernel.sh "node_proj_kernel" ${TAG}_nproj 2>&1 | tail -60 | cut -c1-200
