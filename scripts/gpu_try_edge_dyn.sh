#!/bin/bash
# First evaluation of the forward edge kernels' dynamic remainder (edge_mfma.hip, -DCBGX_EDGE_DYN: prepared at the end of round 4,
# never run on a GPU).  Before the call, in the build container:
#   rm -f ab_libs/*.so; python scripts/build_variant.py base; python scripts/build_variant.py dyn2 -DCBGX_EDGE_DYN=2
#   python scripts/build_variant.py w4 -DCBGX_EDGE_SMALL_W4=1     (4-wave workgroups for inputs of <= 1016 nodes: the 1-graph rows)
# Then (one gpurun, ~6 GPU-minutes): the forward / sampler / training parity tests on the variant, the small-batch rows, the training
# line and the headline on both libraries.
TAG=${1:-dyn}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
CBGX_LIBRARY=$(pwd)/ab_libs/dyn2.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_config_sized.py -m gpu -q -x \
  -k "not rollout_200 and not diffbp_training and not diffsbdd_training" 2>&1 | grep -E "passed|failed|Error|^E |FAILED" | cut -c1-400 | head -20 | tee $OUT/pytest_dyn_$TAG.log
if [ -f ab_libs/w4.so ]; then
CBGX_LIBRARY=$(pwd)/ab_libs/w4.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|^E |FAILED" | cut -c1-400 | head -10 | tee -a $OUT/pytest_dyn_$TAG.log
fi
for lib in ab_libs/*.so ab_libs/*.so; do for cfg in "1 1" "1 10" "4 10"; do set -- $cfg
CBGX_LIBRARY=$(pwd)/$lib timeout 60 python bench.py --pockets $1 --samples $2 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib', '$1 x $2 graphs:', d['value'], {n: v[0] for n, v in k.items() if n.startswith('edge')})"
done; done | tee $OUT/small_dyn_$TAG.log
bash scripts/ab_train.sh 2>&1 | tee $OUT/ab_train_dyn_$TAG.log
bash scripts/ab_fwd.sh 2>&1 | tee $OUT/ab_fwd_dyn_$TAG.log
