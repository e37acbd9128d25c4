#!/bin/bash
# round-2 call c: corrected pipe-overlap micro-benchmarks, split-f16 MFMA probes, the re-specified config-sized gradient test
TAG=${1:-r02c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== mfma_f16_check =="
timeout 120 scripts/ubench/mfma_f16_check 2>&1 | tee $OUT/ubench_f16check_$TAG.log
echo "== ubench pipes =="
timeout 300 scripts/ubench/pipes 2>&1 | grep -v "^ds_read\|wave/SIMD: .*cyc/iter" | tee $OUT/ubench_pipes_$TAG.log
echo "== pytest config-sized gradients =="
timeout 900 python -m pytest tests/test_gpu_config_sized.py -q -s -k training_gradients 2>&1 | grep -v "^$" | tail -15 | tee $OUT/pytest_cfg_$TAG.log
