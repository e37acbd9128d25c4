#!/bin/bash
# the short development call of round 3: GPU parity / range tests, then every library variant in ab_libs/ on the sampling line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_range.py -q -m gpu -x 2>&1 | tail -3 | cut -c1-300
ls ab_libs/*.so >/dev/null 2>&1 && bash scripts/ab_fwd.sh 2>&1 | tail -8
