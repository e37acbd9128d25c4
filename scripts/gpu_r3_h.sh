#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stages_match_reference and mfma" 2>&1 | grep -v "^$" | tail -30 | cut -c1-300
timeout 300 python - <<'PY'
import torch, numpy as np, ctypes
import cbgbench_amd as C
from cbgbench_amd import _native, synthetic_weights
from cbgbench_amd.csrc_layout import *  # noqa
PY
