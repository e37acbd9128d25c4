#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python scripts/dev_pack_diff.py ab_libs/old.so cbgbench_amd/lib/libcbgx.so 2>/dev/null | tail -6 | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu --durations=3 2>&1 | grep -v "^$" | tail -12 | cut -c1-300
