#!/bin/bash
# in-kernel section timers of the x2h backward (libcbgx_ablate.so, CBGX_BWD_ABL bit 512): share of wave time per section
cd ${GRAFT_REPO_ROOT:-/root/repo}
export CBGX_LIBRARY=$(pwd)/cbgbench_amd/lib/libcbgx_ablate.so
CBGX_BWD_ABL=$((512 + ${CBGX_BWD_ABL:-0})) python bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "bx prof" | head -${1:-2}
