#!/bin/bash
TAG=${1:-r02g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/scripts/gpu_quick.sh $TAG all
bash $ROOT/scripts/gpu_pmc_x2h.sh $TAG 2>&1 | grep -A24 "edge_mfma_kernel<true, 8, false>" | head -30
