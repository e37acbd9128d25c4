#!/bin/bash
# developer loop for the x2h backward kernel: spill report, rebuild the libraries, stage tests + training bench on the GPU box
# Usage: bash scripts/dev_bx.sh <tag> [extra command run on the box afterwards]
cd /root/repo || exit 1
( cd cbgbench_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only train_bwd_x2h.hip -o /tmp/bx.s 2>&1 | grep -v hip-link; grep -E "^\s+\.(vgpr_spill_count|sgpr_spill_count)" /tmp/bx.s | tr '\n' ' '; echo )
python -c "
from cbgbench_amd import build
build.build_native(); build.build_native(xcheck=True)" 2>&1 | tail -2
/usr/local/graft/bin/gpurun --timeout 1500 -- "bash scripts/gpu_train_quick.sh $1; $2" 2>&1 | grep -v "amdgpu.ids\|^\[gpurun\] \(sending\|merged\)" | tail -${3:-5}
