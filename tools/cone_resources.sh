#!/bin/bash
# Register / scratch use of the elliptic production variants of the development build (no library is produced): tools/cone_resources.sh [extra flags]
cd "$(dirname "$0")/../gym_quadruped_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-value -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -disable-machine-licm -DGQ_DEV_ONLY=1 "$@" --cuda-device-only -c -o /dev/null -Rpass-analysis=kernel-resource-usage gq_kernels.hip 2>&1 | grep -A9 "Function Name: _ZN2gq11step_kernelILi1ELi0ELb1" | grep "Function Name\|VGPRs:\|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
