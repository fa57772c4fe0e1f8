#!/bin/bash
# Fast development build of libgq for kernel A/B experiments: only the flat-scene self-collision Newton variants are instantiated
# (gq_kernels.hip GQ_DEV_ONLY), ~15 s instead of 90.   Usage: tools/dev_build.sh <out .so> [cone 0|1] [extra hipcc flags...]
# Development builds also read the profiling knobs GQ_STOP_STAGE / GQ_FORCE_SELF / GQ_SELF_CUT / GQ_MB_FLAGS from the environment (-DGQ_DEV_KNOBS);
# the product library reads no environment variable.
# The result goes under ab/ (git-ignored, travels to the GPU box) and is selected with GQ_LIBGQ_PATH (tools/ab_bench.sh).
set -e
OUT=$1; CONE=${2:-0}; shift; shift || true
cd "$(dirname "$0")/../gym_quadruped_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I. -I../../include -Wno-unused-value -Werror=pass-failed \
  -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -disable-machine-licm \
  -DGQ_DEV_ONLY=$CONE -DGQ_DEV_KNOBS "$@" -o "$OUT" gq_kernels.hip gq_api.hip gq_host_model.cpp
