#!/bin/bash
# Annotated ISA of the development build's headline kernel + its resource line: tools/isa_dev.sh <tag> [extra flags]  ->  /tmp/isa/<tag>.txt
TAG=$1; shift
mkdir -p /tmp/isa
cd "$(dirname "$0")/../gym_quadruped_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -I../../include -Wno-unused-value -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -disable-machine-licm -DGQ_DEV_ONLY=0 -gline-tables-only -S --cuda-device-only "$@" -o /tmp/isa/$TAG.s gq_kernels.hip 2>&1 | grep -v "warning\|^$" | head
cd ../..
K=step_kernelILi1ELi0ELb0ELb0ELb1ELb1ELb0E
python tools/isa_walk.py /tmp/isa/$TAG.s $K /tmp/isa/$TAG.txt
echo "instructions $(grep -vc ':$' /tmp/isa/$TAG.txt)  s_waitcnt $(grep -c s_waitcnt /tmp/isa/$TAG.txt)  s_load $(grep -c s_load_dword /tmp/isa/$TAG.txt)  scratch $(grep -c scratch_ /tmp/isa/$TAG.txt)"
awk "/^_ZN2gq11$K/{f=1} f&&/; (NumVgprs|NumSgprs|ScratchSize|Occupancy)/{printf \"%s \", \$0} f&&/Occupancy/{print \"\"; exit}" /tmp/isa/$TAG.s
