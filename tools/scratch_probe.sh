#!/bin/bash
# Scratch bytes / static scratch instructions of the development-build kernel variants.   usage: tools/scratch_probe.sh <cone 0|1> <boxes 0|1|2> [extra flags]
# (boxes 1: world-box variants for hull-only robots, 2: with the exact primitive pairs - aliengo, go2, b2)
CONE=${1:-0}; BOXES=${2:-2}; shift; shift
cd "$(dirname "$0")/../gym_quadruped_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -I../../include -Wno-unused-value -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
  -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -disable-machine-licm -DGQ_DEV_ONLY=$CONE -DGQ_DEV_BOXES=$BOXES -gline-tables-only \
  -Rpass-analysis=kernel-resource-usage -S --cuda-device-only "$@" -o /tmp/dev_probe.s gq_kernels.hip 2> /tmp/dev_probe.err || { tail -20 /tmp/dev_probe.err; exit 1; }
python3 - <<'PY'
import re, collections
rows, cur = {}, None
for line in open('/tmp/dev_probe.err'):
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z][\w ]*?)(?: \[[\w/]+\])?: (\d+)', line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
cnt, fn = collections.Counter(), None
for line in open('/tmp/dev_probe.s'):
    m = re.match(r'^(_Z\w+):', line)
    if m: fn = m.group(1)
    elif fn and re.match(r'\s*scratch_(load|store)', line): cnt[fn] += 1
for fn, d in rows.items():
    if 'step_kernel' in fn:
        print(f"{fn[:70]:70s} VGPR {d.get('VGPRs')} scratch {d.get('ScratchSize')} B/lane, {cnt.get(fn, 0)} scratch instr., occupancy {d.get('Occupancy')}")
PY
