"""Attribute the step kernel's register spills (scratch_load / scratch_store) to source lines.

    python tools/spill_report.py [solver]      # solver: 1 Newton (default), 0 PGS

Compiles csrc/gq_kernels.hip to gfx950 assembly with line tables and counts the scratch instructions of
gq::step_kernel<solver> per (file, line).  Spills cost twice here: latency on the wave's critical path and HBM
WRITE_SIZE traffic (profiles/*_hbm_counters.md)."""
import collections, re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / 'gym_quadruped_amd' / 'csrc'


def main(solver='1', SELF='0'):
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / 'k.s'
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', f'-I{ROOT}/include', f'-I{CSRC}',
                        '-fno-hip-fp32-correctly-rounded-divide-sqrt', '-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=iterative-maxocc', '-mllvm', '-disable-machine-licm', '-gline-tables-only', '-S', '--cuda-device-only', '-o', str(out), str(CSRC / 'gq_kernels.hip')],
                       check=True, capture_output=True)
        files, cur, infn, cnt = {}, (0, 0), False, collections.Counter()
        for line in out.read_text().splitlines():
            m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
            if m:
                files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
                continue
            m = re.match(r'^(_Z\w+):', line)
            if m:
                infn = m.group(1).startswith(f'_ZN2gq11step_kernelILi{solver}ELi0ELb0ELb0ELb{SELF}ELb1ELb0EEE')
                continue
            m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
            if m:
                cur = (int(m.group(1)), int(m.group(2)))
                continue
            m = re.match(r'\s*scratch_(load|store)', line)
            if infn and m:
                cnt[(files.get(cur[0], '?'), cur[1], m.group(1))] += 1
    print(f'step_kernel<{solver}>: {sum(cnt.values())} scratch instructions')
    for (f, l, kind), v in sorted(cnt.items()):
        print(f'  {f}:{l:<5d} {kind:5s} {v}')


if __name__ == '__main__':
    main(*(sys.argv[1:3] or ['1']))
