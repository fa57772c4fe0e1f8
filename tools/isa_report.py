"""Static instruction mix of gq::step_kernel<solver> by source function (inlined callee lines are attributed to the
callee), from the gfx950 assembly with line tables.

    python tools/isa_report.py [solver]

The kernel is VALU-issue bound while all four waves of a SIMD are alive (profiles/*_sq_counters.md), so the static VALU
count per function - times how often the function runs per step - says where instructions are worth removing."""
import collections, re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / 'gym_quadruped_amd' / 'csrc'


def function_ranges(path):
    """[(first_line, name)] of the __device__ functions / stage markers of a header."""
    out = []
    for i, line in enumerate(path.read_text().splitlines(), 1):
        m = re.match(r'\s*(?:template\s*<[^>]*>\s*)?__device__\s+(?:__forceinline__|inline)?\s*[\w:<>\*&\s]+?\b(\w+)\s*\(', line)
        if m:
            out.append((i, m.group(1)))
        m = re.match(r'\s*/\* =+ (S\d+[^:]*)', line)
        if m:
            out.append((i, m.group(1).strip()))
        m = re.match(r'\s*GQ_TICK\((\d+)\)', line)
    return out


def main(solver='1'):
    ranges = {p.name: function_ranges(p) for p in CSRC.glob('*.h')}
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / 'k.s'
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', f'-I{ROOT}/include', f'-I{CSRC}',
                        '-fno-hip-fp32-correctly-rounded-divide-sqrt', '-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=iterative-maxocc', '-mllvm', '-disable-machine-licm', '-gline-tables-only', '-S', '--cuda-device-only', '-o', str(out), str(CSRC / 'gq_kernels.hip')],
                       check=True, capture_output=True)
        files, cur, infn = {}, (0, 0), False
        cnt = collections.defaultdict(collections.Counter)
        for line in out.read_text().splitlines():
            m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
            if m:
                files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
                continue
            m = re.match(r'^(_Z\w+):', line)
            if m:
                infn = m.group(1).startswith(f'_ZN2gq11step_kernelILi{solver}ELi0ELb0ELb0ELb1ELb1ELb0EEE')
                continue
            m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
            if m:
                cur = (int(m.group(1)), int(m.group(2)))
                continue
            m = re.match(r'\s+([a-z]\w+)', line)
            if not (infn and m) or line.lstrip().startswith('.'):
                continue
            op = m.group(1)
            kind = ('valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else
                    'vmem' if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')) else 'other')
            f = files.get(cur[0], '?')
            name = '?'
            for first, nm in ranges.get(f, []):
                if first <= cur[1]:
                    name = nm
            cnt[(f, name)][kind] += 1
    tot = collections.Counter()
    print(f'step_kernel<{solver}> static instruction counts')
    print(f'{"file":22s} {"function / stage":28s} {"valu":>6s} {"salu":>6s} {"lds":>6s} {"vmem":>6s}')
    for (f, name), c in sorted(cnt.items(), key=lambda kv: -kv[1]['valu']):
        print(f'{f:22s} {name:28s} {c["valu"]:6d} {c["salu"]:6d} {c["lds"]:6d} {c["vmem"]:6d}')
        tot.update(c)
    print(f'{"total":51s} {tot["valu"]:6d} {tot["salu"]:6d} {tot["lds"]:6d} {tot["vmem"]:6d}')


if __name__ == '__main__':
    main(*(sys.argv[1:2] or ['1']))
