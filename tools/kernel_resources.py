"""Register / scratch / LDS / occupancy table of every kernel of libgq as the COMPILER allocates them (rocprofv3's kernel-trace
`vgpr` column reports the allocation granule, not the allocation).

    python tools/kernel_resources.py [out.md] [--keep-asm /tmp/gq_kernels.s]

Compiles csrc/gq_kernels.hip for gfx950 with the product's flags plus -Rpass-analysis=kernel-resource-usage (device side only,
~3 min), parses the remarks and counts the scratch_load / scratch_store instructions of each kernel in the assembly."""
import collections, re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / 'gym_quadruped_amd' / 'csrc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', f'-I{ROOT}/include', f'-I{CSRC}', '-Wno-unused-value', '-fno-hip-fp32-correctly-rounded-divide-sqrt',
         '-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=iterative-maxocc', '-mllvm', '-disable-machine-licm', '-gline-tables-only',
         '-Rpass-analysis=kernel-resource-usage', '-S', '--cuda-device-only']


def short(name):
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r'^void ', '', d)
    d = re.sub(r'\(.*\)$', '', d)
    return d.replace('gq::', '')


def main(argv):
    out_md = Path(argv[0]) if argv and not argv[0].startswith('--') else None
    keep = Path(argv[argv.index('--keep-asm') + 1]) if '--keep-asm' in argv else None
    with tempfile.TemporaryDirectory() as td:
        asm = keep or Path(td) / 'k.s'
        r = subprocess.run(['/opt/rocm/bin/hipcc', *FLAGS, '-o', str(asm), str(CSRC / 'gq_kernels.hip')], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        rows, cur = collections.OrderedDict(), None
        for line in r.stderr.splitlines():
            m = re.search(r'remark: Function Name: (\S+)', line)
            if m:
                cur = m.group(1); rows[cur] = {}
                continue
            m = re.search(r'remark:\s+([A-Za-z][\w ]*?)(?: \[[\w/]+\])?: (\d+)', line)
            if m and cur:
                rows[cur][m.group(1).strip()] = int(m.group(2))
        scratch, fn = collections.Counter(), None
        for line in asm.read_text().splitlines():
            m = re.match(r'^(_Z\w+):', line)
            if m:
                fn = m.group(1)
            elif fn and re.match(r'\s*scratch_(load|store)', line):
                scratch[fn] += 1
    lines = ['| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | scratch instr. (static) | LDS B | occupancy (waves/SIMD) |', '|---|---|---|---|---|---|---|---|']
    for fn, d in rows.items():
        lines.append(f"| `{short(fn)}` | {d.get('VGPRs', '?')} | {d.get('AGPRs', '?')} | {d.get('TotalSGPRs', '?')} | {d.get('ScratchSize', '?')} | {scratch.get(fn, 0)} | "
                     f"{d.get('LDS Size', '?')} | {d.get('Occupancy', '?')} |")
    text = ('# Kernel resources as allocated by the compiler (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, product flags)\n\n'
            'step_kernel<SOLVER, MODE, CONE, BOXES, SELF, PRIM, PERSIST>: SOLVER 1 Newton / 0 PGS; MODE 0 production, 1 instrumented, 2 stage cut.\n'
            'mailbox_step_kernel<SOLVER, CONE, BOXES, SELF, PRIM>: the closed-loop persistent rollout.\n\n' + '\n'.join(lines) + '\n')
    if out_md:
        out_md.write_text(text)
    print(text)


if __name__ == '__main__':
    main(sys.argv[1:])
