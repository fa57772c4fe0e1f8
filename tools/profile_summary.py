"""Turn rocprofv3 outputs under gpurun_out/<dir> into the small tracked summaries under profiles/."""
import csv, glob, json, sqlite3, sys
from pathlib import Path

def kernel_stats(db, out_md, title, bench_json=None):
    c = sqlite3.connect(db)
    rows = list(c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'))
    with open(out_md, 'w') as f:
        f.write(f'# {title}\n\nMI355X (gfx950). Durations in microseconds (rocprofv3 --kernel-trace --stats).\n\n')
        f.write('| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n')
        for n, calls, tot, avg, p in rows[:8]:
            f.write(f'| {n.split("(")[0][:70]} | {calls} | {tot:.1f} | {avg:.2f} | {p:.2f} |\n')
        r = list(c.execute("select name, lds_size, scratch_size, grid_x, workgroup_x, avg(duration), count(*), min(duration), max(duration) from kernels where name like 'void gq::%' or name like 'gq::%' group by name"))
        f.write('\nlibgq kernels (ns; register counts: the compiler\'s table in profiles/rNN_kernel_resources.md - rocprofv3\'s vgpr column is the allocation granule, not the use):\n\n| kernel | lds B | scratch B | grid | wg | avg ns | n | min ns | max ns |\n|---|---|---|---|---|---|---|---|---|\n')
        for x in r:
            f.write('| ' + x[0].split('(')[0] + ' | ' + ' | '.join(str(int(v)) for v in x[1:]) + ' |\n')
        if bench_json and Path(bench_json).exists():
            f.write('\nbench line of the profiled run:\n\n```\n' + Path(bench_json).read_text().strip() + '\n```\n')

def pmc(dirs, out_md, n_envs, bytes_per_env):
    with open(out_md, 'w') as f:
        f.write('# HBM traffic counters of gq::step_kernel (rocprofv3 --pmc, one counter group per pass)\n\n')
        f.write('FETCH_SIZE / WRITE_SIZE are in KiB (L2 <-> fabric requests); per MI355X_MICROARCH.md the gfx950 FETCH_SIZE\n'
                'under-counts wide coalesced reads by 2x, so the corrected read figure doubles it.\n\n')
        tot = {}
        for d in dirs:
            for fn in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
                acc = {}
                for row in csv.DictReader(open(fn)):
                    if 'step_kernel' not in row.get('Kernel_Name', ''): continue
                    acc.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
                for k, v in acc.items():
                    tot[k] = (sum(v) / len(v), len(v))
        f.write('| counter | mean per launch | launches |\n|---|---|---|\n')
        for k, (m, n) in tot.items():
            f.write(f'| {k} | {m:.1f} | {n} |\n')
        if 'FETCH_SIZE' in tot and 'WRITE_SIZE' in tot:
            rd, wr = tot['FETCH_SIZE'][0] * 1024, tot['WRITE_SIZE'][0] * 1024
            alg = n_envs * bytes_per_env
            f.write(f'\nper launch ({n_envs} envs): fetched {rd/1e6:.2f} MB (x2 corrected {2*rd/1e6:.2f} MB), written {wr/1e6:.2f} MB; '
                    f'algorithmic {alg/1e6:.2f} MB ({bytes_per_env} B/env-step). traffic (corrected) = {(2*rd+wr)/1e6:.2f} MB '
                    f'= {(2*rd+wr)/alg:.2f}x algorithmic.\n')
            sha = None
            for d in dirs:   # run_profiles.sh leaves the hash of the kernel sources it profiled next to the pass directories
                h = Path(d).parent / 'kernel_src_sha16.txt'
                if h.exists():
                    sha = h.read_text().strip() or None
            print(json.dumps({'traffic_bytes_per_launch': 2 * rd + wr, 'fetch': rd, 'write': wr, 'kernel_src_sha16': sha,
                              'profile': str(out_md)}))

def sq(dirs, out_md, n_envs):
    """Instruction-mix / occupancy counters of gq::step_kernel (SQ_* groups), normalised per wave (= per env-step)."""
    tot = {}
    for d in dirs:
        for fn in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
            acc = {}
            for row in csv.DictReader(open(fn)):
                if 'step_kernel' not in row.get('Kernel_Name', ''): continue
                acc.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
            for k, v in acc.items():
                if k.startswith('SQ_'):
                    tot[k] = (sum(v) / len(v), len(v))
    with open(out_md, 'w') as f:
        f.write('# SQ counters of gq::step_kernel (rocprofv3 --pmc, four counters per pass, no trace domains besides --kernel-trace)\n\n')
        f.write(f'{n_envs} waves per launch (one env per wavefront).  *_CYCLES / ACTIVE / WAIT counters are in quad-cycles (x4 = shader cycles).\n\n')
        f.write('| counter | mean per launch | per wave | launches |\n|---|---|---|---|\n')
        for k in sorted(tot):
            m, n = tot[k]
            f.write(f'| {k} | {m:.0f} | {m / n_envs:.1f} | {n} |\n')
        g = lambda k: tot.get(k, (0, 0))[0]
        if g('SQ_WAVE_CYCLES'):
            sha = None
            for d in dirs:
                h = Path(d).parent / 'kernel_src_sha16.txt'
                if h.exists():
                    sha = h.read_text().strip() or None
            print(json.dumps({'valu_per_wave': g('SQ_INSTS_VALU') / n_envs, 'salu_per_wave': g('SQ_INSTS_SALU') / n_envs, 'lds_per_wave': g('SQ_INSTS_LDS') / n_envs,
                              'active_lanes_per_valu': (g('SQ_THREAD_CYCLES_VALU') / g('SQ_INSTS_VALU')) if g('SQ_INSTS_VALU') and g('SQ_THREAD_CYCLES_VALU') else None,
                              'wave_cycles': 4 * g('SQ_WAVE_CYCLES') / n_envs, 'kernel_src_sha16': sha, 'profile': str(out_md)}))
            f.write(f'\nper wave: {g("SQ_INSTS_VALU")/n_envs:.0f} VALU, {g("SQ_INSTS_SALU")/n_envs:.0f} SALU, {g("SQ_INSTS_LDS")/n_envs:.0f} LDS, '
                    f'{(g("SQ_INSTS_VMEM_RD")+g("SQ_INSTS_VMEM_WR"))/n_envs:.0f} VMEM instructions over {4*g("SQ_WAVE_CYCLES")/n_envs:.0f} cycles of wave lifetime; '
                    f'VALU busy {g("SQ_ACTIVE_INST_VALU")/g("SQ_WAVE_CYCLES")*100:.0f} % of a wave\'s lifetime (x4 resident waves per SIMD), '
                    f'waiting on any instruction {g("SQ_WAIT_INST_ANY")/g("SQ_WAVE_CYCLES")*100:.0f} %, on LDS {g("SQ_WAIT_INST_LDS")/g("SQ_WAVE_CYCLES")*100:.0f} %.\n')

if __name__ == '__main__':
    if sys.argv[1] == 'sq':
        sq(sys.argv[2].split(','), sys.argv[3], int(sys.argv[4]))
    elif sys.argv[1] == 'stats':
        kernel_stats(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
    else:
        pmc(sys.argv[2].split(','), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
