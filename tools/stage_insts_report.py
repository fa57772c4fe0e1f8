import csv, glob, sys, collections
out = sys.argv[1]
order = [1, 2, 3, 4, 5, 14, 6, 7, 8, 9, 10, 11, 12, 0]
names = {1: 'S0+S1 load+kinematics', 2: 'S2 inertias', 3: 'S3 mass matrix', 4: 'S4 factor x2', 5: 'S5 rne+actuation', 14: 'S6a collision scan',
         6: 'S6b contact list', 7: 'S7 rows', 8: 'S8 qacc_smooth', 9: 'S9 solver', 10: 'S10 acc+dump', 11: 'euler+imu', 12: 'S11 obs+term', 0: 'gather+tail'}
vals = {}
for k in order:
    acc = collections.defaultdict(list)
    for fn in glob.glob(f'{out}/s{k}/**/*counter_collection.csv', recursive=True):
        rows = [r for r in csv.DictReader(open(fn)) if 'step_kernel' in r['Kernel_Name']]
        # the last 20 step launches of the process are the cut ones
        per = collections.defaultdict(list)
        for r in rows:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
        for c, v in per.items():
            acc[c] = v[-20:]
    vals[k] = {c: sum(v) / len(v) / 4096 for c, v in acc.items()}
print(f'{"stage":26s} {"VALU":>8s} {"SALU":>8s} {"LDS":>8s} {"wave cycles":>12s}   (per wave, mean over 4096 envs; cycles = 4 x SQ_WAVE_CYCLES)')
prev = collections.defaultdict(float)
for k in order:
    v = vals[k]
    if not v:
        continue
    d = {c: v[c] - prev[c] for c in v}
    print(f'{names[k]:26s} {d.get("SQ_INSTS_VALU", 0):8.0f} {d.get("SQ_INSTS_SALU", 0):8.0f} {d.get("SQ_INSTS_LDS", 0):8.0f} {4 * d.get("SQ_WAVE_CYCLES", 0):12.0f}')
    prev = v
v = vals[0]
print(f'{"total":26s} {v.get("SQ_INSTS_VALU", 0):8.0f} {v.get("SQ_INSTS_SALU", 0):8.0f} {v.get("SQ_INSTS_LDS", 0):8.0f} {4 * v.get("SQ_WAVE_CYCLES", 0):12.0f}')
