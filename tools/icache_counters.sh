#!/bin/bash
# Instruction-cache counters of the step kernel (rocprofv3 --pmc, separate passes, kernel trace only): tools/icache_counters.sh <out dir> [bench args]
OUT=$1; shift
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 20 "$@" > /dev/null 2> $OUT/pmc_$i.log
done
python - $OUT <<'PY'
import csv, glob, sys
tot = {}
for fn in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    acc = {}
    for row in csv.DictReader(open(fn)):
        if 'step_kernel' not in row.get('Kernel_Name', ''): continue
        acc.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
    for k, v in acc.items(): tot[k] = sum(v) / len(v)
for k in sorted(tot): print(f'{k:32s} {tot[k]:14.0f} per launch  {tot[k] / 4096:10.1f} per wave')
PY
