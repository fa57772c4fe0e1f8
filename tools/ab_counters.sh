#!/bin/bash
# Hardware counters of the step kernel for several libgq dev builds on one robot (runs on the GPU box):
#   tools/ab_counters.sh "<lib> <lib> ..." <robot> "<counter group>" ["<counter group>" ...]
# Prints the per-launch mean of every counter for the step kernel; one rocprofv3 --pmc pass per group and library.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
LIBS=$1; ROBOT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do for lib in $LIBS; do
  D=/tmp/abc_$lib; rm -rf $D
  GQ_LIBGQ_PATH=$ROOT/ab/$lib.so rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $D -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --robot $ROBOT --steps 60 --warmup 20 > /dev/null 2> $D.log
  python - "$lib" $D <<'PY'
import sys, glob, csv, collections
lib, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'step_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
if not acc: print(lib, 'no counters (see log)'); print(open(d + '.log').read()[-600:])
for k, v in sorted(acc.items()):
    v = v[len(v) // 3:]
    print(f'{lib:12s} {k:28s} {sum(v) / len(v):16.1f}  (n={len(v)})')
PY
done; done
