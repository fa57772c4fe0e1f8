#!/bin/bash
# A/B of libgq dev builds on several robots: [ABX="<extra bench args>"] tools/ab_robots.sh "<lib> <lib> ..." "<robot> <robot> ..." [rounds]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${3:-2}); do
for robot in $2; do for lib in $1; do
GQ_LIBGQ_PATH=$ROOT/ab/$lib.so timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 1000 --robot $robot $ABX 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $robot', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us')"
done; done; done
