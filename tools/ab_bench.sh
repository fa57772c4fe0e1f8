#!/bin/bash
# A/B of two libgq builds in one GPU session: tools/ab_bench.sh <alt .so> [bench args...]   (interleaved, 2 rounds)
ALT=$1; shift
for round in 1 2; do
  for lib in cur alt; do
    if [ $lib = alt ]; then export GQ_LIBGQ_PATH=$ALT; else unset GQ_LIBGQ_PATH; fi
    python bench.py --no-cpu-baseline --no-secondary --steps 1000 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$*', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us')"
  done
done
