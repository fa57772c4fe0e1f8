#!/bin/bash
# Interleaved A/B of several development libraries (ab/<name>.so, tools/dev_build.sh) in one GPU session, 2 rounds:
#   tools/ab_multi3.sh [-a "<bench args>"] name1 name2 ...
ARGS=""
if [ "$1" = "-a" ]; then ARGS="$2"; shift; shift; fi
for round in 1 2; do
  for lib in "$@"; do
    GQ_LIBGQ_PATH=$PWD/ab/$lib.so python bench.py --no-cpu-baseline --no-secondary --steps 1000 $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$ARGS', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us')"
  done
done
