#!/bin/bash
# A/B/C... of several development builds of libgq (tools/dev_build.sh -> ab/*.so) in one GPU session, interleaved, 2 rounds.
# Usage: tools/ab_multi.sh "<lib1.so lib2.so ...>" "<bench args of case 1>" ["<bench args of case 2>" ...]
LIBS=$1; shift
for round in 1 2; do
  for args in "$@"; do
    for lib in $LIBS; do
      GQ_LIBGQ_PATH=$PWD/$lib python bench.py --no-cpu-baseline --no-secondary --steps 1000 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '[$args]', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us')"
    done
  done
done
