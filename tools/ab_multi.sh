#!/bin/bash
# A/B/C... of several libgq builds in one GPU session (interleaved rounds): tools/ab_multi.sh "<lib1> <lib2> ..." [bench args...]
# Each lib is first held to the benchmark-state parity test (0 mismatches against the oracle), then timed.
LIBS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for lib in $LIBS; do
  GQ_TALLY_FILE=/dev/null GQ_LIBGQ_PATH=$ROOT/ab/$lib.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "benchmark_rollout_states and mini_cheetah" 2>&1 | tail -1 | sed "s/^/parity $lib: /"
done
for round in 1 2 3; do
  for lib in $LIBS; do
    GQ_LIBGQ_PATH=$ROOT/ab/$lib.so timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 1500 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$*', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us')"
  done
done
