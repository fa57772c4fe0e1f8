ROOT=$PWD
for round in 1 2; do
for pair in "base aliengo" "box aliengo" "basec go2" "boxc go2" "base mini_cheetah" "box mini_cheetah"; do set -- $pair
GQ_LIBGQ_PATH=$ROOT/ab/$1.so timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 1000 --robot $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us')"
done; done
GQ_TALLY_FILE=/dev/null GQ_LIBGQ_PATH=$ROOT/ab/box.so python -m pytest tests/test_gpu_parity.py -q -x -k "benchmark_rollout_states and (aliengo or b2)" 2>&1 | tail -1
GQ_TALLY_FILE=/dev/null GQ_LIBGQ_PATH=$ROOT/ab/boxc.so python -m pytest tests/test_gpu_parity.py -q -x -k "benchmark_rollout_states and (go2 or go1)" 2>&1 | tail -1
