#!/bin/bash
# GPU-box session for the pair exchange: equivalence test, headline A/B, wave timeline.   Usage: tools/xq_probe.sh [tag]
TAG=${1:-r06x}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=300 -k "pair_exchange or loaded_native" 2>&1 | tail -5 > $OUT/pytest_xq.txt; cat $OUT/pytest_xq.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_xq_on.json 2> $OUT/bench_xq_on.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-pair-exchange > $OUT/bench_xq_off.json 2> $OUT/bench_xq_off.err
for f in xq_on xq_off; do python -c "import json; d=json.load(open('$OUT/bench_$f.json')); print('$f', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us kernel')"; done
timeout 600 python tools/wave_timeline.py 4096 > $OUT/wave_timeline.txt 2>&1; grep -v "^  hint\|^  niter" $OUT/wave_timeline.txt | tail -22
timeout 600 python bench.py --no-cpu-baseline --no-secondary --self-collision capsule > $OUT/bench_capsule.json 2> $OUT/bench_capsule.err
python -c "import json; d=json.load(open('$OUT/bench_capsule.json')); print('capsule', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us kernel')"
