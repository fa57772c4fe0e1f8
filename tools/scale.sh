#!/bin/bash
# One-command scaling run on an 8-GPU MI355X node (the driver's SCALE protocol): the headline workload at N = 1, 2, 4, 8 ranks - one
# process per GPU, 4096 envs each, independent shards, no data-path collective (torch.distributed over RCCL only for the start / stop
# barrier and the max-over-ranks time) - and BASELINE config 4 in its full form (go2 flat, 32768 envs = 8 x 4096).
#   usage: tools/scale.sh [out_dir] [steps] [warmup]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}; STEPS=${2:-2000}; WARM=${3:-200}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
echo "visible GPUs: $NGPU" | tee "$OUT/scale.log"
python bench.py --gpus 1 --steps $STEPS --warmup $WARM > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
for N in 2 4 8; do
  [ "$N" -le "$NGPU" ] || { echo "skip N=$N (only $NGPU GPUs)" | tee -a "$OUT/scale.log"; continue; }
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps $STEPS --warmup $WARM \
    --no-secondary > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
done
if [ "$NGPU" -ge 8 ]; then   # config 4: go2 flat, 32768 envs sharded over 8 GPUs
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29600 bench.py --gpus 8 --robot go2 --steps $STEPS --warmup $WARM \
    --no-secondary --no-cpu-baseline > "$OUT/bench_cfg4_go2_32768.json" 2> "$OUT/bench_cfg4.err"
fi
python - "$OUT" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
base = None
for f in sorted(glob.glob(os.path.join(out, 'bench_n*.json')), key=lambda p: int(p.split('_n')[1].split('.')[0])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:
        print(f, 'no JSON line:', ex); continue
    base = base or d['value'] / d['n_gpus']
    print(f"N={d['n_gpus']}: {d['value'] / 1e6:8.2f} M env-steps/s  {d['ms_per_step'] * 1e3:6.1f} us/step  per-GPU {d['value'] / d['n_gpus'] / 1e6:6.2f} M  ({d['value'] / d['n_gpus'] / base:.3f} of N=1)")
PY
