mkdir -p gpurun_out/r06c
B="python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-secondary"
for cut in 0 1 2 4 5 6; do GQ_SELF_CUT=$cut $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('self_cut $cut', round(d['value']/1e6,2),'M', round(d['ms_per_step']*1000,2),'us')"; done > gpurun_out/r06c/cuts.txt 2>&1
$B --no-self-collision 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no self', round(d['value']/1e6,2),'M', round(d['ms_per_step']*1000,2),'us')" >> gpurun_out/r06c/cuts.txt 2>&1
cat gpurun_out/r06c/cuts.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "stagewise or self_collision_step_parity or benchmark_rollout" 2>&1 | grep -E "^E  |^tests/|Error|^FAILED|passed|failed|^>" | head -120 > gpurun_out/r06c/fails.log
tail -5 gpurun_out/r06c/fails.log
