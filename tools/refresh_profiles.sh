#!/bin/bash
# Turn one `tools/gpu_session.sh <tag> profiles configs timeline bench` session (gpurun_out/) into the tracked summaries.
# Usage: tools/refresh_profiles.sh <tag> [round prefix, default r02]
set -u
T=$1; R=${2:-r02}
P=gpurun_out/prof_$T
python tools/profile_summary.py stats $P/stats/stats_results.db profiles/${R}_kernel_stats.md "$R kernel stats: python bench.py --no-cpu-baseline --no-secondary --steps 2000 --warmup 200 (mini_cheetah flat, 4096 envs, Newton, self-collision on)" $P/bench_stats.json
python tools/profile_summary.py pmc $P/pmc_1,$P/pmc_2 profiles/${R}_hbm_counters.md 4096 1735 > profiles/latest_traffic.json
python tools/profile_summary.py sq $P/pmc_3,$P/pmc_4,$P/pmc_5,$P/pmc_6,$P/pmc_7 profiles/${R}_sq_counters.md 4096
for t in noself cfg3 cfg4 cfg5; do
  Q=gpurun_out/prof_${T}_$t
  python tools/profile_summary.py stats $Q/stats/stats_results.db profiles/${R}_kernel_stats_$t.md "$R kernel stats: bench.py --no-cpu-baseline --no-secondary $(cut -d' ' -f5- $Q/command.txt) --steps 2000 --warmup 200" $Q/bench_stats.json
  sed -n 7p profiles/${R}_kernel_stats_$t.md
done
sed -n 7p profiles/${R}_kernel_stats.md
for f in cfg3_aliengo_perlin cfg4_go2_flat cfg5_hyqreal1_boxes_imu_hm driver_like_20steps; do
  cp gpurun_out/$T/bench_$f.json profiles/${R}_bench_$f.json
  python -c "import json; d=json.loads(open('profiles/${R}_bench_$f.json').read().strip().splitlines()[0]); print('$f', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us')"
done
cp gpurun_out/$T/bench.json profiles/${R}_bench.json
cp gpurun_out/$T/wave_timeline.txt profiles/${R}_wave_timeline.txt
cp gpurun_out/$T/wave_timeline_noself.txt profiles/${R}_wave_timeline_noself.txt
cat profiles/latest_traffic.json
