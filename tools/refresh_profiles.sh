#!/bin/bash
# Turn one `tools/gpu_session.sh <tag> profiles configs timeline bench` session (gpurun_out/) into the tracked summaries.
# Usage: tools/refresh_profiles.sh <tag> [round prefix, default r02]
set -u
T=$1; R=${2:-r02}; PO=${PROFILES_OUT:-profiles}; mkdir -p $PO
P=gpurun_out/prof_$T
python tools/profile_summary.py stats $P/stats/stats_results.db $PO/${R}_kernel_stats.md "$R kernel stats: python bench.py --no-cpu-baseline --no-secondary --steps 2000 --warmup 200 (mini_cheetah flat, 4096 envs, Newton, self-collision on)" $P/bench_stats.json
python tools/profile_summary.py pmc $P/pmc_1,$P/pmc_2 $PO/${R}_hbm_counters.md 4096 1735 > $PO/latest_traffic.json
python tools/profile_summary.py sq $P/pmc_3,$P/pmc_4,$P/pmc_5,$P/pmc_6,$P/pmc_7,$P/pmc_8 $PO/${R}_sq_counters.md 4096 > $PO/latest_sq.json
# raw rocprofv3 --stats CSVs (small) are kept beside the summaries
RAW=$PO/raw_${R}; mkdir -p $RAW
for t in "" _noself _capsule _cfg3 _cfg4 _cfg5; do
  for f in gpurun_out/prof_${T}$t/stats/*kernel_stats.csv gpurun_out/prof_${T}$t/stats/*domain_stats.csv; do [ -f "$f" ] && cp "$f" $RAW/headline${t}_$(basename $f); done
done
for t in noself capsule cfg3 cfg4 cfg5; do
  Q=gpurun_out/prof_${T}_$t
  python tools/profile_summary.py stats $Q/stats/stats_results.db $PO/${R}_kernel_stats_$t.md "$R kernel stats: bench.py --no-cpu-baseline --no-secondary $(cut -d' ' -f5- $Q/command.txt) --steps 2000 --warmup 200" $Q/bench_stats.json
  sed -n 7p $PO/${R}_kernel_stats_$t.md
done
sed -n 7p $PO/${R}_kernel_stats.md
for f in cfg3_aliengo_perlin cfg4_go2_flat cfg5_hyqreal1_boxes_imu_hm driver_like_20steps; do
  cp gpurun_out/$T/bench_$f.json $PO/${R}_bench_$f.json
  python -c "import json; d=json.loads(open('$PO/${R}_bench_$f.json').read().strip().splitlines()[0]); print('$f', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us')"
done
cp gpurun_out/$T/bench.json $PO/${R}_bench.json
cp gpurun_out/$T/wave_timeline.txt $PO/${R}_wave_timeline.txt
cp gpurun_out/$T/wave_timeline_noself.txt $PO/${R}_wave_timeline_noself.txt
[ -f gpurun_out/$T/latest_tail.json ] && cp gpurun_out/$T/latest_tail.json $PO/latest_tail.json
[ -f gpurun_out/$T/ubench_convex_pair.txt ] && cp gpurun_out/$T/ubench_convex_pair.txt $PO/${R}_ubench_convex_pair.txt
for f in niter_vs_oracle_go2 niter_vs_oracle_hyqreal1 niter_vs_oracle_mini_cheetah ubench_fma_issue stage_cuts4096 stage_cuts4096_aliengo niter_hist; do [ -f gpurun_out/$T/$f.txt ] && cp gpurun_out/$T/$f.txt $PO/${R}_$f.txt; done
cat $PO/latest_traffic.json
