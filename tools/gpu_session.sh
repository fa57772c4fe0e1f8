#!/bin/bash
# (the GQ_STOP_STAGE / GQ_SELF_CUT / GQ_FORCE_SELF knobs used below exist in DEVELOPMENT builds only - tools/dev_build.sh, -DGQ_DEV_KNOBS, selected with
# GQ_LIBGQ_PATH; the product library reads no environment variable)
# One GPU-box session (via gpurun): everything writes under gpurun_out/<tag>/.   Usage: tools/gpu_session.sh <tag> [parts...]
# parts: probe tests perf bench profiles
set -u
TAG=${1:-r02x}; shift
PARTS=${*:-"probe tests perf bench"}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for p in $PARTS; do
  case $p in
    probe) timeout 300 python tools/mujoco_probe.py > $OUT/mujoco_probe.txt 2>&1; cp gpurun_out/mujoco_probe.log $OUT/ 2>/dev/null;;
    tests) timeout 2400 python -m pytest tests -m gpu -q -x --timeout=900 -s 2>&1 | tail -150 > $OUT/pytest_gpu.txt;;
    testsall) timeout 3000 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | tail -250 > $OUT/pytest_gpu.txt;;
    perf) timeout 600 python tools/perf_probe.py stages 64 > $OUT/stages64.txt 2>&1
          timeout 600 python tools/perf_probe.py stages 256 > $OUT/stages256.txt 2>&1
          timeout 600 python tools/perf_probe.py stages 4096 > $OUT/stages4096.txt 2>&1
          timeout 600 python tools/stage_cuts.py 4096 > $OUT/stage_cuts4096.txt 2>&1
          timeout 600 python tools/stage_cuts.py 256 > $OUT/stage_cuts256.txt 2>&1
          timeout 600 python tools/wave_timeline.py 4096 > $OUT/wave_timeline.txt 2>&1;;
    bench) timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err;;
    configs) # BASELINE configs 3, 4 (one shard), 5 + PGS / default obs: driver-grade bench lines
          timeout 600 python bench.py --robot aliengo --scene perlin --no-cpu-baseline > $OUT/bench_cfg3_aliengo_perlin.json 2> $OUT/bench_cfg3.err
          timeout 600 python bench.py --robot go2 --no-cpu-baseline > $OUT/bench_cfg4_go2_flat.json 2> $OUT/bench_cfg4.err
          timeout 600 python bench.py --robot hyqreal1 --scene random_boxes --imu --heightmap --no-cpu-baseline > $OUT/bench_cfg5_hyqreal1_boxes_imu_hm.json 2> $OUT/bench_cfg5.err
          timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_like_20steps.json 2> $OUT/bench_20.err;;
    parity) timeout 1500 python tests/reports/newton_parity_report.py 512 > $OUT/newton_parity_report.txt 2>&1;;
    selfab) timeout 600 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_self_on.json 2> $OUT/bench_self_on.err
            timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-self-collision > $OUT/bench_self_off.json 2> $OUT/bench_self_off.err
            GQ_FORCE_SELF=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-self-collision > $OUT/bench_self_variant_nopairs.json 2> $OUT/bench_self_variant_nopairs.err
            for f in self_on self_off self_variant_nopairs; do python -c "import json; d=json.load(open('$OUT/bench_$f.json')); print('$f', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us kernel')"; done;;
    stages) timeout 600 python tools/perf_probe.py stages 256 > $OUT/stages256.txt 2>&1
            timeout 600 python tools/perf_probe.py stages 4096 > $OUT/stages4096.txt 2>&1;;
    selfcut) for c in 0 2 4; do GQ_SELF_CUT=$c timeout 600 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_selfcut$c.json 2> $OUT/bench_selfcut$c.err; python -c "import json; d=json.load(open('$OUT/bench_selfcut$c.json')); print('self cut $c', round(d['value']/1e6,2), 'M', round(d['roofline']['kernel_ms']*1e3,1), 'us kernel')"; done;;
    timeline) timeout 600 python tools/wave_timeline.py 4096 > $OUT/wave_timeline.txt 2>&1
              timeout 600 python tools/wave_timeline.py 4096 mini_cheetah noself > $OUT/wave_timeline_noself.txt 2>&1
              cp gpurun_out/latest_tail.json $OUT/ 2>/dev/null;;
    robots) for r in go2 aliengo hyqreal1; do
              timeout 600 python tools/wave_timeline.py 4096 $r > $OUT/wave_timeline_$r.txt 2>&1
              timeout 600 python tools/wave_timeline.py 4096 $r noself > $OUT/wave_timeline_${r}_noself.txt 2>&1
              timeout 600 python tools/perf_probe.py stages 4096 $r > $OUT/stages4096_$r.txt 2>&1
            done;;
    extras) for r in go2 hyqreal1 mini_cheetah; do timeout 600 python tools/niter_vs_oracle.py $r 384 2>&1 | grep -v amdgpu.ids > $OUT/niter_vs_oracle_$r.txt; done
            timeout 600 python tools/stage_cuts.py 4096 > $OUT/stage_cuts4096.txt 2>&1
            timeout 600 python tools/stage_cuts.py 4096 aliengo > $OUT/stage_cuts4096_aliengo.txt 2>&1
            timeout 900 python tools/niter_hist.py mini_cheetah aliengo go2 hyqreal1 go1 b2 2>&1 | grep -v amdgpu.ids > $OUT/niter_hist.txt
            (cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/fma_issue fma_issue.hip && /tmp/fma_issue) > $OUT/ubench_fma_issue.txt 2>&1;;
    convex) (hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igym_quadruped_amd/csrc -Iinclude -DGQ_CVX_STATS -Wno-unused-result -o /tmp/convex_pair tools/ubench/convex_pair.hip 2>/dev/null && /tmp/convex_pair) > $OUT/ubench_convex_pair.txt 2>&1;;
    nscan) timeout 900 python tools/nscan.py > $OUT/nscan.txt 2>&1;;
    profiles) timeout 1500 bash tools/run_profiles.sh $TAG pmc > $OUT/run_profiles.txt 2>&1
              timeout 600 bash tools/run_profiles.sh ${TAG}_noself nopmc --no-self-collision >> $OUT/run_profiles.txt 2>&1
              timeout 600 bash tools/run_profiles.sh ${TAG}_capsule nopmc --self-collision capsule >> $OUT/run_profiles.txt 2>&1
              timeout 600 bash tools/run_profiles.sh ${TAG}_cfg3 nopmc --robot aliengo --scene perlin >> $OUT/run_profiles.txt 2>&1
              timeout 600 bash tools/run_profiles.sh ${TAG}_cfg4 nopmc --robot go2 >> $OUT/run_profiles.txt 2>&1
              timeout 900 bash tools/run_profiles.sh ${TAG}_cfg5 nopmc --robot hyqreal1 --scene random_boxes --imu --heightmap >> $OUT/run_profiles.txt 2>&1;;
  esac
done
tail -5 $OUT/pytest_gpu.txt 2>/dev/null; cat $OUT/bench.json 2>/dev/null | head -c 600
