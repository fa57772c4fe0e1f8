#!/bin/bash
# (the GQ_STOP_STAGE / GQ_SELF_CUT / GQ_FORCE_SELF knobs used below exist in DEVELOPMENT builds only - tools/dev_build.sh, -DGQ_DEV_KNOBS, selected with
# GQ_LIBGQ_PATH; the product library reads no environment variable)
# Dynamic instruction counts per stage (runs on the GPU box): the step kernel is cut short after stage marker k
# (GQ_STOP_STAGE) and SQ_INSTS_VALU / SALU / LDS + SQ_WAVE_CYCLES are collected per cut; differences between
# consecutive cuts are the stage costs.  Markers in execution order: 1 2 3 4 5 14 6 7 8 9 10 11 12 13 (0 = full).
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/stage_insts
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 4 5 14 6 7 8 9 10 11 12 0; do
  GQ_STOP_STAGE=$k rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/s$k -o pmc -- \
    python $ROOT/tools/stage_insts_run.py > /dev/null 2> $OUT/s$k.log
done
python $ROOT/tools/stage_insts_report.py $OUT
