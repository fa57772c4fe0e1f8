#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats (+ PMC passes) for a bench workload.
# Usage: tools/run_profiles.sh <tag> [pmc|nopmc] [bench args...]      -> gpurun_out/prof_<tag>/{stats,pmc_*}
set -u
TAG=${1:-rXX}; PMC=${2:-pmc}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
(cd $ROOT && python -c "import bench; print(bench.kernel_source_hash())") > $OUT/kernel_src_sha16.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-secondary $*"
echo "$BENCH" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $OUT/stats -o stats -- $BENCH --steps 2000 --warmup 200 > $OUT/bench_stats.json 2> $OUT/stats.log
if [ "$PMC" = "pmc" ]; then
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" \
           "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_IFETCH SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- $BENCH --steps 60 --warmup 20 > /dev/null 2> $OUT/pmc_$i.log
done
fi
ls $OUT | head -20
