#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for the default bench workload.
# Usage: tools/run_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{stats,pmc_*}
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
(cd $ROOT && python -c "import bench; print(bench.kernel_source_hash())") > $OUT/kernel_src_sha16.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/bench_stats.json 2> $OUT/stats.log
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$i.log
done
ls -R $OUT | head -40
