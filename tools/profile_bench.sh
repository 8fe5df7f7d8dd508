#!/bin/bash
# Collect the rocprofv3 evidence for one bench.py configuration on the GPU box.
#   bash tools/profile_bench.sh TAG [bench.py arguments ...]
# Writes gpurun_out/prof_TAG/{kernel_stats.csv,pmc_summary.json,bench.json}; copy what should be judged
# into profiles/rNN/.  Counters are collected in their own passes with --kernel-trace only
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; FETCH_SIZE is doubled on gfx950).
tag=$1; shift
STEPS=${STEPS:-20}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-host-entry --no-pipelined "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-host-entry --no-pipelined "$@" > $OUT/stats.log 2>&1 || exit 1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
i=0
for set in "WRITE_SIZE" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry --no-pipelined "$@" > $OUT/p$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 $R/tools/summarize_profile.py $OUT "$*"
