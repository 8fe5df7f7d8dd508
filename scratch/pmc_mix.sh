#!/bin/bash
# instruction-mix counters for the headline kernel (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mix
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'trace_kernel' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print(k[0], k[1], len(v), sum(v)/len(v))
PY
