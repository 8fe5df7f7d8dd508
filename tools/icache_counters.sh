#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in cfg5_axisym256k_sg_damp cfg3b_solovev64k_rk4; do
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$cfg -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --config $R/configs/$cfg.in > $OUT/$cfg.log 2>&1 || echo "pmc failed $cfg"
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/$cfg/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    if "trace_kernel" in k: print("$cfg", k, dict(v))
PY
done
