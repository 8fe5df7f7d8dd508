#!/bin/bash
# WRITE_SIZE + kernel time of every variant library on the headline fan
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in $R/rays_amd/lib/librays_hip_exp_*.so; do
  export RAYS_HIP_LIB=$lib
  rm -rf $R/gpurun_out/ws; mkdir -p $R/gpurun_out/ws
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ws -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob("$R/gpurun_out/ws/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if 'trace_kernel' in r['Kernel_Name']]
print("$lib".split('/')[-1], 'WRITE_SIZE MB/launch', sum(v)/len(v)*1024/1e6)
PY
  python3 $R/tools/fan_model.py short 2>&1 | grep "x1:" | head -1
done
