#!/bin/bash
# cfg 3 (SG + finite-difference dD, 64k rays): pass time (bench.py events) and WRITE_SIZE per launch (rocprofv3 --pmc) of the
# lane-group kernel as built in each library given (file names under rays_amd/lib).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  export RAYS_HIP_LIB=$R/rays_amd/lib/$lib
  rm -rf $R/gpurun_out/ws; mkdir -p $R/gpurun_out/ws
  python3 $R/bench.py --config $R/configs/cfg3_solovev64k_sg_num.in --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ws/bench.json 2> /dev/null
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ws -- python3 $R/bench.py --config $R/configs/cfg3_solovev64k_sg_num.in --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv, glob, json
b = json.loads(open("$R/gpurun_out/ws/bench.json").read().strip().splitlines()[-1])
want = "rays::" + b["config"]["kernel"]
v = [float(r['Counter_Value']) for f in glob.glob("$R/gpurun_out/ws/**/*counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if r['Kernel_Name'].split('(')[0].replace('void ', '').strip() == want]
alg = b["roofline"]["algorithmic_bytes_per_launch"]
w = sum(v) / len(v) * 1024 if v else float('nan')
print("$lib", b["config"]["kernel"], "kernel_ms %.2f" % b["roofline"]["kernel_ms"], "WRITE_SIZE %.3f GB = %.2fx algorithmic" % (w / 1e9, w / alg))
PY
done
