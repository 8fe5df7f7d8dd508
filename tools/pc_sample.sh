#!/bin/bash
# PC sampling of one bench.py configuration (rocprofv3 beta feature): where does a wave spend its time?
#   bash tools/pc_sample.sh TAG [bench.py arguments ...]  ->  gpurun_out/pcs_TAG/
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pcs_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for method in stochastic host_trap; do
  unit=cycles; interval=1048576
  [ $method = host_trap ] && unit=time && interval=100
  timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-method $method --pc-sampling-unit $unit \
     --pc-sampling-interval $interval --kernel-trace --output-format csv -d $OUT/$method -- \
     python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/$method.log 2>&1
  echo "$method rc=$?"; tail -3 $OUT/$method.log
  find $OUT/$method -name "*pc_sampling*" | head
  f=$(find $OUT/$method -name "*pc_sampling*csv" | head -1)
  [ -n "$f" ] && { wc -l $f; head -3 $f; break; }
done
