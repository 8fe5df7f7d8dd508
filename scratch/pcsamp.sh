#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pcsamp
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --kernel-trace --output-format csv -d $OUT/st -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline "$@" > $OUT/st.log 2>&1
echo "stochastic rc=$?"
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace --output-format csv -d $OUT/ht -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline "$@" > $OUT/ht.log 2>&1
echo "host_trap rc=$?"
find $OUT -type f | head -30
for f in $(find $OUT -name "*pc_sampling*csv"); do echo $f; wc -l $f; head -5 $f; done
tail -5 $OUT/st.log $OUT/ht.log
