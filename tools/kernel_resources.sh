#!/bin/bash
# Register / scratch / spill figures of one kernel group as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage), NS = 2 kernels only (the FAST instantiation list).
# usage: tools/kernel_resources.sh <solver 0|1> <eq 0|1|2> <deriv 0|1> <unit-exp 0|1> [extra hipcc flags]
set -e
cd "$(dirname "$0")/../rays_amd/csrc"
s=$1; e=$2; d=$3; u=$4; shift 4
mkdir -p build_exp_res
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -DRAYS_INST_FAST \
  -DRAYS_INST_SOLVER=$s -DRAYS_INST_EQ=$e -DRAYS_INST_DERIV=$d -DRAYS_INST_UE=$u -DRAYS_INST_EQT=$((e + 4 * u)) "$@" \
  -Rpass-analysis=kernel-resource-usage -c rays_inst.hip -o build_exp_res/res_$s$e$d$u.o 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|SGPRs:|LDS Size" | sed 's/.*remark: [^ ]* //' | \
  sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | paste -s -d' ' | sed 's/Function Name:/\nFunction Name:/g'
echo
