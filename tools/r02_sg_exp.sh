#!/bin/bash
# SG experiments: section profile of the current kernel + one timed pass per variant library
mkdir -p gpurun_out/r02
RAYS_HIP_LIB=$PWD/rays_amd/lib/librays_hip_exp_sgprof.so timeout -k 10 300 python tools/sg_profile.py > gpurun_out/r02/sg_profile.log 2>&1
cat gpurun_out/r02/sg_profile.log | grep -v amdgpu.ids
for lib in rays_amd/lib/librays_hip_exp_*.so; do
  case $lib in *sgprof*) continue;; esac
  RAYS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/time_configs.py 2>&1 | grep -v amdgpu.ids | grep -E "lib:|cfg5_|cfg3_" || exit 1
done
