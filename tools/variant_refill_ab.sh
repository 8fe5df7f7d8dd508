#!/bin/bash
# tools/refill_ab.py for every rays_amd/lib/librays_hip_exp_*.so given by name, on one box
for n in "$@"; do
  RAYS_HIP_LIB=$PWD/rays_amd/lib/librays_hip_exp_$n.so timeout -k 10 300 python tools/refill_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/refill_ab_$n.txt
done
