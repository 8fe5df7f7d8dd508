#!/bin/bash
# tools/refill_ab.py sg (cfg 5 and cfg 3 pass times + checksums) for every rays_amd/lib/librays_hip_exp_*.so named, one box
for n in "$@"; do
  RAYS_HIP_LIB=$PWD/rays_amd/lib/librays_hip_exp_$n.so timeout -k 10 300 python tools/refill_ab.py sg 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sg_ab_$n.txt
done
