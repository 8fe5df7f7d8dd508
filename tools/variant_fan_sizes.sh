#!/bin/bash
for lib in rays_amd/lib/librays_hip_exp_*.so; do
  RAYS_HIP_LIB=$PWD/$lib python tools/fan_sizes.py 2>&1 | grep -v amdgpu.ids
done
