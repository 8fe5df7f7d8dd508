#!/bin/bash
# time every rays_amd/lib/librays_hip_exp_*.so on the headline fan
for lib in rays_amd/lib/librays_hip_exp_*.so; do
  RAYS_HIP_LIB=$PWD/$lib python tools/fan_model.py short 2>&1 | grep -v amdgpu.ids
done
