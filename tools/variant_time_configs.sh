#!/bin/bash
# one timed pass of the SG / eqdsk configs with every variant library
for lib in rays_amd/lib/librays_hip_exp_*.so; do
  RAYS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/time_configs.py 2>&1 | grep -v amdgpu.ids || exit 1
done
