for lib in rays_amd/lib/librays_hip_exp_*.so; do
  export RAYS_HIP_LIB=$PWD/$lib
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "sg" 2>&1 | tail -1
  timeout -k 10 300 python scratch/sg_time.py 2>&1 | grep -v amdgpu.ids || exit 1
done
