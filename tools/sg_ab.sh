#!/bin/bash
# A/B helper (developer tool): pass times of the two Shampine-Gordon configurations for the library in rays_amd/lib
# (or RAYS_HIP_LIB), then the GPU tests that trace with the SG kernels.   bash tools/sg_ab.sh [label]
set -e
mkdir -p gpurun_out/r04
out=gpurun_out/r04/sg_coef_block_${1:-new}.txt
: > $out
for rep in 1 2; do
  python tools/variant_time_cfg.py configs/cfg5_axisym256k_sg_damp.in exact 3 >> $out
  python tools/variant_time_cfg.py configs/cfg3_solovev64k_sg_num.in exact 2 >> $out
done
cat $out
python -m pytest tests -m gpu -x -q -k "sg or SG or shampine or golden or parity or baseline" 2>&1 | tail -4
