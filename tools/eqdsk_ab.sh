#!/bin/bash
# A/B helper (developer tool): pass times of the eqdsk configurations for the library in rays_amd/lib, then the GPU parity
# tests that touch the axisym equilibrium.   bash tools/eqdsk_ab.sh [label]
set -e
mkdir -p gpurun_out/r04
out=gpurun_out/r04/eqdsk_cell_search_${1:-new}.txt
: > $out
for rep in 1 2; do
  python tools/variant_time_cfg.py configs/cfg5b_axisym256k_rk4_damp.in tolerance 5 >> $out
  python tools/variant_time_cfg.py configs/cfg5b_axisym256k_rk4_damp.in exact 5 >> $out
  python tools/variant_time_cfg.py configs/cfg5_axisym256k_sg_damp.in exact 3 >> $out
done
cat $out
python -m pytest tests -m gpu -x -q -k "axisym or eqdsk or cfg5 or deposition or ray_init or golden or parity or profile" 2>&1 | tail -4
