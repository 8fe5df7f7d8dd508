#!/bin/bash
# the committed profiles of a round: headline RK4 fan, eqdsk RK4 fan, 1M-ray slab fan, SG eqdsk fan, SG numerical fan
R=$PWD
bash tools/profile_bench.sh rk4_64k && \
bash tools/profile_bench.sh rk4_eqdsk256k --config $R/configs/cfg5b_axisym256k_rk4_damp.in && \
bash tools/profile_bench.sh sg_eqdsk256k --config $R/configs/cfg5_axisym256k_sg_damp.in && \
bash tools/profile_bench.sh sg_num64k --config $R/configs/cfg3_solovev64k_sg_num.in && \
bash tools/profile_bench.sh rk4_slab1M --config $R/configs/cfg4_slab1M_rk4.in
