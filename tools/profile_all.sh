#!/bin/bash
# the three committed profiles of a round (headline RK4 fan, SG eqdsk fan, SG numerical fan)
bash tools/profile_bench.sh rk4_64k && bash tools/profile_bench.sh sg_eqdsk256k --config $PWD/configs/cfg5_axisym256k_sg_damp.in && bash tools/profile_bench.sh sg_num64k --config $PWD/configs/cfg3_solovev64k_sg_num.in
