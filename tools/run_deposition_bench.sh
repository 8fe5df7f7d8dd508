#!/bin/bash
# parity of the device deposition profiles + bench.py with the deposition-profile exchange on the
# eqdsk + damping fan, next to the plain trace
python -m pytest tests/test_gpu_parity.py -q -k deposition 2>&1 | tail -2
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --config $PWD/configs/cfg5b_axisym256k_rk4_damp.in
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --config $PWD/configs/cfg5b_axisym256k_rk4_damp.in --exchange deposition
