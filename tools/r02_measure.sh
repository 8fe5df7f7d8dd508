#!/bin/bash
# round-2 measurements quoted in DESIGN.md (raw output -> gpurun_out/r02/measurements/)
M=gpurun_out/r02/measurements; mkdir -p $M
python tools/time_configs.py 2>&1 | grep -v amdgpu.ids | tee $M/time_configs.txt
python tools/scan_speed.py 2>&1 | grep -v amdgpu.ids | tee $M/scan_speed.txt
python tools/host_entry_time.py 2>&1 | grep -v amdgpu.ids | tee $M/host_entry.txt
python bench.py --steps 20 --warmup 5 > $M/bench_n1.json 2> $M/bench_n1.err; cat $M/bench_n1.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fan-scale 4 --nstep-max 400 2>/dev/null | tee $M/bench_fan256k.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --config $PWD/configs/cfg5b_axisym256k_rk4_damp.in --exchange deposition 2>/dev/null | tee $M/bench_cfg5b_deposition.json
