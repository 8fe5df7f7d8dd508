#!/bin/bash
# Round-4 evidence, one GPU call: the -m gpu tier (incl. the full-fan numerics survey -> numerics_evidence.json), the bench
# line, the rocprofv3 profiles of the kernels the BASELINE configs dispatch (both numerics flavours of the headline),
# pass times of every config, the Fortran call sites' timings on cfg 5b, and -- last, from
# the sources as they are -- the kernel resources.  Raw output -> gpurun_out/r04/; copy what is to be judged into profiles/r04/.
R=$PWD; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "EXIT $?" >> $O/gputest.log; tail -3 $O/gputest.log
cp gpurun_out/numerics_evidence.json $O/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; head -c 600 $O/bench_n1.json; echo
python tools/sg_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/sg_pass_times.txt
for n in exact tolerance; do RAYS_HIP_NUMERICS=$n python tools/time_rk4_configs.py 2>&1 | grep -v amdgpu.ids; done | tee $O/rk4_flavours_pass_times.txt
bash tools/dropin_trace_deposition_timing.sh > $O/dropin_cfg5b_trace_deposition_timing.txt 2>&1
bash tools/profile_bench.sh rk4_64k_tol > $O/prof_rk4_64k_tol.log 2>&1
bash tools/profile_bench.sh rk4_64k_exact --numerics exact > $O/prof_rk4_64k_exact.log 2>&1
bash tools/profile_bench.sh rk4_eqdsk256k --config $R/configs/cfg5b_axisym256k_rk4_damp.in > $O/prof_rk4_eqdsk256k.log 2>&1
bash tools/profile_bench.sh sg_num64k --config $R/configs/cfg3_solovev64k_sg_num.in > $O/prof_sg_num64k.log 2>&1
bash tools/profile_bench.sh sg_eqdsk256k --config $R/configs/cfg5_axisym256k_sg_damp.in > $O/prof_sg_eqdsk256k.log 2>&1
STEPS=5 bash tools/profile_bench.sh rk4_slab1M --config $R/configs/cfg4_slab1M_rk4.in > $O/prof_rk4_slab1M.log 2>&1
for t in rk4_64k_tol rk4_64k_exact rk4_eqdsk256k sg_num64k sg_eqdsk256k rk4_slab1M; do
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/prof_$t/pmc_summary.json"))
    print("$t", d["kernel"], "avg_ms %.3f" % (d["kernel_stats"].get("avg_ns", 0) / 1e6), {k: round(v, 4) for k, v in d["derived"].items() if k in ("write_amplification", "f64_arith_fraction_of_valu", "wave_time_parked_in_waitcnt", "valu_insts_per_wave")})
except Exception as e:
    print("$t: no summary:", e)
PY
done | tee $O/profile_overview.txt
# the line to be tracked: once more, now that this session's counters and numerics evidence exist (in this scratch copy of the
# repo), so that its roofline.traffic / roofline_fp64 / numerics_evidence are the ones of these sources
python tools/update_counters.py gpurun_out/prof_*/pmc_summary.json > /dev/null 2>&1; cp gpurun_out/numerics_evidence.json profiles/numerics_evidence.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; head -c 200 $O/bench_n1.json; echo
# (the N = 2 / 4 rehearsal is its own gpurun call -- `bash tools/rehearse_multi_gpu.sh`, ~10 min of gloo copies: with it
#  this script overran the 20-minute limit of one call)
python tools/kernel_resources_all.py > $O/kernel_resources.txt 2> $O/kernel_resources.err; tail -3 $O/kernel_resources.txt
