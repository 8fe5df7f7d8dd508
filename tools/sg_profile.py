"""Developer tool: per-section wave clocks of the SG kernel (library built with
tools/variant_build.sh sgprof -DRAYS_SG_PROFILE; run with RAYS_HIP_LIB pointing at it)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
NAMES = ["loop/refill", "init+vote", "RHS", "CHECK bookkeeping", "AFTER_F2", "AFTER_F3", "CRASH", "DE_BEGIN",
         "DE_TOP", "START_DONE", "COEF tail", "STOP+tail", "COEF coefficient block", "COEF scale+shift",
         "COEF predictor", "RHS, <= 8 lanes served",
         "F2: rows k, k-1", "F2: error sums", "F2: estimates + knew", "F2: accept (round-off rows)", "F3: get/set/add rows",
         "F3: order selection", "DE_TOP: intrp", "DE_TOP: wt + round", "COEF tail: round-off rows", "", "", "", "", "", "", ""]
for cfg, sym in (("configs/cfg5_axisym256k_sg_damp.in", "rays_debug_sg_profile_1_2_0_1_0"),
                 ("configs/cfg3_solovev64k_sg_num.in", "rays_debug_sg_profile_1_1_1_1_0")):
    nml, p, r0, n0 = bench.build_fan(cfg, 1)
    dt = DeviceTrace(p, r0, n0)
    fn = getattr(hip.load(), sym); fn.restype = C.c_int
    out = (C.c_ulonglong * 32)()
    dt.launch(); torch.cuda.synchronize(); fn(out, 1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
    fn(out, 1)
    v = np.array(list(out)[:32], dtype=np.float64); tot = v.sum()
    print(f"{os.path.basename(cfg)}: {e0.elapsed_time(e1):.1f} ms, {hip.kernel_name(p)}; wave-clock share per section:")
    trips = v[26]
    print(f"   lanes served per trip {v[25] / trips:.1f} of {v[27] / trips:.1f} holding a ray; waiting at an interval end {v[28] / trips:.1f}")
    v[25:29] = 0.0
    tot = v.sum()
    for n, x in zip(NAMES, v):
        if not n:
            continue
        print(f"   {n:20s} {100 * x / tot:5.1f} %")
