"""Developer tool: per-section wave clocks of the lane-group SG kernel on cfg 3 (library built with
tools/variant_build.sh sgprof -DRAYS_SG_PROFILE; run with RAYS_HIP_LIB pointing at it)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
NAMES = {0: "loop/refill", 1: "init + order cap + vote", 2: "RHS (no check)", 15: "RHS on check trips", 3: "CHECK bookkeeping",
         4: "AFTER_F2", 5: "AFTER_F3", 7: "CRASH + DE_BEGIN", 8: "DE_TOP (intrp / step entry)", 9: "START_DONE",
         12: "COEF coefficient block", 13: "COEF scale + shift", 14: "COEF predictor", 10: "COEF tail", 11: "STOP + summaries"}
cfg, sym = "configs/cfg3_solovev64k_sg_num.in", "rays_debug_sg_profile_1_1_1_1_0"
nml, p, r0, n0 = bench.build_fan(cfg, 1)
dt = DeviceTrace(p, r0, n0)
fn = getattr(hip.load(), sym); fn.restype = C.c_int
out = (C.c_ulonglong * 32)()
dt.launch(); torch.cuda.synchronize(); fn(out, 1)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
fn(out, 1)
v = np.array(list(out)[:32], dtype=np.float64)
trips = v[26]
print(f"{os.path.basename(cfg)}: {e0.elapsed_time(e1):.1f} ms, {hip.kernel_name(p, len(r0))}")
print(f"   wave trips {trips:.0f}; lanes served per trip {v[25] / trips:.1f} of {v[27] / trips:.1f} holding a ray; waiting at an interval end {v[28] / trips:.1f}")
v[25:29] = 0.0
tot = v.sum()
print(f"   wave clocks per trip {tot / trips:.0f}")
for i, n in NAMES.items():
    print(f"   {n:28s} {100 * v[i] / tot:5.1f} %   {v[i] / trips:8.0f} clocks per trip")
