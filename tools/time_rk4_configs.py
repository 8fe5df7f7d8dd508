"""Time the RK4 configs that refill lanes (several passes each, best and mean)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
print("lib:", os.environ.get("RAYS_HIP_LIB", "default"), flush=True)
for cfg, scale, nstep, reps in (("configs/cfg3b_solovev64k_rk4.in", 1, None, 10), ("configs/cfg3b_solovev64k_rk4.in", 4, 400, 5),
                                ("configs/cfg5b_axisym256k_rk4_damp.in", 1, None, 10), ("configs/cfg4_slab1M_rk4.in", 1, None, 3)):
    nml, p, r0, n0 = bench.build_fan(cfg, 1, scale, nstep)
    dt = DeviceTrace(p, r0, n0)
    dt.launch(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    n = dt.npoints.cpu().numpy().astype(np.int64); st = np.maximum(n - 1, 0).sum()
    print(f"{os.path.basename(cfg)} x{scale} nray={len(n)} steps={st} best {min(ts):.3f} mean {np.mean(ts):.3f} ms  {hip.kernel_name(p, len(n))}", flush=True)
    del dt
