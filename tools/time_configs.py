"""Time the SG configs (one pass each)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
print("lib:", os.environ.get("RAYS_HIP_LIB", "default"), flush=True)
for cfg, reps in (("configs/cfg5_axisym256k_sg_damp.in", 2), ("configs/cfg3_solovev64k_sg_num.in", 1),
                  ("configs/cfg5b_axisym256k_rk4_damp.in", 5), ("configs/cfg4_slab1M_rk4.in", 2)):
    nml, p, r0, n0 = bench.build_fan(cfg, 1)
    dt = DeviceTrace(p, r0, n0)
    dt.launch(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): dt.launch(zero_fill=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    n = dt.npoints.cpu().numpy().astype(np.int64); st = np.maximum(n - 1, 0).sum()
    print(f"{os.path.basename(cfg)} nray={len(n)} steps={st} {ms:.3f} ms {st/ms*1e3:.3e} steps/s  {hip.kernel_name(p, len(n))}", flush=True)
