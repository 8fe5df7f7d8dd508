"""Pass time of one configuration for the library in RAYS_HIP_LIB (A/B of builds on one box).
usage: python tools/variant_time_cfg.py configs/cfg4_slab1M_rk4.in [tolerance|exact] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
cfg = sys.argv[1]; hip.set_numerics(sys.argv[2] if len(sys.argv) > 2 else "tolerance"); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nml, p, r0, n0 = bench.build_fan(cfg, 1, 1, None)
dt = DeviceTrace(p, r0, n0)
dt.launch(); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"{os.path.basename(os.environ.get('RAYS_HIP_LIB', 'default'))} {os.path.basename(cfg)} {hip.kernel_name(p, len(r0))}: best {min(ts):.3f} mean {np.mean(ts):.3f} ms", flush=True)
