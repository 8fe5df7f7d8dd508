"""cfg 5b (and the 99900-ray Solovev fan) with the rays handed out in index order and long-first with neighbourhoods of
2, 4 (default) and 8 rays: RAYS_HIP_RAY_ORDER, read by the library at every launch."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
for cfg, scale, nstep in (("configs/cfg5b_axisym256k_rk4_damp.in", 1, None),):
    nml, p, r0, n0 = bench.build_fan(cfg, 1, scale, nstep)
    for flavour in ("exact", "tolerance"):
        hip.set_numerics(flavour)
        for order in ("index", "pilot2", "pilot", "pilot8", "index", "pilot2", "pilot", "pilot8"):
            os.environ["RAYS_HIP_RAY_ORDER"] = order
            dt = DeviceTrace(p, r0, n0)
            dt.launch(); torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            crc = zlib.crc32(dt.npoints.cpu().numpy().tobytes())
            print(f"{os.path.basename(cfg)} {flavour:9s} {order:6s} best {min(ts):.3f} mean {np.mean(ts):.3f} ms  {hip.kernel_name(p, len(r0))} counts {crc:08x}", flush=True)
            del dt
hip.set_numerics("exact")
