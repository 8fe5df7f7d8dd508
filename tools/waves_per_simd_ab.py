import os, sys, zlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
for cfg, scale, nstep in (("configs/cfg3b_solovev64k_rk4.in", 4, 400), ("configs/cfg3b_solovev64k_rk4.in", 2, 400), ("configs/cfg3b_solovev64k_rk4.in", 8, 400)):
    nml, p, r0, n0 = bench.build_fan(cfg, 1, scale, nstep)
    for flavour in ("exact", "tolerance"):
        hip.set_numerics(flavour)
        for force in ("", "1", "", "1"):
            if force: os.environ["RAYS_HIP_FORCE_WAVES_PER_SIMD"] = force
            else: os.environ.pop("RAYS_HIP_FORCE_WAVES_PER_SIMD", None)
            dt = DeviceTrace(p, r0, n0)
            dt.launch(); torch.cuda.synchronize()
            ts = []
            for _ in range(6):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"x{scale} {len(r0)} rays {flavour:9s} force={force or '-'} best {min(ts):.3f} mean {np.mean(ts):.3f} ms  {hip.kernel_name(p, len(r0))}", flush=True)
            del dt
