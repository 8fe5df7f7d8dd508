"""Repeats the long-first hand-out on cfg 5b and on a 99900-ray Solovev fan and checks every launch for the same bits in
every result array (and npoints >= 1 everywhere): the hand-out races between waves differently on every launch."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for cfg, scale, nstep in (("configs/cfg5b_axisym256k_rk4_damp.in", 1, None), ("configs/cfg3b_solovev64k_rk4.in", 1, 300)):
    nml, p, r0, n0 = bench.build_fan(cfg, 1, scale, nstep)
    if len(r0) == 65536:  # more rays than the launch has lanes, and no multiple of a block of the hand-out
        r0, n0 = np.tile(r0, (2, 1))[:99900].copy(), np.tile(n0, (2, 1))[:99900].copy()
    for flavour in ("exact", "tolerance"):
        hip.set_numerics(flavour)
        dt = DeviceTrace(p, r0, n0)
        ref = None
        for it in range(N):
            dt.launch(); torch.cuda.synchronize()
            assert int(dt.npoints.min()) >= 1
            sig = tuple(int(getattr(dt, k).view(torch.int64).sum().item()) if getattr(dt, k).dtype == torch.float64
                        else int(getattr(dt, k).to(torch.int64).sum().item())
                        for k in ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals"))
            if ref is None: ref = sig
            assert sig == ref, (cfg, flavour, it, sig, ref)
        print(f"{os.path.basename(cfg)} x{scale} {flavour}: {len(r0)} rays, {N} launches, identical sums of every array's bits  {hip.kernel_name(p, len(r0))}", flush=True)
        del dt
hip.set_numerics("exact")
