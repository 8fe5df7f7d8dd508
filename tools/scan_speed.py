"""Fused ray_scan (ONE launch, rays_hip_scan_device) vs the reference's serial loop over runs, on the
1024-ray Solovev fan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd.params import copy_params
from rays_amd.scan import RayScan, scan_values
from rays_amd.trace import DeviceTrace
nml, p, r0, n0 = bench.build_fan("configs/cfg2_solovev1024_rk4.in", 1)
for n_runs in (1, 8, 32, 64):
    vals = scan_values("fixed_increment", n_runs, p_start=float(p.ds), p_incr=float(p.ds) / 64)
    scan = RayScan(p, r0, n0, vals)
    scan.launch(); scan.synchronize()
    t0 = time.perf_counter(); scan.launch(zero_fill=False); scan.synchronize(); fused = time.perf_counter() - t0
    runs = []
    for v in vals:
        q = copy_params(p); q.ds = float(v)
        runs.append(DeviceTrace(q, r0, n0))
    for r in runs:
        r.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in runs:
        r.launch(zero_fill=False); torch.cuda.synchronize()
    serial = time.perf_counter() - t0
    steps = int(torch.clamp(scan.npoints.to(torch.int64) - 1, min=0).sum())
    same = all(torch.equal(scan.ray_vec[i], r.ray_vec) for i, r in enumerate(runs))
    print(f"{n_runs:3d} runs x 1024 rays: one launch {fused*1e3:7.2f} ms ({steps/fused:.3e} steps/s)   "
          f"one after another {serial*1e3:7.2f} ms ({steps/serial:.3e} steps/s)   identical: {same}", flush=True)
