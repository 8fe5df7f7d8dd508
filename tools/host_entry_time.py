"""Wall time of the host-pointer entry rays_hip_trace (what the Fortran drop-in calls) on the 64k fan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rays_amd.trace import RaysRun
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run = RaysRun.from_namelist(os.path.join(root, "configs/cfg3b_solovev64k_rk4.in"))
for i in range(3):
    t0 = time.perf_counter()
    res = run.trace_rays(ngpu=1)
    dt = time.perf_counter() - t0
    print(f"call {i}: {dt*1e3:.1f} ms wall (incl. allocating/zeroing the 4.7 GB of host arrays in Python), "
          f"library-reported {res.elapsed_s*1e3:.1f} ms, {res.total_steps} steps", flush=True)

# the Fortran host's situation: result arrays allocated and zero-filled once, pages resident
from rays_amd import hip
out = hip.trace_host(run.params, run.rvec0, run.rindex_vec0, ngpu=1)
for i in range(3):
    t0 = time.perf_counter()
    out = hip.trace_host(run.params, run.rvec0, run.rindex_vec0, ngpu=1, out=out)
    dt = time.perf_counter() - t0
    print(f"resident arrays, call {i}: {dt*1e3:.1f} ms wall, library-reported {out['elapsed_s']*1e3:.1f} ms", flush=True)

# several slots per device (rays_hip_init_devices): a slot's device-to-host copy overlaps the other slots' traces
ref = {k: out[k].copy() for k in ("ray_vec", "npoints")}
for slots in (2, 4, 8):
    hip.init_devices([0] * slots)
    out = hip.trace_host(run.params, run.rvec0, run.rindex_vec0, ngpu=None, out=out)
    best = 1e9
    for i in range(4):
        t0 = time.perf_counter()
        out = hip.trace_host(run.params, run.rvec0, run.rindex_vec0, ngpu=None, out=out)
        best = min(best, time.perf_counter() - t0)
    same = np.array_equal(out["ray_vec"], ref["ray_vec"]) and np.array_equal(out["npoints"], ref["npoints"])
    print(f"resident arrays, {slots} slots on device 0: best of 4 {best*1e3:.1f} ms wall, identical: {same}", flush=True)
