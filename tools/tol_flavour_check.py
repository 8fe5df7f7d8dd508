"""Tolerance flavour of the cold RK4 kernels against the exact build on full fans: counts, worst per-point
relative deviation, and pass times of both (several passes each, best and mean).
usage: python tools/tol_flavour_check.py [cfgs...]   (default: the four RK4 configs)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace

CASES = {"cfg3b": ("configs/cfg3b_solovev64k_rk4.in", 1, None, 10), "cfg3b_x4": ("configs/cfg3b_solovev64k_rk4.in", 4, 400, 5),
         "cfg2": ("configs/cfg2_solovev1024_rk4.in", 1, None, 10),
         "cfg5b": ("configs/cfg5b_axisym256k_rk4_damp.in", 1, None, 10), "cfg4": ("configs/cfg4_slab1M_rk4.in", 1, None, 3)}
print("lib:", os.environ.get("RAYS_HIP_LIB", "default"), flush=True)
for name in (sys.argv[1:] or ["cfg3b", "cfg2", "cfg3b_x4", "cfg5b", "cfg4"]):
    cfg, scale, nstep, reps = CASES[name]
    nml, p, r0, n0 = bench.build_fan(cfg, 1, scale, nstep)
    res = {}
    for mode in ("exact", "tolerance"):
        hip.set_numerics(mode)
        dt = DeviceTrace(p, r0, n0)
        dt.launch(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = (dt, ts, hip.kernel_name(p, len(r0)))
    (a, ta, ka), (b, tb, kb) = res["exact"], res["tolerance"]
    same_n = bool(torch.equal(a.npoints, b.npoints)); same_c = bool(torch.equal(a.stop_code, b.stop_code))
    nd = int((a.npoints != b.npoints).sum()); cd = int((a.stop_code != b.stop_code).sum())
    # per-point relative deviation, norm-wise on r and k, over points both recorded
    worst = 0.0
    common = torch.minimum(a.npoints, b.npoints)
    live = torch.arange(p.nstep_max + 1, device=a.ray_vec.device)[None, :] < common[:, None]
    chunk = max(1, (1 << 28) // ((p.nstep_max + 1) * p.nv))
    for i in range(0, len(r0), chunk):
        for sl in (slice(0, 3), slice(3, 6)):
            num = torch.linalg.norm(a.ray_vec[i:i + chunk, :, sl] - b.ray_vec[i:i + chunk, :, sl], dim=-1)
            den = torch.linalg.norm(a.ray_vec[i:i + chunk, :, sl], dim=-1)
            m = live[i:i + chunk] & (den > 0)
            if bool(m.any()):
                worst = max(worst, float((num[m] / den[m]).max()))
    st = int(np.maximum(a.npoints.cpu().numpy().astype(np.int64) - 1, 0).sum())
    print(f"{name}: nray={len(r0)} steps={st}\n  exact     {ka}: best {min(ta):.3f} mean {np.mean(ta):.3f} ms\n"
          f"  tolerance {kb}: best {min(tb):.3f} mean {np.mean(tb):.3f} ms  ({min(ta) / min(tb):.3f}x)\n"
          f"  npoints differ on {nd} rays, stop codes on {cd}; worst accumulated rel deviation {worst:.3e}", flush=True)
    del a, b, res
    torch.cuda.empty_cache()
hip.set_numerics("exact")
