"""Lane-group SG kernel (rays_sg_group.hpp) against the one-ray-per-lane kernel and the oracle on cfg 3:
correctness on a sample, then pass times of both (RAYS_HIP_SG_GROUP=0 selects the old kernel; read per launch).
usage: python tools/sg_group_check.py [nray_side]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
from tests import oracle_lib

cfg = "configs/cfg3_solovev64k_sg_num.in"
nml, p, r0, n0 = bench.build_fan(cfg, 1, 1, None)
print("lib:", os.environ.get("RAYS_HIP_LIB", "default"), "rays", len(r0), flush=True)
res = {}
for mode in ("1", "0"):   # RAYS_HIP_SG_GROUP: 1 = lane groups, 0 = one ray per lane (the default)
    os.environ["RAYS_HIP_SG_GROUP"] = mode
    name = hip.kernel_name(p, len(r0))
    dt = DeviceTrace(p, r0, n0)
    dt.launch(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res[mode] = dt
    st = int(np.maximum(dt.npoints.cpu().numpy().astype(np.int64) - 1, 0).sum())
    print(f"{name}: steps {st} best {min(ts):.2f} mean {np.mean(ts):.2f} ms", flush=True)
a, b = res["1"], res["0"]
for k in ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals"):
    x, y = getattr(a, k), getattr(b, k)
    if x.is_floating_point():
        same = bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())
    else:
        same = bool(torch.equal(x, y))
    print(f"  group vs one-ray-per-lane kernel, {k}: {'identical' if same else 'DIFFERENT'}", flush=True)
sel = np.arange(0, len(r0), 512)
ora = oracle_lib.trace(p, r0[sel], n0[sel], nthreads=os.cpu_count() or 1)
idx = torch.as_tensor(sel, device=a.ray_vec.device)
for k in ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec"):
    x = getattr(a, k).index_select(0, idx).cpu().numpy()
    ok = np.array_equal(x, ora[k], equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, ora[k])
    print(f"  group vs oracle (every 512th ray), {k}: {'identical' if ok else 'DIFFERENT'}", flush=True)
