"""A/B of the ray-end / refill handling of the trace kernels: times the configurations whose lanes are refilled (and the
headline fan, which is not) in both numerics flavours with the library RAYS_HIP_LIB names, and prints a checksum of the
results so that two libraries' runs can be compared line by line.  tools/refill_ab.sh runs it for two libraries."""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
print("lib:", os.environ.get("RAYS_HIP_LIB", "default"), flush=True)
CASES = (("configs/cfg3b_solovev64k_rk4.in", 1, None, 10), ("configs/cfg3b_solovev64k_rk4.in", 4, 400, 5),
         ("configs/cfg5b_axisym256k_rk4_damp.in", 1, None, 10), ("configs/cfg4_slab1M_rk4.in", 1, None, 3),
         ("configs/cfg2_solovev1024_rk4.in", 1, None, 10))
if len(sys.argv) > 1 and sys.argv[1] == "sg":
    CASES = (("configs/cfg5_axisym256k_sg_damp.in", 1, None, 3), ("configs/cfg3_solovev64k_sg_num.in", 1, None, 2))
for cfg, scale, nstep, reps in CASES:
    if not os.path.exists(cfg):
        print("missing", cfg); continue
    nml, p, r0, n0 = bench.build_fan(cfg, 1, scale, nstep)
    for flavour, order in (("exact", "index"), ("exact", "pilot"), ("tolerance", "index"), ("tolerance", "pilot")):
        hip.set_numerics(flavour)
        os.environ["RAYS_HIP_RAY_ORDER"] = order  # read by the library at every launch (rays_capi.hip: sched_enabled)
        dt = DeviceTrace(p, r0, n0)
        dt.launch(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        n = dt.npoints.cpu().numpy().astype(np.int64); st = np.maximum(n - 1, 0).sum()
        crc = zlib.crc32(dt.npoints.cpu().numpy().tobytes()) ^ zlib.crc32(dt.stop_code.cpu().numpy().tobytes())
        # every 997th ray's trajectory
        sample = dt.ray_vec[::997].cpu().numpy()
        crc2 = zlib.crc32(sample.tobytes())
        print(f"{os.path.basename(cfg)} x{scale} {flavour:9s} {order:5s} nray={len(n)} steps={st} best {min(ts):.3f} mean {np.mean(ts):.3f} ms  "
              f"{hip.kernel_name(p, len(n))}  counts {crc:08x} traj {crc2:08x}", flush=True)
        del dt
        if hip.kernel_name(p, len(n)).startswith("sg_") and order == "pilot":
            break
hip.set_numerics("exact")
