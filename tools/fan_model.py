"""Where does the 64k-fan pass go?  npoints stats, per-wave max, time of uniform-length fans."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace

def run(cfg, fan_scale=1, nstep_max=None, reps=5):
    nml, p, r0, n0 = bench.build_fan(cfg, 1, fan_scale, nstep_max)
    dt = DeviceTrace(p, r0, n0)
    dt.launch(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): dt.launch(zero_fill=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    n = dt.npoints.cpu().numpy().astype(np.int64)
    st = np.maximum(n - 1, 0)
    nw = (len(st) + 63) // 64
    pad = np.zeros(nw * 64, np.int64); pad[:len(st)] = st
    wmax = pad.reshape(nw, 64).max(1)
    print(f"{os.path.basename(cfg)} x{fan_scale}: nray={len(st)} steps={st.sum()} mean={st.mean():.1f} max={st.max()} "
          f"wave-max mean={wmax.mean():.1f} | {ms:.3f} ms  -> {st.sum()/ms/1e6:.3f} Gsteps/s; "
          f"us/step(longest)={ms*1e3/st.max():.2f}  lane-util={st.sum()/(wmax.sum()*64):.3f}  "
          f"{hip.kernel_name(p, len(st))}", flush=True)
    return ms, st

if __name__ == "__main__":
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs/cfg3b_solovev64k_rk4.in")
    print("lib:", os.environ.get("RAYS_HIP_LIB", "default"))
    if len(sys.argv) > 1 and sys.argv[1] == "short":
        run(cfg, reps=10)
        run(cfg, 1, 100, reps=10)
        run(cfg, 4, 400)
    else:
        run(cfg)
        for nm in (100, 200, 400):
            run(cfg, 1, nm)
        run(cfg, 4, 400)
        run("configs/cfg1_slab16_rk4.in")
