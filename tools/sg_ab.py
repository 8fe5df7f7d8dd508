"""A/B pass times of the SG configurations (cfg 3: Solovev 64k, finite-difference dD; cfg 5: eqdsk 256k + damping) for the
library in RAYS_HIP_LIB; RAYS_HIP_SG_GROUP selects the mapping of cfg 3.  usage: python tools/sg_ab.py [cfg3] [cfg5]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rays_amd import hip
from rays_amd.trace import DeviceTrace
CFG = {"cfg3": "configs/cfg3_solovev64k_sg_num.in", "cfg5": "configs/cfg5_axisym256k_sg_damp.in"}
print("lib:", os.path.basename(os.environ.get("RAYS_HIP_LIB", "default")), "group:", os.environ.get("RAYS_HIP_SG_GROUP", "-"), flush=True)
for name in (sys.argv[1:] or ["cfg3", "cfg5"]):
    nml, p, r0, n0 = bench.build_fan(CFG[name], 1, 1, None)
    dt = DeviceTrace(p, r0, n0)
    dt.launch(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); dt.launch(zero_fill=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    st = int(np.maximum(dt.npoints.cpu().numpy().astype(np.int64) - 1, 0).sum())
    print(f"  {name} {hip.kernel_name(p, len(r0))}: steps {st} best {min(ts):.2f} mean {np.mean(ts):.2f} ms", flush=True)
    del dt
