"""Ray lengths of the large fans (recorded points per ray) -> gpurun_out/npoints_<config>.npy, for the scheduling model
of tools/refill_model.py (developer measurement)."""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rays_amd.trace import DeviceTrace  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for cfg in sys.argv[1:] or ["cfg5b_axisym256k_rk4_damp.in", "cfg4_slab1M_rk4.in", "cfg5_axisym256k_sg_damp.in"]:
    nml, p, r0, n0 = bench.build_fan(os.path.join(ROOT, "configs", cfg), 1, 1, None)
    tr = DeviceTrace(p, r0, n0)
    tr.launch()
    torch.cuda.synchronize()
    npt = tr.npoints.cpu().numpy().astype(np.int32)
    st = tr.stop_code.cpu().numpy().astype(np.int32)
    print(cfg, "rays", len(npt), "steps", int((npt - 1).clip(0).sum()), "mean", float(npt.mean()), "max", int(npt.max()),
          "npoints==1:", int((npt == 1).sum()), flush=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "npoints_" + cfg + ".npz"), npoints=npt, stop_code=st)
    del tr
