"""How many steps do the waves of the 64k fan spend with <= 8 live lanes?  (developer measurement)"""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rays_amd.trace import DeviceTrace, RaysRun  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3b_solovev64k_rk4.in"
run = RaysRun.from_namelist(os.path.join(ROOT, "configs", cfg))
tr = DeviceTrace(run.params, run.rvec0, run.rindex_vec0)
tr.launch()
npt = tr.results().npoints.astype(np.int64) - 1
w = np.sort(npt.reshape(-1, 64), axis=1)[:, ::-1]          # per wave, longest first
wmax, w9 = w[:, 0], w[:, 8]
slow = wmax - w9                                             # steps with <= 8 live lanes
cost = wmax + 0.64 * slow                                    # in units of a full-rate step
print("waves", len(w), "longest", wmax.max(), "mean wave max", wmax.mean())
print("steps with <= 8 live lanes: mean", slow.mean(), "max", slow.max())
i = int(np.argmax(cost))
print("critical wave", i, "max", wmax[i], "9th", w9[i], "cost", cost[i], "vs longest ray", wmax.max(), "ratio", cost[i] / wmax.max())
top = np.argsort(-wmax)[:8]
for j in top:
    print(" wave", j, "sorted head", w[j, :12].tolist())
np.save(os.path.join(ROOT, "gpurun_out", "npoints_" + cfg + ".npy"), npt)
