"""Does a wave with few live rays step more slowly?  (developer measurement)
Fan A: every lane carries the longest ray of the 64k Solovev fan.  Fan B(k): k lanes per wave carry it,
the others a ray launched outside the box (stops at its first check, npoints = 1)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rays_amd.trace import DeviceTrace, RaysRun  # noqa: E402

run = RaysRun.from_namelist(os.path.join(ROOT, "configs", "cfg3b_solovev64k_rk4.in"))
p = run.params
tr = DeviceTrace(p, run.rvec0, run.rindex_vec0)
tr.launch()
npt = tr.results().npoints
il = int(np.argmax(npt))
print("longest ray", il, "npoints", npt[il], flush=True)
n = 65536
for k in (64, 32, 16, 15, 12, 8, 4, 1):
    r0 = np.tile(run.rvec0[il], (n, 1))
    n0 = np.tile(run.rindex_vec0[il], (n, 1))
    dead = (np.arange(n) % 64) >= k
    r0[dead, 0] = 10.0
    t = DeviceTrace(p, r0, n0)
    t.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        t.launch(zero_fill=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    res = t.results()
    print(f"live lanes per wave {k:2d}: {ms:7.3f} ms per pass, live npoints {res.npoints[0]}, dead npoints {res.npoints[63] if k < 64 else '-'}", flush=True)
    del t
