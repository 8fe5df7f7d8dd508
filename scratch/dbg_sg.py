import sys, numpy as np
sys.path.insert(0,'.')
from rays_amd import hip
from tests.common import load_golden
np.set_printoptions(linewidth=200)
g,nml,p = load_golden("gold_solovev64_sg_cold")
for sel in ([0,1],[0,1,2,3]):
    out = hip.trace_host(p, g["rvec0"][sel], g["rindex_vec0"][sel], ngpu=1)
    print(sel, "npoints", out["npoints"], "stop", out["stop_code"], "ref", g["npoints"][sel])
    print(out["end_ray_vec"])
