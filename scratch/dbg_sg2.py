import sys, numpy as np
sys.path.insert(0,'.')
from rays_amd import hip
from tests.common import load_golden
from tests import oracle_lib
np.set_printoptions(linewidth=200)
for name in ("gold_solovev64_sg_num","gold_solovev64_sg_cold"):
    g,nml,p = load_golden(name)
    out = hip.trace_host(p, g["rvec0_full"], g["rindex_vec0_full"], ngpu=1)
    ora = oracle_lib.trace(p, g["rvec0_full"], g["rindex_vec0_full"])
    d = np.abs(out["ray_vec"]-ora["ray_vec"])
    rel_r = np.linalg.norm(out["ray_vec"][...,:3]-ora["ray_vec"][...,:3],axis=-1)/np.maximum(np.linalg.norm(ora["ray_vec"][...,:3],axis=-1),1e-30)
    rel_k = np.linalg.norm(out["ray_vec"][...,3:6]-ora["ray_vec"][...,3:6],axis=-1)/np.maximum(np.linalg.norm(ora["ray_vec"][...,3:6],axis=-1),1e-30)
    print(name, "bitwise rays:", int((d.max(axis=(1,2))==0).sum()), "/", len(d), "max rel r", rel_r.max(), "max rel k", rel_k.max())
    bad = np.argwhere(rel_k>1e-12)
    if len(bad):
        r0 = bad[0][0]; print(" first differing ray", r0, "first step", bad[bad[:,0]==r0][0], "npoints", ora["npoints"][r0], "stop", ora["stop_code"][r0])
        i = np.argmax(rel_k[r0]); print("  worst step", i, rel_k[r0,i], rel_r[r0,i], out["ray_vec"][r0,i], ora["ray_vec"][r0,i])
        nz = np.nonzero(rel_k[r0]>0)[0]; print("  first nonzero diff step", nz[:5], rel_k[r0,nz[:5]])
