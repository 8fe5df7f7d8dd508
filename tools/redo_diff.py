import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from rays_amd import hip
from tests import oracle_lib
nml,p,r0,n0 = bench.build_fan("configs/cfg3b_solovev64k_rk4.in",1,1,None)
rays=np.arange(0,65536,256)
o=oracle_lib.trace(p,r0[rays],n0[rays],nthreads=16)
v0=[];v1=[];s0=[]
ds=float(p.ds)
for i in range(len(rays)):
    n=int(o["npoints"][i])
    if n<2: continue
    v0.append(o["ray_vec"][i,:n-1]); v1.append(o["ray_vec"][i,1:n]); s0.append(np.concatenate([[0.0],np.cumsum(np.full(n,ds))])[:n-1])
v0=np.concatenate(v0);v1=np.concatenate(v1);s0=np.concatenate(s0)
v0=v0[:65536];v1=v1[:65536];s0=s0[:65536]
hip.set_numerics("exact"); e,_,_=hip.ode_step(p,v0,s0)
print("exact == oracle:", np.array_equal(e,v1))
hip.set_numerics("tolerance"); t,_,_=hip.ode_step(p,v0,s0)
print("kernel", hip.kernel_name(p,65536))
d=(t!=v1)
print("states with any differing component:", int(d.any(axis=1).sum()), "of", len(v0))
print("differing per component:", d.sum(axis=0))
rel=np.abs(t-v1)/np.maximum(np.abs(v1),1e-300)
print("max rel per component:", rel.max(axis=0))
print("median rel per component (differing only):", [float(np.median(rel[d[:,c],c])) if d[:,c].any() else 0 for c in range(7)])
