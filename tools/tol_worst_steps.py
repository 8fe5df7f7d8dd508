"""Where does the tolerance flavour deviate most in ONE step, and how well conditioned is that step?  cfg 2 fan: the worst
one-step restarts (from the oracle's points), with the EXACT kernel's response to a one-ulp change of the same input
state as the yardstick (if one ulp of input moves the exact result by as much, the reference itself defines the step
only to that accuracy)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rays_amd import hip
from tests import oracle_lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "configs/cfg2_solovev1024_rk4.in"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nml, p, r0, n0 = bench.build_fan(cfg, 1, 1, None)
sel = np.arange(0, len(r0), stride)
ora = oracle_lib.trace(p, r0[sel], n0[sel], nthreads=os.cpu_count() or 1)
v0, v1, s0, who = [], [], [], []
for r in range(len(sel)):
    n = int(ora["npoints"][r])
    if n < 2:
        continue
    s = np.concatenate([[0.0], np.cumsum(np.full(n - 1, float(p.ds)))])
    v0.append(ora["ray_vec"][r, :n - 1]); v1.append(ora["ray_vec"][r, 1:n]); s0.append(s[:n - 1])
    who += [(int(sel[r]), k, n) for k in range(n - 1)]
v0, v1, s0 = np.concatenate(v0), np.concatenate(v1), np.concatenate(s0)
relk = lambda a, b: np.linalg.norm(a[:, 3:6] - b[:, 3:6], axis=-1) / np.linalg.norm(b[:, 3:6], axis=-1)
relr = lambda a, b: np.linalg.norm(a[:, 0:3] - b[:, 0:3], axis=-1) / np.linalg.norm(b[:, 0:3], axis=-1)
hip.set_numerics("exact")
ex, _, cex = hip.ode_step(p, v0, s0)
print("exact kernel vs oracle: max rel err on k", relk(ex, v1).max(), "(bit-identical:", bool(np.array_equal(ex, v1)), ")")
hip.set_numerics("tolerance")
tol, _, ctol = hip.ode_step(p, v0, s0)
ek, er = relk(tol, v1), relr(tol, v1)
print(f"tolerance: {len(v0)} restarts; rel err on k: max {ek.max():.3e}, > 1e-10: {(ek > 1e-10).sum()}, > 1e-11: {(ek > 1e-11).sum()}, > 1e-12: {(ek > 1e-12).sum()}; on r: max {er.max():.3e}")
# conditioning: exact kernel, input moved by one ulp in each of x, kx (relative 2.2e-16)
hip.set_numerics("exact")
sens = np.zeros(len(v0))
for c in (0, 3, 4):
    vp = v0.copy()
    vp[:, c] = np.nextafter(vp[:, c], np.inf)
    e2, _, _ = hip.ode_step(p, vp, s0)
    sens = np.maximum(sens, relk(e2, ex))
worst = np.argsort(-ek)[:12]
print("worst steps (ray, point k of n): tolerance rel err on k | exact kernel's response to ONE ULP of input (max over x, kx, ky)")
for i in worst:
    print(f"  ray {who[i][0]:5d} point {who[i][1]:4d} of {who[i][2]:4d}: {ek[i]:.3e} | {sens[i]:.3e}   ratio {ek[i] / max(sens[i], 1e-300):.2f} ulp-equivalents")
big = ek > 1e-13   # (below that both are a few units of the last place of the OUTPUT)
print(f"over the {big.sum()} restarts with an error above 1e-13: max of (tolerance error / one-ulp response) = "
      f"{(ek[big] / np.maximum(sens[big], 1e-300)).max():.2f}; restarts whose one-ulp response alone exceeds 1e-10: {(sens > 1e-10).sum()}")
