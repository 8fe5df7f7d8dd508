"""Per-step deviation of the tolerance flavour on samples of the BASELINE fans: the flavour restarted from every point of
the ORACLE's trajectories (bit-identical to the reference) for one output step, against the oracle's next point;
norm-wise relative error on r and k.  north_star's bar: 1e-10.
usage: python tools/tol_per_step_survey.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rays_amd import hip
from tests import oracle_lib

CASES = [("cfg3b", "configs/cfg3b_solovev64k_rk4.in", 128), ("cfg4", "configs/cfg4_slab1M_rk4.in", 4096),
         ("cfg5b", "configs/cfg5b_axisym256k_rk4_damp.in", 512), ("cfg2", "configs/cfg2_solovev1024_rk4.in", 4)]
hip.set_numerics("tolerance")
for name, cfg, stride in CASES:
    nml, p, r0, n0 = bench.build_fan(cfg, 1, 1, None)
    if bench.build_fan.tables is not None:
        oracle_lib.set_axisym_tables(bench.build_fan.tables)
    sel = np.arange(0, len(r0), stride)
    ora = oracle_lib.trace(p, r0[sel], n0[sel], nthreads=os.cpu_count() or 1)
    v0, v1, s0 = [], [], []
    for r in range(len(sel)):
        n = int(ora["npoints"][r])
        if n < 2:
            continue
        s = np.concatenate([[0.0], np.cumsum(np.full(n - 1, float(p.ds)))])
        v0.append(ora["ray_vec"][r, :n - 1]); v1.append(ora["ray_vec"][r, 1:n]); s0.append(s[:n - 1])
    v0, v1, s0 = np.concatenate(v0), np.concatenate(v1), np.concatenate(s0)
    got, resid, code = hip.ode_step(p, v0, s0)
    ok = code == 0
    rel = lambda sl: np.linalg.norm(got[ok][:, sl] - v1[ok][:, sl], axis=-1) / np.linalg.norm(v1[ok][:, sl], axis=-1)
    er, ek = rel(slice(0, 3)), rel(slice(3, 6))
    print(f"{name}: {hip.kernel_name(p, len(v0))}  {len(sel)} rays, {len(v0)} one-step restarts, {int((~ok).sum())} stopped; "
          f"per-step rel err on r: max {er.max():.2e} median {np.median(er):.1e}; on k: max {ek.max():.2e} median {np.median(ek):.1e}", flush=True)
hip.set_numerics("exact")
