"""Full-fan numerics survey of a trace kernel against the CPU oracle (test infrastructure; used by
tests/test_gpu_numerics_full_fans.py and tools/numerics_evidence.py).

Two measurements per fan, both against the oracle's trajectories (bit-identical to the reference CPU path):

  * per step: the HIP path restarted from EVERY recorded oracle point of the selected rays for one output step
    (`rays_hip_ode_step_device` = ode_solver + check_save), norm-wise relative error on r and k of the point it
    lands on against the oracle's next point (north_star: "within 1e-10 relative per step");
  * pointwise: the fan traced as a whole by the kernel the configuration dispatches, every recorded point against the
    oracle's point with the same index (accumulated deviation along the ray), and npoints / stop codes of every ray.

The oracle runs in chunks of rays (its padded arrays are nray x (nstep_max + 1) x nv doubles)."""
from __future__ import annotations

import os

import numpy as np

from rays_amd import hip
from tests import oracle_lib

THRESHOLDS = (1e-10, 1e-11, 1e-12)


def _rel(a, b, sl):
    num = np.linalg.norm(a[:, sl] - b[:, sl], axis=-1)
    den = np.linalg.norm(b[:, sl], axis=-1)
    out = np.zeros(len(a))
    m = den > 0
    out[m] = num[m] / den[m]
    return out


def survey(p, r0, n0, ray_stride=1, chunk_rays=4096, restart_batch=65536, n_worst=8, progress=None, per_step=True):
    """Returns a dict of plain numbers (JSON-ready).  `restart_batch`: states per `ode_step` call -- below two waves per
    SIMD worth of states the one-wave-per-SIMD build of the kernel serves the call, from 131072 on the two-waves build:
    pick the one the fan itself dispatches (`hip.kernel_name(p, len(r0))`).  per_step=False: the pointwise comparison and
    the counts only (Shampine-Gordon fans: an output step there is a whole restarted integration whose tolerances the ray
    carries along, so a restart from a recorded point is not the reference's next step)."""
    import torch
    from rays_amd.trace import DeviceTrace

    nthreads = os.cpu_count() or 1
    sel = np.arange(0, len(r0), ray_stride)
    kernel = hip.kernel_name(p, len(r0))
    step_kernel = hip.kernel_name(p, restart_batch)
    tr = DeviceTrace(p, r0, n0)
    tr.launch()
    torch.cuda.synchronize()
    d_npts, d_codes = tr.npoints.cpu().numpy(), tr.stop_code.cpu().numpy()
    ds = float(p.ds)
    st = dict(steps_restarted=0, restarts_stopped=0, max_per_step=0.0, max_per_step_r=0.0, max_per_step_k=0.0,
              points_compared=0, points_not_identical=0, max_pointwise=0.0, rays_with_other_counts=0, rays_surveyed=int(len(sel)))
    n_step_above = {t: 0 for t in THRESHOLDS}
    n_point_above = {t: 0 for t in THRESHOLDS}
    worst = []   # (err, ray, point, npoints)
    med = []
    for c0 in range(0, len(sel), chunk_rays):
        rays = sel[c0:c0 + chunk_rays]
        ora = oracle_lib.trace(p, r0[rays], n0[rays], nthreads=nthreads)
        npts = ora["npoints"].astype(np.int64)
        st["rays_with_other_counts"] += int(((d_npts[rays] != ora["npoints"]) | (d_codes[rays] != ora["stop_code"])).sum())
        # ---- pointwise: the traced fan against the oracle, point by point ----
        idx = torch.as_tensor(rays, device=tr.ray_vec.device)
        nmax = int(npts.max())
        got = tr.ray_vec.index_select(0, idx)[:, :nmax].cpu().numpy()
        ref = ora["ray_vec"][:, :nmax]
        live = np.arange(nmax)[None, :] < np.minimum(npts, d_npts[rays])[:, None]
        g2, r2 = got[live], ref[live]
        pe = np.maximum(_rel(g2, r2, slice(0, 3)), _rel(g2, r2, slice(3, 6)))
        st["points_compared"] += int(len(pe))
        st["points_not_identical"] += int(((g2 != r2) & ~(np.isnan(g2) & np.isnan(r2))).any(axis=-1).sum())  # all nv components
        if len(pe):
            st["max_pointwise"] = max(st["max_pointwise"], float(pe.max()))
            for t in THRESHOLDS:
                n_point_above[t] += int((pe > t).sum())
        del got, g2, r2
        if not per_step:
            if progress:
                progress(f"  rays {rays[0]}..{rays[-1]}: {st['points_compared']} points so far, max pointwise {st['max_pointwise']:.3e}")
            continue
        # ---- per step: one-step restarts from every oracle point ----
        has = npts >= 2
        cnt = np.where(has, npts - 1, 0)
        tot = int(cnt.sum())
        if tot == 0:
            continue
        first = np.arange(nmax - 1)[None, :] < cnt[:, None]          # [ray][k]: point k has a successor
        v0 = ora["ray_vec"][:, :nmax - 1][first]
        v1 = ora["ray_vec"][:, 1:nmax][first]
        kk = np.broadcast_to(np.arange(nmax - 1)[None, :], first.shape)[first]
        rr = np.broadcast_to(rays[:, None], first.shape)[first]
        # s of point k as trace_rays accumulates it: sout = sout + ds, k times (ray_tracing.f90:118-121)
        s_tab = np.concatenate([[0.0], np.cumsum(np.full(nmax, ds))])
        s0 = s_tab[kk]
        err = np.empty(tot)
        for b0 in range(0, tot, restart_batch):
            b1 = min(tot, b0 + restart_batch)
            g, _, code = hip.ode_step(p, v0[b0:b1], s0[b0:b1])
            ok = code == 0
            st["restarts_stopped"] += int((~ok).sum())
            er, ek = _rel(g, v1[b0:b1], slice(0, 3)), _rel(g, v1[b0:b1], slice(3, 6))
            er[~ok] = 0.0
            ek[~ok] = 0.0
            st["max_per_step_r"] = max(st["max_per_step_r"], float(er.max()))
            st["max_per_step_k"] = max(st["max_per_step_k"], float(ek.max()))
            err[b0:b1] = np.maximum(er, ek)
        st["steps_restarted"] += tot
        st["max_per_step"] = max(st["max_per_step"], float(err.max()))
        for t in THRESHOLDS:
            n_step_above[t] += int((err > t).sum())
        med.append(float(np.median(err)))
        top = np.argsort(-err)[:n_worst]
        nn = npts[np.searchsorted(rays, rr[top])]
        worst += [(float(err[i]), int(rr[i]), int(kk[i]), int(n)) for i, n in zip(top, nn)]
        worst = sorted(worst, reverse=True)[:n_worst]
        if progress:
            progress(f"  rays {rays[0]}..{rays[-1]}: {st['steps_restarted']} restarts so far, max per step {st['max_per_step']:.3e}, "
                     f"max pointwise {st['max_pointwise']:.3e}")
    del tr
    torch.cuda.empty_cache()
    out = dict(st)
    out.update(kernel=kernel, restart_kernel=step_kernel, ray_stride=int(ray_stride), rays_total=int(len(r0)),
               median_per_step=float(np.median(med)) if med else 0.0,
               worst_steps=[dict(rel_err=e, ray=r, point=k, npoints=n) for e, r, k, n in worst])
    for t in THRESHOLDS:
        out[f"n_above_{t:g}"] = n_step_above[t]
        out[f"points_above_{t:g}"] = n_point_above[t]
    out["frac_points_above_1e-10"] = n_point_above[1e-10] / max(1, st["points_compared"])
    return out
