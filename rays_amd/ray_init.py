"""Host-side ray launchers: the `ray_init_m` contract (nray, rvec0(3,nray), rindex_vec0(3,nray)).

North-star scope keeps ray initialisation on the host (SURVEY.md 2 #14, row f1 is "next"):
it runs once, O(nray).  These are numpy mirrors of

    simple_slab_ray_init            RAYS_project/RAYS_lib/simple_slab_ray_init_m.f90:59-187
    ray_init_solovev_nphi_ntheta    RAYS_project/RAYS_lib/solovev_ray_init_nphi_ntheta_m.f90:60-198
    solve_n1_vs_n2_n3 / solve_nx_vs_ny_nz_by_bz   dispersion_solvers_m.f90:49-112, 116-160
    solve_cold_n1sq_vs_n3           disp_solve_cold_n1sq_vs_n3.f90:1-90
    RLSDP_cold                      suscep_m.f90:180-219

including their quirks (slab z launch uses dy_launch, :122; nray pre-count ignores n_y/n_z, :108;
evanescent launches are silently dropped).  The launch-point equilibrium (B direction, alpha,
gamma) is evaluated on the host in float64 following the reference's operation order, and the
complex divisions follow compiler-rt's __divdc3 (what flang emits), so the produced fans are
bit-identical to the reference's for the BASELINE configs (checked in tests against golden dumps).
"""
from __future__ import annotations

import math
from typing import Any, Dict, Tuple

import numpy as np

from .params import (EQUILIB, RAY_INIT, WAVE_MODE, ConfigError, RaysFan, RaysParams, SLAB_BY, SLAB_BZ, SLAB_N,
                     SOLOVEV_N, _arr)


# ---- launch-point equilibrium (host, scalar) ----------------------------------------------------
def _pow(x: float, y: float) -> float:
    try:
        return math.pow(x, y)
    except (ValueError, OverflowError):
        return float("nan")


def _host_fields(p: RaysParams, rvec) -> Tuple[int, np.ndarray, np.ndarray, Dict[str, Any]]:
    """(err, bvec[3], ns[0:nspec], extra) -- B and density only (what ray init consumes)."""
    x, y, z = (float(t) for t in rvec)
    n = p.nspec + 1
    ns = np.zeros(n)
    extra: Dict[str, Any] = {}
    if p.equilib_model == EQUILIB["slab"]:
        s = p.slab
        err = 0
        if x < s.xmin or x > s.xmax:
            err = 10
        if y < s.ymin or y > s.ymax:
            err = 11
        if z < s.zmin or z > s.zmax:
            err = 12
        if err:
            return err, np.zeros(3), ns, extra
        b = np.zeros(3)
        if s.by_prof_model == SLAB_BY["constant"]:
            b[1] = s.by0
        elif s.by_prof_model == SLAB_BY["toroid"]:
            b[1] = s.by0 / (1.0 + x / s.rmaj)
        elif s.by_prof_model == SLAB_BY["linear_shear"]:
            b[1] = s.by0 * x / s.LBy_shear_scale
        if s.bz_prof_model == SLAB_BZ["constant"]:
            b[2] = s.bz0
        elif s.bz_prof_model == SLAB_BZ["toroid"]:
            b[2] = s.bz0 / (1.0 + x / s.rmaj)
        elif s.bz_prof_model == SLAB_BZ["linear"]:
            b[2] = s.bz0 * (1.0 + x / s.LBz_scale)
        else:
            b[2] = s.bz0 + s.dBzdx * (x - s.x0)
        for i in range(n):
            if s.dens_prof_model == SLAB_N["constant"]:
                ns[i] = p.n0s[i]
            elif s.dens_prof_model == SLAB_N["linear"]:
                ns[i] = p.n0s[i] * (1.0 + x / s.Ln_scale)
            elif s.dens_prof_model == SLAB_N["linear_2"]:
                ns[i] = p.n0s[i] + s.dndx * p.eta[i] * (x - s.x0)
            elif s.dens_prof_model == SLAB_N["parabolic"]:
                f = 0.0
                if x < 1.0:
                    f = _pow(1.0 - _pow(x, s.alphan2), s.alphan1)
                if f < s.n_min:
                    f = s.n_min
                ns[i] = p.n0s[i] * f
            else:
                t = x / s.rmin
                ns[i] = p.n0s[i] * math.exp(-3.0 * s.alphan1 * (t * t))
        if ns.min() < 0.0:
            err = 13
        return err, b, ns, extra
    # ---- solovev ------------------------------------------------------------------------------
    s = p.solovev
    r = math.sqrt(x * x + y * y)
    err = 0
    if r < s.box_rmin or r > s.box_rmax:
        err = 20
    if z < s.box_zmin or z > s.box_zmax:
        err = 21
    bp0 = s.bphi0 * s.iota0
    rk = s.rmaj * s.kappa
    t1 = r * z / rk
    t2 = r * r - s.rmaj * s.rmaj
    psi = 0.5 * bp0 * (t1 * t1 + (t2 * t2) / (s.rmaj * s.rmaj) / 4.0)
    br = -bp0 * r * z / (rk * rk)
    zz = z / rk
    rr = r / s.rmaj
    bz = bp0 * (zz * zz + 0.5 * (rr * rr - 1.0))
    gradpsi = np.array([x * bz, y * bz, -r * br])
    psiN = psi / s.psiB
    extra.update(psi=psi, gradpsi=gradpsi, psiN=psiN)
    if err:
        return err, np.zeros(3), ns, extra
    bphi = s.bphi0 * s.rmaj / r
    b = np.array([br * x / r - bphi * y / r, br * y / r + bphi * x / r, bz])
    for i in range(n):
        if s.dens_prof_model == SOLOVEV_N["constant"]:
            ns[i] = p.n0s[i]
        elif psiN < 1.0:
            ns[i] = p.n0s[i] * _pow(1.0 - _pow(psiN, s.alphan2), s.alphan1)
    if ns.min() < 0.0:
        err = 13
    return err, b, ns, extra


# ---- axisym_toroid launch-point fields from the host-built spline tables ------------------------------
def _spl_cell(x, xget):
    nx = len(x)
    z = min(max(xget, x[0]), x[-1])
    nxm = nx - 1
    i = int(1 + nxm * (z - x[0]) / (x[-1] - x[0]))
    i = max(1, min(i, nxm))
    if z < x[i - 1]:
        i -= 1
    elif z > x[i]:
        i += 1
    i = max(1, min(i, nxm))
    return i, z - x[i - 1]


def _spl1(grid, fspl, x):
    i, dx = _spl_cell(grid, x)
    c = np.asarray(fspl).reshape(-1, 4)[i - 1]
    return c[0] + dx * (c[1] + dx * (c[2] + dx * c[3]))


def _axisym_fields(p: RaysParams, rvec, tab):
    """B, density and grad(psi) at a launch point (eqdsk_magnetics_spline_interp_m.f90:206-282,
    bcspevfn f / fx / fy), for axisym_toroid ray initialisation."""
    from .params import AXI_N

    a = p.axisym
    x, y, z = (float(t) for t in rvec)
    n = p.nspec + 1
    r = math.sqrt(x * x + y * y)
    tiny = 10.0e-14
    err = 0
    if r < a.box_rmin - tiny or r > a.box_rmax + tiny:
        err = 22
    if z < a.box_zmin - tiny or z > a.box_zmax + tiny:
        err = 23
    if err:
        return err, np.zeros(3), np.zeros(n), {}
    i, dx = _spl_cell(tab["r_grid"], r)
    j, dy = _spl_cell(tab["z_grid"], z)
    nr = len(tab["r_grid"])
    F = np.asarray(tab["psi_fspl"]).reshape(-1, 4, 4)[(i - 1) + nr * (j - 1)].T  # F[a-1][b-1] = f(a,b,i,j)
    psi = F[0][0] + dy * (F[0][1] + dy * (F[0][2] + dy * F[0][3])) + \
        dx * (F[1][0] + dy * (F[1][1] + dy * (F[1][2] + dy * F[1][3])) +
              dx * (F[2][0] + dy * (F[2][1] + dy * (F[2][2] + dy * F[2][3])) +
                    dx * (F[3][0] + dy * (F[3][1] + dy * (F[3][2] + dy * F[3][3])))))
    psi_r = F[1][0] + dy * (F[1][1] + dy * (F[1][2] + dy * F[1][3])) + \
        2.0 * dx * (F[2][0] + dy * (F[2][1] + dy * (F[2][2] + dy * F[2][3])) +
                    1.5 * dx * (F[3][0] + dy * (F[3][1] + dy * (F[3][2] + dy * F[3][3]))))
    psi_z = F[0][1] + dy * (2.0 * F[0][2] + dy * 3.0 * F[0][3]) + \
        dx * (F[1][1] + dy * (2.0 * F[1][2] + dy * 3.0 * F[1][3]) +
              dx * (F[2][1] + dy * (2.0 * F[2][2] + dy * 3.0 * F[2][3]) +
                    dx * (F[3][1] + dy * (2.0 * F[3][2] + dy * 3.0 * F[3][3]))))
    rbphi = _spl1(tab["rb_grid"], tab["rb_fspl"], r)
    br, bz, bphi = psi_z / r, -psi_r / r, rbphi / r
    gradpsi = np.array([-x * bz, -y * bz, r * br])
    psiN = psi / a.psiB
    b = np.array([br * x / r - bphi * y / r, br * y / r + bphi * x / r, bz])
    if psiN > a.plasma_psi_limit:
        err = 24
    ns = np.zeros(n)
    if a.density_prof_model == AXI_N["constant"]:
        dens = 1.0
    elif a.density_prof_model == AXI_N["parabolic"]:
        dens = 0.0
        if psiN < 1.0:
            dens = _pow(1.0 - _pow(psiN, a.alphan2), a.alphan1)
        dens = max(dens, a.d_scrape_off) if dens < a.d_scrape_off else dens
    else:
        dens = _spl1(tab["ne_grid"], tab["ne_fspl"], psiN) if psiN <= 1.0 else 0.0
        if dens < a.d_scrape_off:
            dens = a.d_scrape_off
    for k in range(n):
        ns[k] = p.n0s[k] if a.density_prof_model == AXI_N["constant"] else p.n0s[k] * dens
    if ns.min() < 0.0:
        err = 13
    return err, b, ns, dict(gradpsi=gradpsi, psiN=psiN)


def _launch_eq(p: RaysParams, rvec, tab=None):
    """bunit, alpha(0:nspec), gamma(0:nspec) as `equilibrium` would give (equilibrium_m.f90:238-265)."""
    if p.equilib_model == EQUILIB["axisym_toroid"]:
        err, b, ns, extra = _axisym_fields(p, rvec, tab)
    else:
        err, b, ns, extra = _host_fields(p, rvec)
    if err:
        return err, None, None, None, extra
    bmag = math.sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2])
    bunit = b / bmag
    n = p.nspec + 1
    alpha, gamma = np.zeros(n), np.zeros(n)
    for i in range(n):
        omgc = p.qs[i] * bmag / p.ms[i]
        omgp2 = ns[i] * (p.qs[i] * p.qs[i]) / (p.eps0 * p.ms[i])
        alpha[i] = omgp2 / (p.omgrf * p.omgrf)
        gamma[i] = omgc / p.omgrf
    return 0, bunit, alpha, gamma, extra


# ---- cold dispersion root for the launch index ---------------------------------------------------
def _divdc3_real(a, c):
    """Re[(a + 0i)/(c + 0i)] exactly as compiler-rt's __divdc3 computes it (vectorised)."""
    a = np.asarray(a, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        _, e = np.frexp(np.abs(c))            # |c| = m * 2**e, m in [0.5, 1)
        k = np.where((c != 0) & np.isfinite(c), e - 1, 0)   # ilogb
        cs = np.ldexp(c, -k)
        denom = cs * cs + 0.0
        return np.ldexp((a * cs + 0.0) / denom, -k)


def _rlsdp_cold(alpha, gamma):
    R = L = P = 0.0
    for a, g in zip(alpha, gamma):
        R = R - a / (1.0 + g)
        L = L - a / (1.0 - g)
        P = P - a
    R, L, P = 1.0 + R, 1.0 + L, 1.0 + P
    return (R + L) / 2.0, (R - L) / 2.0, P, R, L


def solve_n1_vs_n2_n3(alpha, gamma, wave_mode: str, k_sign: int, n2, n3):
    """Real n1 (NaN where evanescent) -- dispersion_solvers_m.f90:49-112 + disp_solve_cold_n1sq_vs_n3."""
    modes = {"plus": 0, "minus": 1, "fast": 2, "slow": 3}
    if wave_mode.strip() not in modes:
        raise ConfigError(f"solve_disp: improper wave_mode = {wave_mode!r}")
    S, D, P, R, L = _rlsdp_cold(alpha, gamma)
    n2 = np.asarray(n2, dtype=np.float64)
    n3 = np.asarray(n3, dtype=np.float64)
    n3sq = n3 * n3
    a = S
    b = -R * L - P * S + n3sq * (P + S)
    c = P * (n3sq - R) * (n3sq - L)
    discr = b * b - 4.0 * a * c
    ok = discr >= 0.0
    with np.errstate(invalid="ignore", divide="ignore"):
        sd = np.sqrt(np.where(ok, discr, np.nan))
        neg = np.copysign(1.0, b) < 0.0
        # sgn_b < 0: plus=(-b+sd)/(2a), minus=2c/(-b+sd) ; else minus=(-b-sd)/(2a), plus=2c/(-b-sd)
        t = np.where(neg, -b + sd, -b - sd)
        big = _divdc3_real(t, 2.0 * a)
        small = _divdc3_real(2.0 * c, t)
        plus = np.where(neg, big, small)
        minus = np.where(neg, small, big)
        fast = np.where(np.abs(plus) <= np.abs(minus), plus, minus)
        slow = np.where(np.abs(plus) <= np.abs(minus), minus, plus)
        nperp_sq = [plus, minus, fast, slow][modes[wave_mode.strip()]]
        arg = nperp_sq - n2 * n2
        n1 = np.where(ok & (arg >= 0.0), float(k_sign) * np.sqrt(np.abs(arg)), np.nan)
    return n1


# ---- fans ---------------------------------------------------------------------------------------
def simple_slab_ray_init(p: RaysParams, nml: Dict[str, Dict[str, Any]]):
    g = nml.get("simple_slab_ray_init_list", {})
    rf = nml.get("rf_list", {})
    nray_max = int(nml.get("ray_init_list", {}).get("nray_max", 0))
    gi = lambda k, d=0: int(g.get(k, d))
    gf = lambda k, d=0.0: float(g.get(k, d))
    n_x, n_y, n_z = gi("n_x_launch", 1), gi("n_y_launch", 1), gi("n_z_launch", 1)
    n_ky, n_kz = gi("n_ky_launch"), gi("n_kz_launch")
    nray = n_x * n_ky * n_kz  # :108 (ignores n_y_launch, n_z_launch)
    if not (0 < nray <= nray_max):
        raise ConfigError(f"simple slab ray init: improper number of rays  nray={nray}")
    rv, nv = [], []
    for iz in range(n_z):
        z = gf("z_launch0") + iz * gf("dy_launch")  # :122 uses dy_launch
        for iy in range(n_y):
            y = gf("y_launch0") + iy * gf("dy_launch")
            for ix in range(n_x):
                x = gf("x_launch0") + ix * gf("dx_launch")
                rvec = np.array([x, y, z])
                err, bunit, alpha, gamma, _ = _launch_eq(p, rvec)
                if err:
                    continue
                iky = np.arange(n_ky, dtype=np.float64)[:, None]
                ikz = np.arange(n_kz, dtype=np.float64)[None, :]
                ny = np.broadcast_to(gf("rindex_y0") + iky * gf("delta_rindex_y0"), (n_ky, n_kz))
                nz = np.broadcast_to(gf("rindex_z0") + ikz * gf("delta_rindex_z0"), (n_ky, n_kz))
                n2 = ny * bunit[2] - nz * bunit[1]
                n3 = ny * bunit[1] + nz * bunit[2]
                nx = solve_n1_vs_n2_n3(alpha, gamma, str(rf.get("wave_mode", "")),
                                       int(rf.get("k0_sign", 1)), n2, n3)
                keep = ~np.isnan(nx)
                cnt = int(keep.sum())
                rv.append(np.broadcast_to(rvec, (cnt, 3)))
                nv.append(np.stack([nx[keep], ny[keep], nz[keep]], axis=1))
    if not rv or sum(len(a) for a in rv) == 0:
        raise ConfigError("No successful ray initializations")
    rvec0 = np.ascontiguousarray(np.concatenate(rv, axis=0))
    rindex_vec0 = np.ascontiguousarray(np.concatenate(nv, axis=0))
    n = len(rvec0)
    ray_pwr_wt = np.full(n, 1.0) / n / n  # :179,182 divided by nray twice
    return rvec0, rindex_vec0, ray_pwr_wt


def ray_init_solovev_nphi_ntheta(p: RaysParams, nml: Dict[str, Dict[str, Any]]):
    g = nml.get("solovev_ray_init_nphi_ktheta_list", {})
    rf = nml.get("rf_list", {})
    nray_max = int(nml.get("ray_init_list", {}).get("nray_max", 0))
    gi = lambda k, d=0: int(g.get(k, d))
    gf = lambda k, d=0.0: float(g.get(k, d))
    n_r, n_th = gi("n_r_launch"), gi("n_theta_launch")
    n_nt, n_np = gi("n_rindex_theta"), gi("n_rindex_phi")
    nray = n_r * n_th * n_nt * n_np
    if not (0 < nray <= nray_max):
        raise ConfigError(f"solovev ray init: improper number of rays  nray={nray}")
    s = p.solovev
    rv, nv = [], []
    marks = []  # ray count at the end of each r-launch loop (:196)
    for ir in range(n_r):
        for ith in range(n_th):
            theta = gf("theta_launch0") + ith * gf("dtheta_launch")
            rmin_launch = gf("r_launch0") + ir * gf("dr_launch")
            x = s.rmaj + rmin_launch * math.cos(theta)
            z = rmin_launch * math.sin(theta)
            rvec = np.array([x, 0.0, z])
            err, bunit, alpha, gamma, extra = _launch_eq(p, rvec)
            if err:
                continue
            gradpsi = extra["gradpsi"]
            psi_unit = gradpsi / math.sqrt((gradpsi[0] * gradpsi[0] + gradpsi[1] * gradpsi[1])
                                           + gradpsi[2] * gradpsi[2])
            phi_unit = np.array([0.0, 1.0, 0.0])
            theta_unit = np.array([-gradpsi[2], 0.0, gradpsi[0]])
            theta_unit = theta_unit / math.sqrt((theta_unit[0] * theta_unit[0] + 0.0)
                                                + theta_unit[2] * theta_unit[2])
            trans_unit = np.array([bunit[1] * psi_unit[2] - bunit[2] * psi_unit[1],
                                   bunit[2] * psi_unit[0] - bunit[0] * psi_unit[2],
                                   bunit[0] * psi_unit[1] - bunit[1] * psi_unit[0]])
            i_nt = np.arange(n_nt, dtype=np.float64)[:, None]
            i_np = np.arange(n_np, dtype=np.float64)[None, :]
            r_th = np.broadcast_to(gf("rindex_theta0") + i_nt * gf("delta_rindex_theta"), (n_nt, n_np))
            r_ph = np.broadcast_to(gf("rindex_phi0") + i_np * gf("delta_rindex_phi"), (n_nt, n_np))
            rindex = r_ph[..., None] * phi_unit + r_th[..., None] * theta_unit
            n3 = (bunit[0] * rindex[..., 0] + bunit[1] * rindex[..., 1]) + bunit[2] * rindex[..., 2]
            n2 = (trans_unit[0] * rindex[..., 0] + trans_unit[1] * rindex[..., 1]) \
                + trans_unit[2] * rindex[..., 2]
            npsi = solve_n1_vs_n2_n3(alpha, gamma, str(rf.get("wave_mode", "")),
                                     int(rf.get("k0_sign", 1)), n2, n3)
            keep = ~np.isnan(npsi)
            cnt = int(keep.sum())
            rv.append(np.broadcast_to(rvec, (cnt, 3)))
            nv.append(rindex[keep] - npsi[keep][:, None] * psi_unit)
        marks.append(sum(len(a) for a in rv))
    if not rv or sum(len(a) for a in rv) == 0:
        raise ConfigError("No successful ray initializations")
    rvec0 = np.ascontiguousarray(np.concatenate(rv, axis=0))
    rindex_vec0 = np.ascontiguousarray(np.concatenate(nv, axis=0))
    # the reference sets only ray_pwr_wt(count) = 1. after each r-launch loop (:196) and leaves the
    # other entries of the freshly allocated array unset (zero here)
    ray_pwr_wt = np.zeros(len(rvec0))
    for c in marks:
        if c >= 1:
            ray_pwr_wt[c - 1] = 1.0
    return rvec0, rindex_vec0, ray_pwr_wt


def ray_init_axisym_toroid_R_Z_nphi_ntheta(p: RaysParams, nml: Dict[str, Dict[str, Any]], tab):
    """axisym_toroid_ray_init_R_Z_nphi_ntheta_m.f90:67-244 (same construction as the Solovev fan,
    launch point given directly as R, Z; the launch position is NOT stepped: :143-147)."""
    g = nml.get("axisym_toroid_ray_init_r_z_nphi_ntheta_list", {})
    rf = nml.get("rf_list", {})
    nray_max = int(nml.get("ray_init_list", {}).get("nray_max", 0))
    gi = lambda k, d=1: int(g.get(k, d))
    gf = lambda k, d=0.0: float(g.get(k, d))
    n_R, n_Z = gi("n_r_launch"), gi("n_z_launch")
    n_nt, n_np = gi("n_rindex_theta"), gi("n_rindex_phi")
    nray = n_R * n_Z * n_nt * n_np
    if not (0 < nray <= nray_max):
        raise ConfigError(f"axisym_toroid ray init: improper number of rays  nray={nray}")
    rv, nv = [], []
    for _ in range(n_R * n_Z):
        rvec = np.array([gf("r_launch0"), 0.0, gf("z_launch0")])
        err, bunit, alpha, gamma, extra = _launch_eq(p, rvec, tab)
        if err:
            continue
        gradpsi = extra["gradpsi"]
        psi_unit = gradpsi / math.sqrt((gradpsi[0] * gradpsi[0] + gradpsi[1] * gradpsi[1])
                                       + gradpsi[2] * gradpsi[2])
        phi_unit = np.array([0.0, 1.0, 0.0])
        theta_unit = np.array([-gradpsi[2], 0.0, gradpsi[0]])
        theta_unit = theta_unit / math.sqrt((theta_unit[0] * theta_unit[0] + 0.0) + theta_unit[2] * theta_unit[2])
        trans_unit = np.array([bunit[1] * psi_unit[2] - bunit[2] * psi_unit[1],
                               bunit[2] * psi_unit[0] - bunit[0] * psi_unit[2],
                               bunit[0] * psi_unit[1] - bunit[1] * psi_unit[0]])
        i_nt = np.arange(n_nt, dtype=np.float64)[:, None]
        i_np = np.arange(n_np, dtype=np.float64)[None, :]
        r_th = np.broadcast_to(gf("rindex_theta0") + i_nt * gf("delta_rindex_theta"), (n_nt, n_np))
        r_ph = np.broadcast_to(gf("rindex_phi0") + i_np * gf("delta_rindex_phi"), (n_nt, n_np))
        rindex = r_ph[..., None] * phi_unit + r_th[..., None] * theta_unit
        n3 = (bunit[0] * rindex[..., 0] + bunit[1] * rindex[..., 1]) + bunit[2] * rindex[..., 2]
        n2 = (trans_unit[0] * rindex[..., 0] + trans_unit[1] * rindex[..., 1]) + trans_unit[2] * rindex[..., 2]
        npsi = solve_n1_vs_n2_n3(alpha, gamma, str(rf.get("wave_mode", "")), int(rf.get("k0_sign", 1)), n2, n3)
        keep = ~np.isnan(npsi)
        rv.append(np.broadcast_to(rvec, (int(keep.sum()), 3)))
        nv.append(rindex[keep] - npsi[keep][:, None] * psi_unit)
    if not rv or sum(len(a) for a in rv) == 0:
        raise ConfigError("No successful ray initializations")
    rvec0 = np.ascontiguousarray(np.concatenate(rv, axis=0))
    rindex_vec0 = np.ascontiguousarray(np.concatenate(nv, axis=0))
    n = len(rvec0)
    return rvec0, rindex_vec0, np.full(n, 1.0) / n


# ---- single rays given by position and direction -------------------------------------------------
def solve_cold_nsq_vs_theta(alpha, gamma, theta: float):
    """nsq(1:4) = plus, minus, fast, slow roots of the cold dispersion relation at angle theta to B
    (disp_solve_cold_nsq_vs_theta.f90:1-71; real arithmetic).  None where the reference returns with nsq unset."""
    S, D, P, R, L = _rlsdp_cold(alpha, gamma)
    c = math.cos(theta)
    cos2 = c * c
    sin2 = 1.0 - cos2
    a = S * sin2 + P * cos2
    b = -(R * L * sin2) - P * S * (1.0 + cos2)
    cc = P * R * L
    discr = b * b - 4.0 * a * cc
    if discr < 0.0:
        return None
    sd = math.sqrt(discr)
    if math.copysign(1.0, b) < 0.0:
        plus = (-b + sd) / (2.0 * a)
        minus = 2.0 * cc / (-b + sd)
    else:
        minus = (-b - sd) / (2.0 * a)
        plus = 2.0 * cc / (-b - sd)
    fast, slow = (plus, minus) if abs(plus) <= abs(minus) else (minus, plus)
    return plus, minus, fast, slow


def ray_init_XYZ_k_direction(p: RaysParams, rf: Dict[str, Any], rvec, nvec, tab=None):
    """one_ray_init_XYZ_k_direction_m.f90:116-160: the refractive index vector of the requested mode along the
    direction `nvec` at `rvec`; (None, err) when the equilibrium refuses the point.  The reference takes the REAL
    square root of nsq (dispersion_solvers_m.f90:221), so an evanescent direction yields NaNs, not a dropped ray.
    acos / cos are the host libm's, as in the reference."""
    modes = {"plus": 0, "minus": 1, "fast": 2, "slow": 3}
    wm = str(rf.get("wave_mode", "")).strip()
    if wm not in modes:
        raise ConfigError(f"solve_disp: improper wave_mode = {wm!r}")
    err, bunit, alpha, gamma, _ = _launch_eq(p, np.asarray(rvec, dtype=np.float64), tab)
    if err:
        return None, err
    n = np.asarray(nvec, dtype=np.float64)
    n = n / math.sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])
    cos_theta = (bunit[0] * n[0] + bunit[1] * n[1]) + bunit[2] * n[2]
    theta = math.acos(cos_theta)
    nsq = solve_cold_nsq_vs_theta(alpha, gamma, theta)
    if nsq is None:
        raise ConfigError("disp_solve_cold_nsq_vs_theta: evanescent root (nsq undefined in the reference)")
    v = nsq[modes[wm]]
    n_re = float(int(rf.get("k0_sign", 1))) * (math.sqrt(v) if v >= 0.0 else float("nan"))
    return n_re * n, 0


def one_ray_init_XYZ_n_direction(p: RaysParams, nml: Dict[str, Dict[str, Any]], tab=None):
    """one_ray_init_XYZ_k_direction_m.f90:28-114: one ray from /one_ray_init_XYZ_k_direction_list/."""
    g = nml.get("one_ray_init_xyz_k_direction_list", {})
    rvec = np.array([float(g.get(k, 0.0)) for k in ("x", "y", "z")])
    nvec = np.array([float(g.get(k, 0.0)) for k in ("nx", "ny", "nz")])
    if int(nml.get("ray_init_list", {}).get("nray_max", 1)) < 1:
        raise ConfigError("one_ray_init_XYZ_n_direction: improper number of rays  nray= 1")
    if not bool(g.get("use_this_n_vec", False)):
        nvec, err = ray_init_XYZ_k_direction(p, nml.get("rf_list", {}), rvec, nvec, tab)
        if err:
            raise ConfigError("No successful ray initializations")
    return rvec[None, :].copy(), np.asarray(nvec)[None, :].copy(), np.ones(1)


def file_input_ray_init(p: RaysParams, nml: Dict[str, Dict[str, Any]], base_dir: str, tab=None):
    """file_input_ray_init_m.f90:34-143: positions and directions from ray_init_<run_label>.in; rays the equilibrium
    refuses are dropped.  (ray_pwr_wt: the reference scales a work array it never fills, :131 -- zeros -- unless
    every ray_pwr_wt_in is zero, :128-129.)"""
    import os

    from .namelist import read_namelist

    label = str(nml.get("diagnostics_list", {}).get("run_label", "")).strip()
    nray_max = int(nml.get("ray_init_list", {}).get("nray_max", 1))
    g = read_namelist(os.path.join(base_dir, f"ray_init_{label}.in")).get("file_input_ray_init_list", {})
    n_in = int(g.get("n_rays_in", 0))
    if n_in < 1 or n_in > nray_max:
        raise ConfigError(f"file_input_ray_init: improper number of rays  n_rays_in= {n_in}")
    r_in = np.array(_arr(g.get("rvec_in"), 3 * nray_max, 0.0), dtype=np.float64).reshape(nray_max, 3)
    n_in_vec = np.array(_arr(g.get("rindex_vec_in"), 3 * nray_max, 0.0), dtype=np.float64).reshape(nray_max, 3)
    w_in = np.array(_arr(g.get("ray_pwr_wt_in"), nray_max, 1.0), dtype=np.float64)
    rv, nv = [], []
    failed = False   # the reference's `init_err` is a saved variable that is never cleared (:33, :99-107): once a
    for i in range(n_in):   # ray has been refused, every later ray of the file is dropped as well
        n, err = ray_init_XYZ_k_direction(p, nml.get("rf_list", {}), r_in[i], n_in_vec[i], tab)
        failed = failed or bool(err)
        if failed:
            continue
        rv.append(r_in[i].copy())
        nv.append(n)
    if not rv:
        raise ConfigError("No successful ray initializations")
    nray = len(rv)
    w = np.full(nray, 1.0 / nray) if w_in.max() == 0.0 else np.zeros(nray)
    return np.ascontiguousarray(rv), np.ascontiguousarray(nv), w


def initialize_ray_init(p: RaysParams, nml: Dict[str, Dict[str, Any]], axisym_tables=None, base_dir: str = "."):
    """ray_init_m.f90:72-127 dispatch on ray_init_model (`base_dir`: where a file_input_ray_init run finds its
    ray_init_<run_label>.in -- the run directory)."""
    model = str(nml.get("ray_init_list", {}).get("ray_init_model", "")).strip()
    if model in ("one_ray_init_XYZ_n_direction", "file_input_ray_init"):
        if p.equilib_model == EQUILIB["axisym_toroid"] and p.axisym.magnetics_model != 0:
            raise ConfigError(f"{model}: the Python host mirror has no axisym_toroid fields for this magnetics_model")
        if model == "file_input_ray_init":
            return file_input_ray_init(p, nml, base_dir, axisym_tables)
        return one_ray_init_XYZ_n_direction(p, nml, axisym_tables)
    if model == "simple_slab":
        return simple_slab_ray_init(p, nml)
    if model == "solovev":
        return ray_init_solovev_nphi_ntheta(p, nml)
    if model == "axisym_toroid_ray_init_R_Z_nphi_ntheta":
        if p.axisym.magnetics_model != 0:   # 'solovev_magnetics', 'eqdsk_magnetics_lin_interp': no host mirror; the device launcher is the product's
            from . import hip
            fan, nray_max = fan_from_namelist(nml)
            return hip.ray_init_host(p, fan, nray_max)
        if axisym_tables is None:
            raise ConfigError("axisym_toroid ray init needs the host-built spline tables")
        return ray_init_axisym_toroid_R_Z_nphi_ntheta(p, nml, axisym_tables)
    raise ConfigError(f"initialize_ray_init: invalid ray_init_model = {model!r}")


def fan_from_namelist(nml: Dict[str, Dict[str, Any]]) -> Tuple[RaysFan, int]:
    """(rays_fan_t, nray_max) for the device-side launcher (rays_hip_ray_init): the launcher's
    namelist group + wave_mode / k0_sign of /rf_list/, defaults as in the reference modules."""
    model = str(nml.get("ray_init_list", {}).get("ray_init_model", "")).strip()
    if model not in RAY_INIT:
        raise ConfigError(f"initialize_ray_init: invalid ray_init_model = {model!r}")
    rf = nml.get("rf_list", {})
    wm = str(rf.get("wave_mode", "")).strip()
    if wm not in WAVE_MODE:
        raise ConfigError(f"solve_disp: improper wave_mode = {wm!r}")
    f = RaysFan()
    f.model, f.wave_mode, f.k0_sign = RAY_INIT[model], WAVE_MODE[wm], int(rf.get("k0_sign", 1))
    nray_max = int(nml.get("ray_init_list", {}).get("nray_max", 0))
    if model == "simple_slab":
        g = nml.get("simple_slab_ray_init_list", {})
        for k in ("n_x_launch", "n_y_launch", "n_z_launch"):
            setattr(f, k, int(g.get(k, 1)))
        for k in ("n_ky_launch", "n_kz_launch"):
            setattr(f, k, int(g.get(k, 0)))
        for k in ("x_launch0", "dx_launch", "y_launch0", "dy_launch", "rindex_y0", "delta_rindex_y0",
                  "rindex_z0", "delta_rindex_z0"):
            setattr(f, k, float(g.get(k, 0.0)))
        f.slab_z_launch0 = float(g.get("z_launch0", 0.0))
        return f, nray_max
    g = nml.get("solovev_ray_init_nphi_ktheta_list" if model == "solovev"
                else "axisym_toroid_ray_init_r_z_nphi_ntheta_list", {})
    f.n_r_launch = int(g.get("n_r_launch", 1))
    f.n_theta_launch = int(g.get("n_theta_launch" if model == "solovev" else "n_z_launch", 1))
    f.n_rindex_theta, f.n_rindex_phi = int(g.get("n_rindex_theta", 1)), int(g.get("n_rindex_phi", 1))
    for k in ("r_launch0", "dr_launch", "theta_launch0", "dtheta_launch", "z_launch0", "rindex_theta0",
              "delta_rindex_theta", "rindex_phi0", "delta_rindex_phi"):
        setattr(f, k, float(g.get(k, 0.0)))
    return f, nray_max
