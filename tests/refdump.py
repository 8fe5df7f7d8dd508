"""Reader for the raw dumps written by oracle/ref_dump_driver.f90 (test infrastructure)."""
from __future__ import annotations

import numpy as np

MAGIC = 0x52415953


def read_dump(path: str) -> dict:
    b = open(path, "rb").read()
    hdr = np.frombuffer(b, dtype="<i4", count=8)
    assert int(hdr[0]) == MAGIC and int(hdr[1]) == 2, hdr
    nray, nv, nstep_max, nspec, probe_stride = (int(x) for x in hdr[2:7])
    off = 32
    out = dict(nray=nray, nv=nv, nstep_max=nstep_max, nspec=nspec)

    def take(dtype, count, shape=None):
        nonlocal off
        a = np.frombuffer(b, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a.reshape(shape) if shape else a

    sc = take("<f8", 8)
    for k, v in zip(("omgrf", "k0", "clight", "eps0", "ds", "s_max", "dispersion_resid_limit",
                     "trace_wall_s"), sc):
        out[k] = float(v)
    for k in ("qs", "ms", "n0s", "t0s", "eta"):
        out[k] = take("<f8", 6).copy()
    sol = take("<f8", 6)
    out["solovev"] = dict(zip(("rmaj", "kappa", "bphi0", "iota0", "outer_bound", "psiB"),
                              (float(x) for x in sol)))
    out["rvec0"] = take("<f8", 3 * nray, (nray, 3)).copy()
    out["rindex_vec0"] = take("<f8", 3 * nray, (nray, 3)).copy()
    out["npoints"] = take("<i4", nray).copy()
    flags = take("S60", nray)
    out["stop_flag"] = [f.decode().rstrip() for f in flags]
    npt = nstep_max + 1
    out["ray_vec"] = take("<f8", nv * npt * nray, (nray, npt, nv))
    out["residual"] = take("<f8", npt * nray, (nray, npt))
    out["end_ray_vec"] = take("<f8", nv * nray, (nray, nv)).copy()
    if probe_stride > 0:
        nprobe = int(take("<i4", 1)[0])
        neq = 28 + 12 * (nspec + 1)
        rec = np.dtype([("iray", "<i4"), ("j", "<i4"), ("v", "<f8", (nv,)), ("eq", "<f8", (neq,)),
                        ("cold", "<f8", (7,)), ("num", "<f8", (7,)), ("dvds", "<f8", (nv,)),
                        ("resid", "<f8")])
        out["probes"] = np.frombuffer(b, dtype=rec, count=nprobe, offset=off).copy()
        off += rec.itemsize * nprobe
    assert off == len(b), (off, len(b))
    return out


def read_axisym_tables(path: str) -> dict:
    """Spline tables written by ref_dump_driver.f90 (RAYS_DUMP_AXISYM)."""
    b = open(path, "rb").read()
    nr, nz, n_rb, n_ne, n_te, n_ti = (int(x) for x in np.frombuffer(b, dtype="<i4", count=6))
    off = 24
    sc = np.frombuffer(b, dtype="<f8", count=6, offset=off)
    off += 48
    out = dict(zip(("box_rmin", "box_rmax", "box_zmin", "box_zmax", "plasma_psi_limit", "psiB"),
                   (float(x) for x in sc)))

    def take(n):
        nonlocal off
        a = np.frombuffer(b, dtype="<f8", count=n, offset=off).copy()
        off += 8 * n
        return a

    if n_rb == -1:   # 'eqdsk_magnetics_lin_interp': eqdsk_utilities_m's dR, dZ, R_grid, Z_grid, Psi(nr, nz), T(nr)
        out["lin_dR"], out["lin_dZ"] = (float(v) for v in take(2))
        out["r_grid"], out["z_grid"] = take(nr), take(nz)
        out["lin_psi"] = take(nr * nz)            # Psi(nr, nz) Fortran order, PSIAXIS subtracted
        out["lin_t"] = take(nr)
    else:
        out["r_grid"], out["z_grid"] = take(nr), take(nz)
        out["psi_fspl"] = take(16 * nr * nz)          # fspl(4,4,nr,nz) Fortran order
        out["rb_grid"], out["rb_fspl"] = take(n_rb), take(4 * n_rb)
    for key, n in (("ne", n_ne), ("te", n_te), ("ti", n_ti)):
        if n:
            out[key + "_grid"], out[key + "_fspl"] = take(n), take(4 * n)
    assert off == len(b), (off, len(b))
    return out


def read_deposition(path: str) -> dict:
    """Deposition profiles written by ref_dump_driver.f90 (RAYS_DUMP_DEPOSITION): the reference's
    post_process_lib/deposition_profiles_m applied to the run's ray_results_m arrays."""
    b = open(path, "rb").read()
    n_prof, n_bins, nray = (int(x) for x in np.frombuffer(b, dtype="<i4", count=3))
    off = 12

    def take(n):
        nonlocal off
        a = np.frombuffer(b, dtype="<f8", count=n, offset=off).copy()
        off += 8 * n
        return a

    out = dict(n_bins=n_bins, power=take(nray), names=[], work=[], profile=[], q_sum=[])
    n_rho = int(np.frombuffer(b, dtype="<i4", count=1, offset=off)[0])
    off += 4
    if n_rho:
        out["rho_grid"], out["rho_fspl"] = take(n_rho), take(4 * n_rho)
    for _ in range(n_prof):
        out["names"].append(b[off:off + 20].decode().strip())
        off += 20
        take(2)  # grid_min, grid_max
        out["work"].append(take(n_bins * nray).reshape(nray, n_bins))   # work(n_bins, nray)
    for _ in range(n_prof):
        out["profile"].append(take(n_bins))
        out["q_sum"].append(float(take(1)[0]))
    assert off == len(b), (off, len(b))
    return out
