"""ctypes wrapper around tests/hip_emul/librays_emul.so: the PRODUCT kernel source compiled for the
host (one lane per wave) -- test infrastructure for the CPU tier and for sanitizer runs."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rays_amd.params import AxisymTables, RaysFan, RaysParams, axisym_tables_struct

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "tests", "hip_emul")
_LIB = os.path.join(_DIR, "librays_emul.so")
_lib = None
# second build of the same sources with the SG kernel's fast storage tiers shrunk (rays_sg.hpp:
# coefficient entries <= 2 and 2 + 1 phi rows instead of <= 8 and 4 + 4), so that ordinary rays
# cross every tier boundary
_TIERS_LIB = os.path.join(_DIR, "librays_emul_tiers.so")
_TIERS_DEFS = ["-DRAYS_SG_TIER=2", "-DRAYS_SG_REG_ROWS=2", "-DRAYS_SG_LDS_ROWS=1"]
_tiers_lib = None


def build(sanitize: bool = False, out: str = None, defs=()):
    global _LIB
    _LIB_SAVE = _LIB
    if out is not None:
        _LIB = out
    try:
        _build(sanitize, list(defs))
    finally:
        _LIB = _LIB_SAVE


def _build(sanitize, defs):
    srcs = [os.path.join(_DIR, "emul_trace.cpp"), os.path.join(_DIR, "hip", "hip_runtime.h")]
    srcs += [os.path.join(_ROOT, "rays_amd", "csrc", f) for f in
             ("rays_libm.hpp", "rays_libm_tables.inc", "rays_device.hpp", "rays_device_arith.inc", "rays_trace.hpp", "rays_rk4.hpp", "rays_rk4_body.inc", "rays_rk4_pass.inc", "rays_sg.hpp", "rays_dev_params.inc",
              "rays_ray_init.hpp", "rays_fan_setup.inc", "rays_deposition.hpp")]
    if os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs):
        return
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-extern-tls-init", "-fPIC", "-shared",
           "-w", *defs, "-I", _DIR, srcs[0], "-o", _LIB]
    if sanitize:
        cmd[1:1] = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"]
    subprocess.check_call(cmd)


def _set_zfun(fn):
    """Z-function spline table for damp_fund_ECH: the data file cut from the reference's
    initialize_spline_coeffs (rays_amd/data/zfun_spline_re.npz)."""
    z = np.load(os.path.join(_ROOT, "rays_amd", "data", "zfun_spline_re.npz"))
    f = np.ascontiguousarray(z["fspl_re"], dtype=np.float64)
    fn(f.ctypes.data_as(C.POINTER(C.c_double)), len(f), float(z["x_min"]), float(z["x_max"]))


def _load(path):
    if True:
        _lib = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        _lib.rays_emul_trace.restype = C.c_int
        _lib.rays_emul_trace.argtypes = [C.POINTER(RaysParams), C.c_int, dp, dp, dp, dp, ip, ip, dp, dp, dp]
        _lib.rays_emul_trace_ex.restype = C.c_int
        _lib.rays_emul_trace_ex.argtypes = [C.POINTER(RaysParams), C.c_int, dp, dp, dp, dp, ip, ip, dp, dp, dp,
                                            dp, dp, dp, C.c_int]
        _lib.rays_emul_set_zfun_table.restype = C.c_int
        _lib.rays_emul_set_zfun_table.argtypes = [dp, C.c_int, C.c_double, C.c_double]
        _set_zfun(_lib.rays_emul_set_zfun_table)
    return _lib


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _load(_LIB)
    return _lib


def tiers_lib():
    global _tiers_lib
    if _tiers_lib is None:
        build(out=_TIERS_LIB, defs=_TIERS_DEFS)
        _tiers_lib = _load(_TIERS_LIB)
    return _tiers_lib


def set_axisym_tables(tab: dict, small_tiers: bool = False):
    t, keep = axisym_tables_struct(tab)
    fn = (tiers_lib() if small_tiers else lib()).rays_emul_set_axisym_tables
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(AxisymTables), C.c_int, C.c_double, C.c_double]
    lin = "lin_psi" in tab   # 'eqdsk_magnetics_lin_interp'
    fn(C.byref(t), int(lin), float(tab["lin_dR"]) if lin else 0.0, float(tab["lin_dZ"]) if lin else 0.0)


def _offset_zeros(shape, offset):
    """zeros of `shape` whose first element sits `offset` doubles past a 64-byte boundary"""
    n = int(np.prod(shape))
    raw = np.zeros(n + 16)
    skew = (-(raw.ctypes.data // 8) + offset) % 8
    return raw[skew:skew + n].reshape(shape)


def trace(p: RaysParams, rvec0, rindex_vec0, small_tiers: bool = False, vec_offset: int = 0, res_offset: int = 0) -> dict:
    """vec_offset / res_offset: position of the two trajectory arrays within a 64-byte sector, in doubles
    (the RK4 kernel's PointWindow writes whole sectors of the global address)."""
    rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
    rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
    nray, nv, npt = len(rvec0), p.nv, p.nstep_max + 1
    out = dict(
        ray_vec=_offset_zeros((nray, npt, nv), vec_offset), residual=_offset_zeros((nray, npt), res_offset),
        npoints=np.zeros(nray, dtype=np.int32), stop_code=np.zeros(nray, dtype=np.int32),
        end_ray_vec=np.zeros((nray, nv)), end_residuals=np.zeros(nray), max_residuals=np.zeros(nray))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = (tiers_lib() if small_tiers else lib()).rays_emul_trace(C.byref(p), nray, d(rvec0), d(rindex_vec0), d(out["ray_vec"]),
                               d(out["residual"]), i(out["npoints"]), i(out["stop_code"]),
                               d(out["end_ray_vec"]), d(out["end_residuals"]), d(out["max_residuals"]))
    if rc:
        raise RuntimeError(f"rays_emul_trace rc={rc}")
    return out


def _out(nray, nv, npt):
    return dict(ray_vec=np.zeros((nray, npt, nv)), residual=np.zeros((nray, npt)),
                npoints=np.zeros(nray, dtype=np.int32), stop_code=np.zeros(nray, dtype=np.int32),
                end_ray_vec=np.zeros((nray, nv)), end_residuals=np.zeros(nray), max_residuals=np.zeros(nray))


def _trace_ex(p, nray, rvec0, rindex_vec0, out, v0=None, s0=None, ds_run=None, rays_per_run=0):
    dp = C.POINTER(C.c_double)
    d = lambda a: None if a is None else a.ctypes.data_as(dp)
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = lib().rays_emul_trace_ex(C.byref(p), nray, d(rvec0), d(rindex_vec0), d(out["ray_vec"]), d(out["residual"]),
                                  i(out["npoints"]), i(out["stop_code"]), d(out["end_ray_vec"]),
                                  d(out["end_residuals"]), d(out["max_residuals"]), d(v0), d(s0), d(ds_run),
                                  int(rays_per_run))
    if rc:
        raise RuntimeError(f"rays_emul_trace_ex rc={rc}")


def scan(p: RaysParams, rvec0, rindex_vec0, ds_values) -> dict:
    """The fused scan's launch (rays_hip_scan_device: run = ray // nray, ds per run) through the kernel source on
    the host; arrays carry a leading run dimension."""
    rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
    rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
    ds = np.ascontiguousarray(ds_values, dtype=np.float64)
    nray, R = len(rvec0), len(ds)
    out = _out(R * nray, p.nv, p.nstep_max + 1)
    _trace_ex(p, R * nray, rvec0, rindex_vec0, out, ds_run=ds, rays_per_run=nray)
    return {k: v.reshape((R, nray) + v.shape[1:]) for k, v in out.items()}


def ode_step(p: RaysParams, v0, s0=None):
    """rays_hip_ode_step_device's launch (nstep_max = 1 from the caller's states) on the host:
    (v1, resid, stop_code) with stop_code 0 where the step was taken and kept."""
    from rays_amd.params import copy_params
    v0 = np.ascontiguousarray(v0, dtype=np.float64).reshape(-1, p.nv)
    s0 = None if s0 is None else np.ascontiguousarray(s0, dtype=np.float64)
    q = copy_params(p)
    q.nstep_max = 1
    q.s_max = 1.7976931348623157e308
    n = len(v0)
    out = _out(n, p.nv, 2)
    _trace_ex(q, n, v0, v0, out, v0=v0, s0=s0)
    ok = out["npoints"] == 2
    return (np.where(ok[:, None], out["ray_vec"][:, 1], 0.0), np.where(ok, out["residual"][:, 1], 0.0),
            np.where(ok, 0, out["stop_code"]).astype(np.int32))


def ray_init(p: RaysParams, fan: RaysFan, nray_max: int):
    """The device launcher's per-member function (rays_ray_init.hpp: fan_member) run on the host."""
    n = int(nray_max)
    rvec0, rindex_vec0 = np.zeros((n, 3)), np.zeros((n, 3))
    nray = C.c_int32(0)
    fn = lib().rays_emul_ray_init
    fn.restype = C.c_int
    dp = C.POINTER(C.c_double)
    fn.argtypes = [C.POINTER(RaysParams), C.POINTER(RaysFan), C.c_int, dp, dp, C.POINTER(C.c_int32)]
    rc = fn(C.byref(p), C.byref(fan), n, rvec0.ctypes.data_as(dp), rindex_vec0.ctypes.data_as(dp), C.byref(nray))
    if rc:
        raise RuntimeError(f"rays_emul_ray_init rc={rc}")
    return rvec0[:nray.value].copy(), rindex_vec0[:nray.value].copy()


def deposition(p: RaysParams, which: int, n_bins: int, ray_vec, npoints, power, rho_grid, rho_fspl):
    """rays_deposition.hpp: deposit_ray + the ray-ordered profile sum, run on the host.
    ray_vec[nray][nstep_max+1][nv] (padded reference layout)."""
    nray = len(npoints)
    ray_vec = np.ascontiguousarray(ray_vec, dtype=np.float64)
    npoints = np.ascontiguousarray(npoints, dtype=np.int32)
    power = np.ascontiguousarray(power, dtype=np.float64)
    rho_grid = np.ascontiguousarray(rho_grid, dtype=np.float64)
    rho_fspl = np.ascontiguousarray(rho_fspl, dtype=np.float64)
    work, profile = np.zeros((nray, n_bins)), np.zeros(n_bins)
    fn = lib().rays_emul_deposition
    fn.restype = C.c_int
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    fn.argtypes = [C.POINTER(RaysParams), C.c_int, C.c_int, C.c_int, dp, ip, dp, dp, dp, C.c_int, dp, dp]
    d = lambda a: a.ctypes.data_as(dp)
    rc = fn(C.byref(p), which, n_bins, nray, d(ray_vec), npoints.ctypes.data_as(ip), d(power), d(rho_grid),
            d(rho_fspl), len(rho_grid), d(work), d(profile))
    if rc:
        raise RuntimeError(f"rays_emul_deposition rc={rc}")
    return work, profile
