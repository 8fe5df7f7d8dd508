"""ctypes wrapper around oracle/librays_oracle.so -- the CPU checker (test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rays_amd.params import AxisymTables, RaysParams, axisym_tables_struct

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(_ROOT, "oracle", "librays_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "librays_oracle.so"])


def _set_zfun(fn):
    """Z-function spline table for damp_fund_ECH: the data file cut from the reference's
    initialize_spline_coeffs (rays_amd/data/zfun_spline_re.npz)."""
    z = np.load(os.path.join(_ROOT, "rays_amd", "data", "zfun_spline_re.npz"))
    f = np.ascontiguousarray(z["fspl_re"], dtype=np.float64)
    fn(f.ctypes.data_as(C.POINTER(C.c_double)), len(f), float(z["x_min"]), float(z["x_max"]))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        _lib.rays_oracle_trace.restype = C.c_int
        _lib.rays_oracle_trace.argtypes = [C.POINTER(RaysParams), C.c_int, dp, dp, dp, dp, ip, ip,
                                           dp, dp, dp, C.c_int, C.POINTER(C.c_longlong)]
        _lib.rays_oracle_probe.restype = None
        _lib.rays_oracle_probe.argtypes = [C.POINTER(RaysParams), dp, dp, dp, dp, dp, dp, ip]
        _lib.rays_oracle_set_zfun_table.restype = C.c_int
        _lib.rays_oracle_set_zfun_table.argtypes = [dp, C.c_int, C.c_double, C.c_double]
        _set_zfun(_lib.rays_oracle_set_zfun_table)
        _lib.rays_oracle_set_axisym_tables.restype = C.c_int
        _lib.rays_oracle_set_axisym_tables.argtypes = [C.POINTER(AxisymTables)]
        _lib.rays_oracle_check_params.restype = C.c_int
        _lib.rays_oracle_check_params.argtypes = [C.POINTER(RaysParams)]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def set_axisym_tables(tab: dict):
    t, keep = axisym_tables_struct(tab)
    if "lin_psi" in tab:
        lib().rays_oracle_set_eqdsk_lin_tables(C.byref(t), C.c_double(float(tab["lin_dR"])), C.c_double(float(tab["lin_dZ"])))
        return
    lib().rays_oracle_set_axisym_tables(C.byref(t))


def trace(p: RaysParams, rvec0, rindex_vec0, nthreads: int = 0) -> dict:
    rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
    rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
    nray, nv, npt = len(rvec0), p.nv, p.nstep_max + 1
    out = dict(
        ray_vec=np.zeros((nray, npt, nv)), residual=np.zeros((nray, npt)),
        npoints=np.zeros(nray, dtype=np.int32), stop_code=np.zeros(nray, dtype=np.int32),
        end_ray_vec=np.zeros((nray, nv)), end_residuals=np.zeros(nray), max_residuals=np.zeros(nray))
    nrhs = C.c_longlong(0)
    rc = lib().rays_oracle_trace(C.byref(p), nray, _dp(rvec0), _dp(rindex_vec0), _dp(out["ray_vec"]),
                                 _dp(out["residual"]), _ip(out["npoints"]), _ip(out["stop_code"]),
                                 _dp(out["end_ray_vec"]), _dp(out["end_residuals"]),
                                 _dp(out["max_residuals"]), nthreads, C.byref(nrhs))
    if rc:
        raise RuntimeError(f"rays_oracle_trace rc={rc}")
    out["nrhs"] = nrhs.value
    return out


def probe(p: RaysParams, v) -> dict:
    v = np.ascontiguousarray(v, dtype=np.float64)
    neq = 28 + 12 * (p.nspec + 1)
    eq, cold, num = np.zeros(neq), np.zeros(7), np.zeros(7)
    dvds, resid, codes = np.zeros(p.nv), np.zeros(1), np.zeros(4, dtype=np.int32)
    lib().rays_oracle_probe(C.byref(p), _dp(v), _dp(eq), _dp(cold), _dp(num), _dp(dvds), _dp(resid),
                            _ip(codes))
    return dict(eq=eq, cold=cold, num=num, dvds=dvds, resid=float(resid[0]), codes=codes)
