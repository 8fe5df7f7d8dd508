"""ctypes wrapper around tests/hip_emul/librays_emul_group.so: the lane-group SG kernel (rays_sg_group.hpp) compiled
for the host wave emulator (64 lanes per wave as fibers) -- test infrastructure."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rays_amd.params import AxisymTables, RaysParams, axisym_tables_struct

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "tests", "hip_emul")
_LIB = os.path.join(_DIR, "librays_emul_group.so")
# second build with only two register rows of phi: ordinary rays then use the workspace rows all the time
_LIB_KR2 = os.path.join(_DIR, "librays_emul_group_kr2.so")
_lib = None
_lib_kr2 = None
_variants = {}


def build(out=None, defs=()):
    global _LIB
    if out is not None:
        keep, _LIB = _LIB, out
        try:
            _build(list(defs))
        finally:
            _LIB = keep
    else:
        _build([])


def _build(defs):
    srcs = [os.path.join(_DIR, f) for f in ("emul_group.cpp", "emul_trace.cpp", "hip/hip_runtime.h", "hip/hip_wave_emul.h")]
    srcs += [os.path.join(_ROOT, "rays_amd", "csrc", f) for f in
             ("rays_libm.hpp", "rays_device.hpp", "rays_device_arith.inc", "rays_trace.hpp", "rays_sg.hpp", "rays_sg_group.hpp", "rays_dev_params.inc",
              "rays_rk4.hpp", "rays_rk4_body.inc", "rays_rk4_pass.inc")]
    if os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-extern-tls-init", "-fPIC",
                           "-shared", "-w", *defs, "-I", _DIR, srcs[0], "-o", _LIB])


def _load(path):
    l = C.CDLL(path)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    l.rays_emul_trace_group.restype = C.c_int
    l.rays_emul_trace_group.argtypes = [C.POINTER(RaysParams), C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, ip, ip, dp, dp, dp]
    l.rays_emul_trace_rk4_waves.restype = C.c_int
    l.rays_emul_trace_sg_waves.restype = C.c_int
    l.rays_emul_trace_sg_waves.argtypes = [C.POINTER(RaysParams), C.c_int, C.c_int, dp, dp, dp, dp, ip, ip, dp, dp, dp]
    l.rays_emul_trace_rk4_waves.argtypes = [C.POINTER(RaysParams), C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, ip, ip, dp, dp, dp]
    l.rays_emul_set_zfun_table.restype = C.c_int
    l.rays_emul_set_zfun_table.argtypes = [dp, C.c_int, C.c_double, C.c_double]
    from tests.emul_lib import _set_zfun
    _set_zfun(l.rays_emul_set_zfun_table)
    return l


def lib_variant(tag: str, defs):
    """The library compiled with extra -D switches (its own .so per tag)."""
    if tag not in _variants:
        path = os.path.join(_DIR, f"librays_emul_group_{tag}.so")
        build(out=path, defs=list(defs))
        _variants[tag] = _load(path)
    return _variants[tag]


def lib(small_tier=False):
    global _lib, _lib_kr2
    if small_tier:
        if _lib_kr2 is None:
            build(out=_LIB_KR2, defs=["-DRAYS_SG_GROUP_KR=2"])
            _lib_kr2 = _load(_LIB_KR2)
        return _lib_kr2
    if _lib is None:
        build()
        _lib = _load(_LIB)
    return _lib


def set_axisym_tables(tab: dict, library=None):
    t, keep = axisym_tables_struct(tab)
    fn = (library or lib()).rays_emul_set_axisym_tables
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(AxisymTables), C.c_int, C.c_double, C.c_double]
    lin = "lin_psi" in tab
    fn(C.byref(t), int(lin), float(tab["lin_dR"]) if lin else 0.0, float(tab["lin_dZ"]) if lin else 0.0)


def trace(p: RaysParams, rvec0, rindex_vec0, G: int = 8, resident_blocks: int = 2, small_tier: bool = False) -> dict:
    rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
    rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
    nray, nv, npt = len(rvec0), p.nv, p.nstep_max + 1
    out = dict(ray_vec=np.zeros((nray, npt, nv)), residual=np.zeros((nray, npt)),
               npoints=np.zeros(nray, dtype=np.int32), stop_code=np.zeros(nray, dtype=np.int32),
               end_ray_vec=np.zeros((nray, nv)), end_residuals=np.zeros(nray), max_residuals=np.zeros(nray))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = lib(small_tier).rays_emul_trace_group(C.byref(p), int(G), int(resident_blocks), nray, d(rvec0), d(rindex_vec0), d(out["ray_vec"]),
                                     d(out["residual"]), i(out["npoints"]), i(out["stop_code"]), d(out["end_ray_vec"]),
                                     d(out["end_residuals"]), d(out["max_residuals"]))
    if rc:
        raise RuntimeError(f"rays_emul_trace_group rc={rc}")
    return out


def trace_rk4_waves(p: RaysParams, rvec0, rindex_vec0, nwaves: int = 1, library=None, stride: int = 0,
                    w2_body: bool = False) -> dict:
    """The one-ray-per-lane RK4 kernel on `nwaves` whole 64-lane waves (rays beyond 64 * nwaves are pulled by lanes
    whose ray has ended; stride > 1: in the "long rays first" order of rays_trace.hpp: take_rays; w2_body: the
    body of the two-waves-per-SIMD build -- one loop, index order, residual(:) alone through the LDS window)."""
    rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
    rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
    nray, nv, npt = len(rvec0), p.nv, p.nstep_max + 1
    out = dict(ray_vec=np.zeros((nray, npt, nv)), residual=np.zeros((nray, npt)),
               npoints=np.zeros(nray, dtype=np.int32), stop_code=np.zeros(nray, dtype=np.int32),
               end_ray_vec=np.zeros((nray, nv)), end_residuals=np.zeros(nray), max_residuals=np.zeros(nray))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    (library or lib()).rays_emul_rk4_waves_use_w2_body(int(w2_body))
    rc = (library or lib()).rays_emul_trace_rk4_waves(C.byref(p), int(nwaves), int(stride), nray, d(rvec0), d(rindex_vec0),
                                                      d(out["ray_vec"]), d(out["residual"]), i(out["npoints"]),
                                                      i(out["stop_code"]), d(out["end_ray_vec"]), d(out["end_residuals"]),
                                                      d(out["max_residuals"]))
    if rc:
        raise RuntimeError(f"rays_emul_trace_rk4_waves rc={rc}")
    return out


def trace_sg_waves(p: RaysParams, rvec0, rindex_vec0, nwaves: int = 1, library=None) -> dict:
    """The one-ray-per-lane Shampine-Gordon kernel on `nwaves` whole 64-lane waves (rays beyond 64 * nwaves are pulled
    by lanes whose ray has ended)."""
    rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
    rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
    nray, nv, npt = len(rvec0), p.nv, p.nstep_max + 1
    out = dict(ray_vec=np.zeros((nray, npt, nv)), residual=np.zeros((nray, npt)),
               npoints=np.zeros(nray, dtype=np.int32), stop_code=np.zeros(nray, dtype=np.int32),
               end_ray_vec=np.zeros((nray, nv)), end_residuals=np.zeros(nray), max_residuals=np.zeros(nray))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = (library or lib()).rays_emul_trace_sg_waves(C.byref(p), int(nwaves), nray, d(rvec0), d(rindex_vec0),
                                                     d(out["ray_vec"]), d(out["residual"]), i(out["npoints"]),
                                                     i(out["stop_code"]), d(out["end_ray_vec"]), d(out["end_residuals"]),
                                                     d(out["max_residuals"]))
    if rc:
        raise RuntimeError(f"rays_emul_trace_sg_waves rc={rc}")
    return out
