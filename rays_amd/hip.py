"""ctypes binding of the C ABI in include/rays_hip.h (librays_hip.so).

This is the only compute path of the package: if the HIP library is missing or no GPU is
visible the calls raise -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .params import AxisymTables, RaysFan, RaysParams, axisym_tables_struct


class DeviceResult(C.Structure):
    """rays_device_result_t (include/rays_hip.h)"""
    _fields_ = [("device", C.c_int32), ("nray", C.c_int32), ("ray_vec", C.c_void_p), ("residual", C.c_void_p),
                ("npoints", C.c_void_p), ("stop_code", C.c_void_p), ("end_ray_vec", C.c_void_p),
                ("end_residuals", C.c_void_p), ("max_residuals", C.c_void_p)]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAYS_HIP_LIB") or os.path.join(_HERE, "lib", "librays_hip.so")

# every symbol include/rays_hip.h declares
EXPORTED_SYMBOLS = (
    "rays_hip_init", "rays_hip_init_devices", "rays_hip_finalize", "rays_hip_device_count", "rays_hip_sizeof_params",
    "rays_hip_last_error", "rays_hip_set_zfun_table", "rays_hip_set_axisym_tables", "rays_hip_set_eqdsk_lin_tables",
    "rays_hip_stop_flag_text", "rays_hip_check_params", "rays_hip_trace", "rays_hip_trace_gather", "rays_hip_result_to_host", "rays_hip_trace_device", "rays_hip_scan_device", "rays_hip_ode_step_device",
    "rays_hip_kernel_name", "rays_hip_kernel_name_for", "rays_hip_probe", "rays_hip_pack_device", "rays_hip_unpack_device",
    "rays_hip_sizeof_fan", "rays_hip_ray_init", "rays_hip_ray_init_device",
    "rays_hip_set_rho_table", "rays_hip_deposition_device", "rays_hip_deposition",
    "rays_hip_keep_last_result", "rays_hip_deposition_last",
    "rays_hip_set_numerics", "rays_hip_get_numerics",
)

_lib = None


class RaysHipError(RuntimeError):
    pass


def load():
    """Load librays_hip.so (built by `make -C rays_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime (libamdhip64.so.7, the SONAME of the system's too).  Whichever is loaded
    # first serves the whole process; loaded in the other order -- this library (and with it /opt/rocm's runtime)
    # first, torch afterwards -- the second runtime to initialise finds "no ROCm-capable device".  The Python host
    # uses torch for device memory and streams (DeviceTrace, ode_step, bench.py), so torch goes first, always.
    # A host that needs the C ABI alone (no DeviceTrace / ode_step / bench) sets RAYS_AMD_NO_TORCH=1 and saves the import.
    if "torch" in __import__("sys").modules or not os.environ.get("RAYS_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise RaysHipError(
            f"{LIB_PATH} not found: build it with `make -C rays_amd/csrc` (needs hipcc, gfx950). "
            "rays_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    pp = C.POINTER(RaysParams)
    lib.rays_hip_init.restype = C.c_int
    lib.rays_hip_init.argtypes = [C.c_int]
    lib.rays_hip_init_devices.restype = C.c_int
    lib.rays_hip_init_devices.argtypes = [C.c_int, C.POINTER(C.c_int)]
    lib.rays_hip_finalize.restype = C.c_int
    lib.rays_hip_device_count.restype = C.c_int
    lib.rays_hip_sizeof_params.restype = C.c_int
    if lib.rays_hip_sizeof_params() != C.sizeof(RaysParams):
        raise RaysHipError("rays_params_t layout mismatch between librays_hip.so and rays_amd.params")
    lib.rays_hip_last_error.restype = C.c_int
    lib.rays_hip_last_error.argtypes = [C.c_char_p, C.c_int]
    lib.rays_hip_stop_flag_text.restype = C.c_char_p
    lib.rays_hip_stop_flag_text.argtypes = [C.c_int]
    lib.rays_hip_set_zfun_table.restype = C.c_int
    lib.rays_hip_set_zfun_table.argtypes = [dp, C.c_int, C.c_double, C.c_double]
    lib.rays_hip_set_axisym_tables.restype = C.c_int
    lib.rays_hip_set_axisym_tables.argtypes = [C.POINTER(AxisymTables)]
    lib.rays_hip_set_eqdsk_lin_tables.restype = C.c_int
    lib.rays_hip_set_eqdsk_lin_tables.argtypes = [C.POINTER(AxisymTables), C.c_double, C.c_double]
    lib.rays_hip_check_params.restype = C.c_int
    lib.rays_hip_check_params.argtypes = [pp]
    lib.rays_hip_kernel_name.restype = C.c_char_p
    lib.rays_hip_kernel_name.argtypes = [pp]
    lib.rays_hip_kernel_name_for.restype = C.c_char_p
    lib.rays_hip_kernel_name_for.argtypes = [pp, C.c_int]
    lib.rays_hip_trace.restype = C.c_int
    lib.rays_hip_trace.argtypes = [pp, C.c_int, dp, dp, dp, dp, ip, ip, dp, dp, dp, dp]
    lib.rays_hip_trace_gather.restype = C.c_int
    lib.rays_hip_trace_gather.argtypes = [pp, C.c_int, dp, dp, C.POINTER(DeviceResult)]
    lib.rays_hip_result_to_host.restype = C.c_int
    lib.rays_hip_result_to_host.argtypes = [pp, C.POINTER(DeviceResult), dp, dp, ip, ip, dp, dp, dp]
    lib.rays_hip_trace_device.restype = C.c_int
    lib.rays_hip_trace_device.argtypes = [pp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.rays_hip_scan_device.restype = C.c_int
    lib.rays_hip_scan_device.argtypes = [pp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.rays_hip_ode_step_device.restype = C.c_int
    lib.rays_hip_ode_step_device.argtypes = [pp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.rays_hip_pack_device.restype = C.c_int
    lib.rays_hip_pack_device.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.rays_hip_unpack_device.restype = C.c_int
    lib.rays_hip_unpack_device.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.rays_hip_probe.restype = C.c_int
    lib.rays_hip_probe.argtypes = [pp, C.c_int, dp, dp, dp, dp, dp, ip]
    lib.rays_hip_sizeof_fan.restype = C.c_int
    if lib.rays_hip_sizeof_fan() != C.sizeof(RaysFan):
        raise RaysHipError("rays_fan_t layout mismatch between librays_hip.so and rays_amd.params")
    fp = C.POINTER(RaysFan)
    lib.rays_hip_ray_init.restype = C.c_int
    lib.rays_hip_ray_init.argtypes = [pp, fp, C.c_int, dp, dp, dp, ip]
    lib.rays_hip_ray_init_device.restype = C.c_int
    lib.rays_hip_ray_init_device.argtypes = [pp, fp, C.c_int, vp, vp, ip, vp]
    lib.rays_hip_set_rho_table.restype = C.c_int
    lib.rays_hip_set_rho_table.argtypes = [dp, dp, C.c_int]
    lib.rays_hip_deposition.restype = C.c_int
    lib.rays_hip_deposition.argtypes = [pp, C.c_int, C.c_int, C.c_int, dp, ip, dp, dp, dp]
    lib.rays_hip_deposition_device.restype = C.c_int
    lib.rays_hip_deposition_device.argtypes = [pp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.rays_hip_keep_last_result.restype = C.c_int
    lib.rays_hip_keep_last_result.argtypes = [C.c_int]
    lib.rays_hip_deposition_last.restype = C.c_int
    lib.rays_hip_deposition_last.argtypes = [pp, C.c_int, C.c_int, C.c_int, dp, dp, dp]
    lib.rays_hip_set_numerics.restype = C.c_int
    lib.rays_hip_set_numerics.argtypes = [C.c_int]
    lib.rays_hip_get_numerics.restype = C.c_int
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    load().rays_hip_last_error(buf, 1024)
    return buf.value.decode(errors="replace")


def _check(rc: int, what: str):
    if rc != 0:
        raise RaysHipError(f"{what} failed (rc={rc}): {last_error()}")


NUMERICS = {"exact": 0, "tolerance": 1}


def set_numerics(mode: str) -> str:
    """rays_hip_set_numerics: "exact" (bit-identical to the reference CPU path; default) or "tolerance" (not
    bit-identical: every step within 1e-10 relative of the reference's -- 4e-15 measured on the headline fan --, exact
    ray counts / step indices / stop flags; cold RK4 kernels; include/rays_hip.h).  Returns the previous setting."""
    prev = load().rays_hip_set_numerics(NUMERICS[mode])
    if prev < 0:
        raise RaysHipError("rays_hip_set_numerics: " + last_error())
    return "tolerance" if prev == 1 else "exact"


def get_numerics() -> str:
    return "tolerance" if load().rays_hip_get_numerics() == 1 else "exact"


def stop_flag_text(code: int) -> str:
    return load().rays_hip_stop_flag_text(int(code)).decode()


_ZFUN_PATH = os.path.join(_HERE, "data", "zfun_spline_re.npz")
_zfun_set = False


def set_zfun_table(fspl_re=None, x_min=None, x_max=None):
    """Hand the Z-function spline table (fsplRe[nx][4]) to the library.  Default: the table shipped
    in rays_amd/data (cut from the reference's initialize_spline_coeffs, zfunctions_m.f90:436-466)."""
    global _zfun_set
    if fspl_re is None:
        z = np.load(_ZFUN_PATH)
        fspl_re, x_min, x_max = z["fspl_re"], float(z["x_min"]), float(z["x_max"])
    fspl_re = np.ascontiguousarray(fspl_re, dtype=np.float64)
    _check(load().rays_hip_set_zfun_table(_dp(fspl_re), len(fspl_re), float(x_min), float(x_max)),
           "rays_hip_set_zfun_table")
    _zfun_set = True


def set_axisym_tables(tab: dict):
    """Hand the host-built spline tables of an eqdsk equilibrium to the library (copied)."""
    t, keep = axisym_tables_struct(tab)
    if "lin_psi" in tab:   # magnetics_model = 'eqdsk_magnetics_lin_interp'
        _check(load().rays_hip_set_eqdsk_lin_tables(C.byref(t), float(tab["lin_dR"]), float(tab["lin_dZ"])),
               "rays_hip_set_eqdsk_lin_tables")
        return
    _check(load().rays_hip_set_axisym_tables(C.byref(t)), "rays_hip_set_axisym_tables")


def ensure_tables(p: RaysParams):
    if p.damping_model and not _zfun_set:
        set_zfun_table()


def check_params(p: RaysParams):
    _check(load().rays_hip_check_params(C.byref(p)), "rays_hip_check_params")


def kernel_name(p: RaysParams, nray: int = 0) -> str:
    """Kernel specialisation a trace of `nray` rays would launch (0: the default build)."""
    lib = load()
    name = (lib.rays_hip_kernel_name_for(C.byref(p), int(nray)) if nray else lib.rays_hip_kernel_name(C.byref(p))).decode()
    if not name:   # refused parameters or a shape the library was not built with (make FULL=1): never a silent ""
        check_params(p)
        raise RaysHipError("rays_hip_kernel_name: no kernel for this configuration")
    return name


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def ray_init_host(p: RaysParams, fan: RaysFan, nray_max: int):
    """rays_hip_ray_init: the launch fan built on the GPU, returned as host arrays
    (rvec0[nray][3], rindex_vec0[nray][3], ray_pwr_wt[nray]) -- the ray_init_m contract."""
    n = int(nray_max)
    rvec0, rindex_vec0, w = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
    nray = C.c_int32(0)
    _check(load().rays_hip_ray_init(C.byref(p), C.byref(fan), n, _dp(rvec0), _dp(rindex_vec0), _dp(w),
                                    C.byref(nray)), "rays_hip_ray_init")
    k = nray.value
    return rvec0[:k].copy(), rindex_vec0[:k].copy(), w[:k].copy()


def ray_init_device(p: RaysParams, fan: RaysFan, nray_max: int, d_rvec0: int, d_rindex_vec0: int,
                    stream: int = 0) -> int:
    """rays_hip_ray_init_device: fills the device arrays (nray_max x 3 doubles each), returns nray."""
    nray = C.c_int32(0)
    _check(load().rays_hip_ray_init_device(C.byref(p), C.byref(fan), int(nray_max), d_rvec0, d_rindex_vec0,
                                           C.byref(nray), stream), "rays_hip_ray_init_device")
    return nray.value


DEP_PROFILES = {"Ptotal_psi": 0, "Ptotal_rho": 1, "Ptotal_x": 2}


def set_rho_table(grid, fspl):
    """rho(psiN) spline of the eqdsk equilibrium (needed for the Ptotal_rho deposition profile)."""
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    fspl = np.ascontiguousarray(fspl, dtype=np.float64)
    _check(load().rays_hip_set_rho_table(_dp(grid), _dp(fspl), len(grid)), "rays_hip_set_rho_table")


def deposition_device(p: RaysParams, which: str, n_bins: int, nray: int, d_ray_vec: int, d_npoints: int,
                      d_power: int, d_work: int, d_profile_in, d_profile_out: int, stream: int = 0):
    """rays_hip_deposition_device on device pointers (torch `.data_ptr()`s); d_profile_in may be None."""
    _check(load().rays_hip_deposition_device(C.byref(p), DEP_PROFILES[which], int(n_bins), int(nray), d_ray_vec,
                                             d_npoints, d_power, d_work, d_profile_in, d_profile_out, stream),
           "rays_hip_deposition_device")


def init_devices(device_ids):
    """rays_hip_init_devices: explicit slot -> device list for trace_host (repeats allowed)."""
    ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
    if load().rays_hip_init_devices(len(device_ids), ids) < 0:
        raise RaysHipError("rays_hip_init_devices: " + last_error())


def deposition_host(p: RaysParams, which: str, n_bins: int, ray_vec, npoints, initial_ray_power):
    """rays_hip_deposition: host arrays in (the ray_results_m image), (work[nray][n_bins], profile[n_bins]) out."""
    ray_vec = np.ascontiguousarray(ray_vec, dtype=np.float64)
    npoints = np.ascontiguousarray(npoints, dtype=np.int32)
    power = np.ascontiguousarray(initial_ray_power, dtype=np.float64)
    nray = len(npoints)
    work, prof = np.zeros((nray, n_bins)), np.zeros(n_bins)
    _check(load().rays_hip_deposition(C.byref(p), DEP_PROFILES[which], int(n_bins), nray, _dp(ray_vec), _ip(npoints),
                                      _dp(power), _dp(work), _dp(prof)), "rays_hip_deposition")
    return work, prof


NO_KEPT_RESULT = 5   # RAYS_HIP_NO_KEPT_RESULT


def keep_last_result(on: bool) -> bool:
    """rays_hip_keep_last_result: later trace_host calls leave their result's device image in place for
    deposition_last (until the next trace_host / keep_last_result(False)).  Returns the previous setting."""
    return bool(load().rays_hip_keep_last_result(1 if on else 0))


def deposition_last(p: RaysParams, which: str, n_bins: int, initial_ray_power, want_work: bool = True):
    """rays_hip_deposition_last: the profiles of the rays the last trace_host call traced, binned on the device(s) that
    hold them (no trajectory upload).  Returns (work[nray][n_bins] or None, profile[n_bins]), or None when no such
    image is held."""
    power = np.ascontiguousarray(initial_ray_power, dtype=np.float64)
    nray = len(power)
    work = np.zeros((nray, n_bins)) if want_work else None
    prof = np.zeros(n_bins)
    rc = load().rays_hip_deposition_last(C.byref(p), DEP_PROFILES[which], int(n_bins), nray, _dp(power),
                                         _dp(work) if want_work else None, _dp(prof))
    if rc == NO_KEPT_RESULT:
        return None
    _check(rc, "rays_hip_deposition_last")
    return work, prof


def trace_host(p: RaysParams, rvec0, rindex_vec0, ngpu: int = 0, out: dict = None) -> dict:
    """rays_hip_trace: host numpy arrays in / out (the Fortran drop-in entry).  `out`: the result
    arrays of an earlier call over the same fan to write into (the library overwrites points
    1..npoints only, so they must hold zeros or an earlier result of the same fan -- the Fortran host's
    situation, whose arrays are allocated and zero-filled once by initialize_ray_results_m)."""
    lib = load()
    rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
    rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
    nray, nv, npt = len(rvec0), p.nv, p.nstep_max + 1
    if ngpu is not None and lib.rays_hip_init(int(ngpu)) < 0:   # None: keep the init_devices() selection
        raise RaysHipError("rays_hip_init: " + last_error())
    ensure_tables(p)
    if out is None:
        out = dict(
            ray_vec=np.zeros((nray, npt, nv)), residual=np.zeros((nray, npt)),
            npoints=np.zeros(nray, dtype=np.int32), stop_code=np.zeros(nray, dtype=np.int32),
            end_ray_vec=np.zeros((nray, nv)), end_residuals=np.zeros(nray), max_residuals=np.zeros(nray))
    elif out["ray_vec"].shape != (nray, npt, nv) or not out["ray_vec"].flags.c_contiguous:
        raise ValueError("trace_host: `out` does not match this fan")
    el = C.c_double(0.0)
    rc = lib.rays_hip_trace(C.byref(p), nray, _dp(rvec0), _dp(rindex_vec0), _dp(out["ray_vec"]),
                            _dp(out["residual"]), _ip(out["npoints"]), _ip(out["stop_code"]),
                            _dp(out["end_ray_vec"]), _dp(out["end_residuals"]),
                            _dp(out["max_residuals"]), C.byref(el))
    _check(rc, "rays_hip_trace")
    out["elapsed_s"] = el.value
    return out


def trace_gather(p: RaysParams, rvec0, rindex_vec0, to_host: bool = True):
    """rays_hip_trace_gather: rays sharded over the selected devices, trajectories gathered on the root device
    with RCCL.  Returns the device-resident result block and, with to_host, the arrays copied out."""
    lib = load()
    ensure_tables(p)
    rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
    rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
    res = DeviceResult()
    _check(lib.rays_hip_trace_gather(C.byref(p), len(rvec0), _dp(rvec0), _dp(rindex_vec0), C.byref(res)),
           "rays_hip_trace_gather")
    if not to_host:
        return res, None
    nray, nv, npt = len(rvec0), p.nv, p.nstep_max + 1
    out = dict(ray_vec=np.zeros((nray, npt, nv)), residual=np.zeros((nray, npt)),
               npoints=np.zeros(nray, dtype=np.int32), stop_code=np.zeros(nray, dtype=np.int32),
               end_ray_vec=np.zeros((nray, nv)), end_residuals=np.zeros(nray), max_residuals=np.zeros(nray))
    _check(lib.rays_hip_result_to_host(C.byref(p), C.byref(res), _dp(out["ray_vec"]), _dp(out["residual"]),
                                       _ip(out["npoints"]), _ip(out["stop_code"]), _dp(out["end_ray_vec"]),
                                       _dp(out["end_residuals"]), _dp(out["max_residuals"])),
           "rays_hip_result_to_host")
    return res, out


def trace_device(p: RaysParams, nray: int, d_rvec0: int, d_rindex_vec0: int, d_ray_vec: int,
                 d_residual: int, d_npoints: int, d_stop_code: int, d_end_ray_vec: int = 0,
                 d_end_residuals: int = 0, d_max_residuals: int = 0, stream: int = 0,
                 zero_fill: bool = True):
    """rays_hip_trace_device: raw device pointers (ints), asynchronous on `stream`."""
    ensure_tables(p)
    rc = load().rays_hip_trace_device(
        C.byref(p), int(nray), d_rvec0, d_rindex_vec0, d_ray_vec, d_residual, d_npoints, d_stop_code,
        d_end_ray_vec or None, d_end_residuals or None, d_max_residuals or None, stream or None,
        0 if zero_fill else 1)
    _check(rc, "rays_hip_trace_device")


def scan_device(p: RaysParams, n_runs: int, d_ds_values: int, nray: int, d_rvec0: int, d_rindex_vec0: int,
                d_ray_vec: int, d_residual: int, d_npoints: int, d_stop_code: int, d_end_ray_vec: int = 0,
                d_end_residuals: int = 0, d_max_residuals: int = 0, stream: int = 0, zero_fill: bool = True):
    """rays_hip_scan_device: all runs of a `ds` scan in ONE launch (outputs carry a leading run dimension)."""
    ensure_tables(p)
    rc = load().rays_hip_scan_device(
        C.byref(p), int(n_runs), d_ds_values, int(nray), d_rvec0, d_rindex_vec0, d_ray_vec, d_residual,
        d_npoints, d_stop_code, d_end_ray_vec or None, d_end_residuals or None, d_max_residuals or None,
        stream or None, 0 if zero_fill else 1)
    _check(rc, "rays_hip_scan_device")


def ode_step(p: RaysParams, v0, s0=None):
    """rays_hip_ode_step_device on host arrays (through torch device buffers): one output step of the
    configured ODE solver + check_save from each state v0[n][nv].  Returns (v1, resid, stop_code)."""
    import torch

    ensure_tables(p)
    v0 = np.ascontiguousarray(v0, dtype=np.float64).reshape(-1, p.nv)
    n = len(v0)
    d_v0 = torch.as_tensor(v0).cuda()
    d_s0 = None if s0 is None else torch.as_tensor(np.ascontiguousarray(s0, dtype=np.float64)).cuda()
    d_v1 = torch.zeros((n, p.nv), dtype=torch.float64, device="cuda")
    d_res = torch.zeros(n, dtype=torch.float64, device="cuda")
    d_sc = torch.zeros(n, dtype=torch.int32, device="cuda")
    rc = load().rays_hip_ode_step_device(C.byref(p), n, d_v0.data_ptr(), None if d_s0 is None else d_s0.data_ptr(),
                                         d_v1.data_ptr(), d_res.data_ptr(), d_sc.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream or None)
    _check(rc, "rays_hip_ode_step_device")
    return d_v1.cpu().numpy(), d_res.cpu().numpy(), d_sc.cpu().numpy()


def pack_device(nray, nv, nstep_max, d_npoints, d_offsets, d_ray_vec, d_residual, d_packed_vec,
                d_packed_res, stream=0):
    _check(load().rays_hip_pack_device(int(nray), int(nv), int(nstep_max), d_npoints, d_offsets,
                                       d_ray_vec, d_residual, d_packed_vec, d_packed_res,
                                       stream or None), "rays_hip_pack_device")


def unpack_device(nray, nv, nstep_max, d_npoints, d_offsets, d_packed_vec, d_packed_res, d_ray_vec,
                  d_residual, stream=0):
    _check(load().rays_hip_unpack_device(int(nray), int(nv), int(nstep_max), d_npoints, d_offsets,
                                         d_packed_vec, d_packed_res, d_ray_vec, d_residual,
                                         stream or None), "rays_hip_unpack_device")


def probe(p: RaysParams, v) -> dict:
    v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1, p.nv)
    n = len(v)
    cold, num = np.zeros((n, 7)), np.zeros((n, 7))
    dvds, resid, codes = np.zeros((n, p.nv)), np.zeros(n), np.zeros((n, 4), dtype=np.int32)
    rc = load().rays_hip_probe(C.byref(p), n, _dp(v), _dp(cold), _dp(num), _dp(dvds), _dp(resid),
                               _ip(codes))
    _check(rc, "rays_hip_probe")
    return dict(cold=cold, num=num, dvds=dvds, resid=resid, codes=codes)
