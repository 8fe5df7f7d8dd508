"""Fused `ray_scan`: the reference's scan driver re-initialises and calls
`trace_rays` once per scan value, serially (RAYS_project/ray_scan/ray_scan.f90:33-49, scanner_m.f90:
100-205; the scanned parameter is the step size `ds`).  A scan is just more independent rays: here
ALL runs go out as ONE launch of the trace kernel over n_runs x nray rays (rays_hip_scan_device): ray
i belongs to run i // nray, which sets its step `ds`; the persistent waves and their refill counter
treat the lot as one fan, so small fans (a 1024-ray fan occupies 16 of the GPU's 1024 SIMDs) fill the
machine together and 64 runs of a 1024-ray fan cost about one 64k-ray pass.  Results per run are
exactly those of a stand-alone trace.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .params import ConfigError, RaysParams, copy_params

from .trace import DeviceTrace, RayResults


def scan_values(scan_algorithm: str, n_runs: int, p_start: float = 0.0, p_incr: float = 0.0, p_max: float = 0.0,
                max_divide: int = 0, S_max: float = 0.0, n_max: int = 0, k_factor: int = 0,
                delta: float = float(np.float32(1.0e-14))) -> np.ndarray:  # scanner_m.f90:39 (single literal)
    """p_values(1:n_runs) of scanner_m.f90:115-160 (the algorithms that scan a real parameter)."""
    i = np.arange(1, n_runs + 1, dtype=np.float64)
    a = scan_algorithm.strip()
    if a == "fixed_increment":
        return p_start + (i - 1.0) * p_incr
    if a == "pwr_of_2":
        return p_max / np.float32(2.0) ** (n_runs - i) - delta
    if a == "integer_divide":
        return p_max / (max_divide - i + 1.0)
    if a == "algorithm_1":
        return S_max / (n_max + k_factor * (n_runs - i)) - delta
    raise ConfigError(f"initialize_scanner_m: unknown scan algorithm = {scan_algorithm!r}")


class RayScan:
    """All runs of a `ds` scan in one launch; outputs carry a leading run dimension."""

    def __init__(self, params: RaysParams, rvec0, rindex_vec0, ds_values: Sequence[float],
                 scan_parameter: str = "ds", device=None):
        import torch

        from . import hip

        if scan_parameter.strip() != "ds":  # scanner_m.f90:185-201 ('*_num_threads' has no meaning here)
            raise ConfigError(f"initialize_scanner_m: unknown scan parameter = {scan_parameter!r}")
        self.torch, self.hip = torch, hip
        self.params = params
        hip.check_params(params)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.ds_values = np.ascontiguousarray(ds_values, dtype=np.float64)
        self.n_runs, self.nray = len(self.ds_values), len(rvec0)
        nv, npt = params.nv, params.nstep_max + 1
        f64, i32 = torch.float64, torch.int32
        R, N = self.n_runs, self.nray
        with torch.cuda.device(self.device):
            self.ds = torch.as_tensor(self.ds_values).to(self.device)
            self.rvec0 = torch.as_tensor(np.ascontiguousarray(rvec0), dtype=f64).to(self.device)
            self.rindex_vec0 = torch.as_tensor(np.ascontiguousarray(rindex_vec0), dtype=f64).to(self.device)
            self.ray_vec = torch.zeros((R, N, npt, nv), dtype=f64, device=self.device)
            self.residual = torch.zeros((R, N, npt), dtype=f64, device=self.device)
            self.npoints = torch.zeros((R, N), dtype=i32, device=self.device)
            self.stop_code = torch.zeros((R, N), dtype=i32, device=self.device)
            self.end_ray_vec = torch.zeros((R, N, nv), dtype=f64, device=self.device)
            self.end_residuals = torch.zeros((R, N), dtype=f64, device=self.device)
            self.max_residuals = torch.zeros((R, N), dtype=f64, device=self.device)

    def launch(self, zero_fill: bool = True):
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        self.hip.scan_device(self.params, self.n_runs, self.ds.data_ptr(), self.nray, self.rvec0.data_ptr(),
                             self.rindex_vec0.data_ptr(), self.ray_vec.data_ptr(), self.residual.data_ptr(),
                             self.npoints.data_ptr(), self.stop_code.data_ptr(), self.end_ray_vec.data_ptr(),
                             self.end_residuals.data_ptr(), self.max_residuals.data_ptr(), stream=stream,
                             zero_fill=zero_fill)

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def results(self) -> List[RayResults]:
        self.synchronize()
        c = lambda x: x.cpu().numpy()
        arrays = [c(a) for a in (self.ray_vec, self.residual, self.npoints, self.stop_code, self.end_ray_vec,
                                 self.end_residuals, self.max_residuals)]
        return [RayResults(*(a[r] for a in arrays)) for r in range(self.n_runs)]
