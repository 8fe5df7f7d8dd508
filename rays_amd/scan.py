"""Fused `ray_scan`: the reference's scan driver re-initialises and calls
`trace_rays` once per scan value, serially (RAYS_project/ray_scan/ray_scan.f90:33-49, scanner_m.f90:
100-205; the scanned parameter is the step size `ds`).  A scan is just more independent rays: here
every scan value gets its own parameter block and output arrays and all runs are launched
back-to-back on separate HIP streams, so small fans (a 1024-ray fan occupies 16 of the GPU's 1024
SIMDs) fill the machine together.  Results per run are exactly those of a stand-alone trace.

How many runs really overlap is set by the HIP runtime's hardware queues: 4 by default.  With
GPU_MAX_HW_QUEUES=16 in the environment BEFORE the process first touches the GPU, 64 runs of the
1024-ray fan take 16.3 ms instead of 44.9 ms (162 ms one after another; 8 queues: 25.1 ms, 32: slower).
`HW_QUEUES_ENV` below is that setting; tools/scan_speed.py applies it.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .params import ConfigError, RaysParams, copy_params

HW_QUEUES_ENV = ("GPU_MAX_HW_QUEUES", "16")
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
    """All runs of a `ds` scan in flight at once."""

    def __init__(self, params: RaysParams, rvec0, rindex_vec0, ds_values: Sequence[float],
                 scan_parameter: str = "ds"):
        import torch

        if scan_parameter.strip() != "ds":  # scanner_m.f90:185-201 ('*_num_threads' has no meaning here)
            raise ConfigError(f"initialize_scanner_m: unknown scan parameter = {scan_parameter!r}")
        self.torch = torch
        self.runs: List[DeviceTrace] = []
        for v in ds_values:
            q = copy_params(params)
            q.ds = float(v)
            self.runs.append(DeviceTrace(q, rvec0, rindex_vec0))
        self.streams = [torch.cuda.Stream() for _ in self.runs]

    def launch(self, zero_fill: bool = True):
        t = self.torch
        ready = t.cuda.Event()
        ready.record()
        for s, r in zip(self.streams, self.runs):
            s.wait_event(ready)
            with t.cuda.stream(s):
                r.launch(zero_fill=zero_fill)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def results(self) -> List[RayResults]:
        self.synchronize()
        return [r.results() for r in self.runs]
