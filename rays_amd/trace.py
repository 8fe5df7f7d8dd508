"""Host-side mirror of the reference's run contract for the hot path:

    call initialize(read_input)   ->  RaysRun.from_namelist(path)      (module state)
    call trace_rays               ->  run.trace_rays()                 (the HIP path)
    ray_results_m arrays          ->  RayResults

(RAYS_project/RAYS_code/RAYS.f90:10-16, RAYS_lib/ray_results_m.f90:44-58.)  PyTorch is used only
as plumbing for device memory / streams / torch.distributed; the computation is
librays_hip.so through the C ABI.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, Optional

import numpy as np

from . import hip
from .namelist import read_namelist
from .params import RaysParams, params_from_namelist
from .ray_init import initialize_ray_init


@dataclasses.dataclass
class RayResults:
    """Image of the ray_results_m module arrays (ray_results_m.f90:44-58), C order."""

    ray_vec: np.ndarray          # [nray][nstep_max+1][nv]
    residual: np.ndarray         # [nray][nstep_max+1]
    npoints: np.ndarray          # [nray]
    stop_code: np.ndarray        # [nray]   integer image of ray_stop_flag
    end_ray_vec: np.ndarray      # [nray][nv]
    end_residuals: np.ndarray    # [nray]
    max_residuals: np.ndarray    # [nray]
    elapsed_s: float = 0.0

    @property
    def ray_stop_flag(self):
        return [hip.stop_flag_text(int(c)) for c in self.stop_code]

    @property
    def start_ray_vec(self):      # ray_tracing.f90:259
        return self.ray_vec[:, 0, :]

    @property
    def end_ray_parameter(self):  # ray_tracing.f90:257
        return self.end_ray_vec[:, 6]

    @property
    def total_steps(self) -> int:
        return int(np.maximum(self.npoints.astype(np.int64) - 1, 0).sum())


def load_axisym_tables(namelist_path: str, nml: Dict[str, Dict[str, Any]]) -> Optional[Dict[str, Any]]:
    """Host-built spline tables of an eqdsk equilibrium: `<eqdsk_file_name>.tables.npz` next to the
    namelist (None for the analytic equilibria)."""
    import os

    if str(nml.get("equilibrium_list", {}).get("equilib_model", "")).strip() != "axisym_toroid":
        return None
    if str(nml.get("axisym_toroid_eq_list", {}).get("magnetics_model", "")).strip() == "solovev_magnetics":
        return None   # analytic magnetics: nothing to load (spline PROFILE models would still need their tables)
    here = os.path.dirname(os.path.abspath(namelist_path))
    if str(nml.get("axisym_toroid_eq_list", {}).get("magnetics_model", "")).strip() == "eqdsk_magnetics_lin_interp":
        # bilinear eqdsk model: no spline fit, the host mirror reads the g-eqdsk itself (rays_amd/eqdsk.py); splined
        # PROFILES still come from the RAYS host's tables file when there is one
        from .eqdsk import eqdsk_lin_tables
        eq = str(nml.get("eqdsk_magnetics_lin_interp_list", {}).get("eqdsk_file_name", "")).strip()
        tab = eqdsk_lin_tables(os.path.join(here, eq))
        f = os.path.join(here, eq + ".tables.npz")
        if os.path.exists(f):
            z = np.load(f)
            tab.update({k: z[k] for k in z.files if k[:3] in ("ne_", "te_", "ti_")})
        return tab
    eq = str(nml.get("eqdsk_magnetics_spline_interp_list", {}).get("eqdsk_file_name", "")).strip()
    f = os.path.join(here, eq + ".tables.npz")
    if not os.path.exists(f):
        raise FileNotFoundError(f"{f}: spline tables of the eqdsk equilibrium (built by the RAYS host)")
    z = np.load(f)
    return {k: (float(z[k]) if z[k].ndim == 0 else z[k]) for k in z.files}


class RaysRun:
    """Module state after `initialize`: parameters + launched fan."""

    def __init__(self, params: RaysParams, rvec0, rindex_vec0, ray_pwr_wt=None,
                 namelist: Optional[Dict[str, Dict[str, Any]]] = None):
        self.params = params
        self.rvec0 = np.ascontiguousarray(rvec0, dtype=np.float64)
        self.rindex_vec0 = np.ascontiguousarray(rindex_vec0, dtype=np.float64)
        self.ray_pwr_wt = ray_pwr_wt
        self.namelist = namelist
        hip.check_params(params)

    @classmethod
    def from_namelist(cls, path: str, axisym_tables: Optional[Dict[str, Any]] = None) -> "RaysRun":
        """`initialize(read_input=.true.)`.  For equilib_model = 'axisym_toroid' the host-built spline
        tables are taken from `axisym_tables` or from `<eqdsk_file_name>.tables.npz` next to the
        namelist (written by a RAYS host, see tests/golden/make_golden.py)."""
        nml = read_namelist(path)
        tab = axisym_tables if axisym_tables is not None else load_axisym_tables(path, nml)
        p = params_from_namelist(nml, tab)
        if tab is not None:
            hip.set_axisym_tables(tab)
            if "rho_grid" in tab:
                hip.set_rho_table(tab["rho_grid"], tab["rho_fspl"])
        import os
        r0, n0, w = initialize_ray_init(p, nml, tab, base_dir=os.path.dirname(os.path.abspath(path)))
        return cls(p, r0, n0, w, nml)

    @property
    def nray(self) -> int:
        return len(self.rvec0)

    def trace_rays(self, ngpu: int = 1) -> RayResults:
        out = hip.trace_host(self.params, self.rvec0, self.rindex_vec0, ngpu=ngpu)
        return RayResults(**out)

    def finalize_run(self, res: RayResults, directory: str = ".", list_directed: bool = True, netcdf: bool = True):
        """finalize_run.f90:20-28: write run_results.<run_label> (list-directed) and/or
        run_results.<run_label>.nc from the results of this run (rays_amd/results.py)."""
        import os

        from . import results as rf

        label = str((self.namelist or {}).get("diagnostics_list", {}).get("run_label", "")).strip()
        rr = rf.RunResults(res, self.ray_pwr_wt, label)
        base = os.path.join(directory, "run_results." + label)
        if list_directed:
            rf.write_results_LD(base, rr)
        if netcdf:
            rf.write_results_NC(base + ".nc", rr)
        return rr


class DeviceTrace:
    """Device-resident trace: inputs/outputs are torch CUDA tensors, launches are asynchronous on
    the current torch stream (used by bench.py and by multi-GPU runs)."""

    def __init__(self, params: RaysParams, rvec0, rindex_vec0, device=None):
        import torch

        self.torch = torch
        self.params = params
        hip.check_params(params)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.nray = len(rvec0)
        nv, npt = params.nv, params.nstep_max + 1
        f64, i32 = torch.float64, torch.int32
        with torch.cuda.device(self.device):
            self.rvec0 = torch.as_tensor(np.ascontiguousarray(rvec0), dtype=f64).to(self.device)
            self.rindex_vec0 = torch.as_tensor(np.ascontiguousarray(rindex_vec0), dtype=f64).to(self.device)
            # zero-filled once, like initialize_ray_results_m (ray_results_m.f90:154-164)
            self.ray_vec = torch.zeros((self.nray, npt, nv), dtype=f64, device=self.device)
            self.residual = torch.zeros((self.nray, npt), dtype=f64, device=self.device)
            self.npoints = torch.zeros(self.nray, dtype=i32, device=self.device)
            self.stop_code = torch.zeros(self.nray, dtype=i32, device=self.device)
            self.end_ray_vec = torch.zeros((self.nray, nv), dtype=f64, device=self.device)
            self.end_residuals = torch.zeros(self.nray, dtype=f64, device=self.device)
            self.max_residuals = torch.zeros(self.nray, dtype=f64, device=self.device)

    def launch(self, zero_fill: bool = True):
        t = self.torch
        stream = t.cuda.current_stream(self.device).cuda_stream
        hip.trace_device(self.params, self.nray, self.rvec0.data_ptr(), self.rindex_vec0.data_ptr(),
                         self.ray_vec.data_ptr(), self.residual.data_ptr(), self.npoints.data_ptr(),
                         self.stop_code.data_ptr(), self.end_ray_vec.data_ptr(),
                         self.end_residuals.data_ptr(), self.max_residuals.data_ptr(),
                         stream=stream, zero_fill=zero_fill)

    def results(self) -> RayResults:
        self.torch.cuda.synchronize(self.device)
        c = lambda x: x.cpu().numpy()
        return RayResults(c(self.ray_vec), c(self.residual), c(self.npoints), c(self.stop_code),
                          c(self.end_ray_vec), c(self.end_residuals), c(self.max_residuals))
