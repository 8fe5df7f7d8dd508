"""Host-side image of the module state `trace_rays` reads.

Mirrors, value for value, what the reference's `initialize` leaves in
constants_m / species_m / rf_m / ode_m / SG_ode_m / slab_eq_m / solovev_eq_m
(RAYS_project/RAYS_lib/intialize.f90:50-75) and packs it into the POD ``rays_params_t`` of
include/rays_hip.h.  The reference assigns *single-precision literals* to doubles
(constants_m.f90:39-48); those are reproduced through ``np.float32`` so every constant handed to
the device is bit-identical to the Fortran host's.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict

import numpy as np

NS0 = 6  # species_m.f90:25 nspec0 = 5 -> arrays 0:5
ABI_VERSION = 3

ODE = {"RK4_ODE": 0, "SG_ODE": 1}
DERIV = {"cold": 0, "numerical": 1}
RAY_PARAM = {"arcl": 0, "time": 1}
EQUILIB = {"slab": 0, "solovev": 1, "axisym_toroid": 2}
AXI_MAGNETICS = {"eqdsk_magnetics_spline_interp": 0, "solovev_magnetics": 1, "eqdsk_magnetics_lin_interp": 2}
AXI_N = {"constant": 0, "parabolic": 1, "density_spline_interp": 2}
AXI_T = {"zero": 0, "constant": 1, "parabolic": 2, "temperature_spline_interp": 3}
SLAB_BX = {"zero": 0}
SLAB_BY = {"zero": 0, "constant": 1, "toroid": 2, "linear_shear": 3}
SLAB_BZ = {"constant": 0, "toroid": 1, "linear": 2, "linear_2": 3}
SLAB_N = {"constant": 0, "linear": 1, "linear_2": 2, "parabolic": 3, "Gaussian": 4}
SLAB_T = {"zero": 0, "constant": 1, "linear": 2, "linear_2": 3, "parabolic": 4}
SOLOVEV_N = {"constant": 0, "parabolic": 1}
SOLOVEV_T = {"zero": 0, "parabolic": 2}
DAMPING = {"no_damp": 0, "damp_fund_ECH": 1}

# per-ray stop codes <-> reference ode_stop_flag strings (include/rays_hip.h)
STOP_FLAG_TEXT = {
    0: "",
    1: "sout > s_max",
    2: " nstep > nstep_max",
    10: "x out_of_bounds",
    11: "y out_of_bounds",
    12: "z out_of_bounds",
    13: "negative_dens",
    14: "negative_temp",
    20: "R out_of_box",
    21: "z out_of_box",
    22: "R_out_of_box",
    23: "Z_out_of_box",
    24: "out_of_plasma",
    30: "infinite Vg",
    31: "ray stalled",
    40: "dispersion_residual",
    41: "infinite_Vg",
    42: "total_absorption",
    50: "ODE total error",
    51: "step number .ge. maxnum",
    52: "equations stiff",
    53: "t == tout",
    54: "relerr or abserr < 0",
    55: "eps <= 0",
}
STOP_CODE = {v: k for k, v in STOP_FLAG_TEXT.items()}


class SlabParams(C.Structure):
    _fields_ = [
        ("bx_prof_model", C.c_int32), ("by_prof_model", C.c_int32),
        ("bz_prof_model", C.c_int32), ("dens_prof_model", C.c_int32),
        ("t_prof_model", C.c_int32 * NS0), ("pad_", C.c_int32 * 2),
        ("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double),
        ("ymax", C.c_double), ("zmin", C.c_double), ("zmax", C.c_double),
        ("rmaj", C.c_double), ("rmin", C.c_double), ("x0", C.c_double),
        ("bx0", C.c_double), ("by0", C.c_double), ("bz0", C.c_double),
        ("LBy_shear_scale", C.c_double), ("LBz_scale", C.c_double), ("dBzdx", C.c_double),
        ("Ln_scale", C.c_double), ("dndx", C.c_double), ("alphan1", C.c_double),
        ("alphan2", C.c_double), ("n_min", C.c_double),
        ("LT_scale", C.c_double), ("dtdx", C.c_double),
        ("alphat1", C.c_double * NS0), ("alphat2", C.c_double * NS0), ("T_min", C.c_double * NS0),
    ]


class SolovevParams(C.Structure):
    _fields_ = [
        ("dens_prof_model", C.c_int32), ("t_prof_model", C.c_int32 * NS0), ("pad_", C.c_int32 * 1),
        ("rmaj", C.c_double), ("kappa", C.c_double), ("bphi0", C.c_double),
        ("iota0", C.c_double), ("outer_bound", C.c_double), ("psiB", C.c_double),
        ("alphan1", C.c_double), ("alphan2", C.c_double),
        ("alphat1", C.c_double * NS0), ("alphat2", C.c_double * NS0),
        ("box_rmin", C.c_double), ("box_rmax", C.c_double),
        ("box_zmin", C.c_double), ("box_zmax", C.c_double),
    ]


class AxisymParams(C.Structure):
    _fields_ = [
        ("magnetics_model", C.c_int32), ("density_prof_model", C.c_int32),
        ("t_prof_model", C.c_int32 * NS0),
        ("box_rmin", C.c_double), ("box_rmax", C.c_double),
        ("box_zmin", C.c_double), ("box_zmax", C.c_double),
        ("plasma_psi_limit", C.c_double), ("psiB", C.c_double),
        ("alphan1", C.c_double), ("alphan2", C.c_double),
        ("d_scrape_off", C.c_double), ("T_scrape_off", C.c_double),
        ("alphat1", C.c_double * NS0), ("alphat2", C.c_double * NS0),
    ]


class AxisymTables(C.Structure):
    """ctypes image of rays_axisym_tables_t."""

    _fields_ = [
        ("nr", C.c_int32), ("nz", C.c_int32), ("n_rb", C.c_int32), ("n_ne", C.c_int32),
        ("n_te", C.c_int32), ("n_ti", C.c_int32),
        ("r_grid", C.POINTER(C.c_double)), ("z_grid", C.POINTER(C.c_double)),
        ("psi_fspl", C.POINTER(C.c_double)),
        ("rb_grid", C.POINTER(C.c_double)), ("rb_fspl", C.POINTER(C.c_double)),
        ("ne_grid", C.POINTER(C.c_double)), ("ne_fspl", C.POINTER(C.c_double)),
        ("te_grid", C.POINTER(C.c_double)), ("te_fspl", C.POINTER(C.c_double)),
        ("ti_grid", C.POINTER(C.c_double)), ("ti_fspl", C.POINTER(C.c_double)),
    ]


def axisym_tables_struct(tab: Dict[str, Any]):
    """dict of numpy arrays (keys r_grid, z_grid, psi_fspl, rb_grid, rb_fspl[, ne_*, te_*, ti_*]) ->
    (AxisymTables, keepalive list)."""
    t = AxisymTables()
    keep = []

    def put(name):
        a = tab.get(name)
        if a is None or len(a) == 0:
            return 0
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        setattr(t, name, a.ctypes.data_as(C.POINTER(C.c_double)))
        return len(a)

    t.nr, t.nz = put("r_grid"), put("z_grid")
    if "lin_psi" in tab:   # 'eqdsk_magnetics_lin_interp' (rays_hip_set_eqdsk_lin_tables): raw Psi(nr, nz), T(nr)
        for src, dst in (("lin_psi", "psi_fspl"), ("lin_t", "rb_fspl")):
            a = np.ascontiguousarray(tab[src], dtype=np.float64)
            keep.append(a)
            setattr(t, dst, a.ctypes.data_as(C.POINTER(C.c_double)))
        t.n_rb = t.nr
    else:
        put("psi_fspl")
        t.n_rb = put("rb_grid")
        put("rb_fspl")
    t.n_ne = put("ne_grid")
    put("ne_fspl")
    t.n_te = put("te_grid")
    put("te_fspl")
    t.n_ti = put("ti_grid")
    put("ti_fspl")
    return t, keep


RAY_INIT = {"solovev": 0, "axisym_toroid_ray_init_R_Z_nphi_ntheta": 1, "simple_slab": 2}
WAVE_MODE = {"plus": 0, "minus": 1, "fast": 2, "slow": 3}


class RaysFan(C.Structure):
    """rays_fan_t of include/rays_hip.h: the ray launcher's namelist + wave_mode / k0_sign."""
    _fields_ = [
        ("model", C.c_int32), ("wave_mode", C.c_int32), ("k0_sign", C.c_int32),
        ("n_r_launch", C.c_int32), ("n_theta_launch", C.c_int32),
        ("n_rindex_theta", C.c_int32), ("n_rindex_phi", C.c_int32),
        ("r_launch0", C.c_double), ("dr_launch", C.c_double),
        ("theta_launch0", C.c_double), ("dtheta_launch", C.c_double), ("z_launch0", C.c_double),
        ("rindex_theta0", C.c_double), ("delta_rindex_theta", C.c_double),
        ("rindex_phi0", C.c_double), ("delta_rindex_phi", C.c_double),
        ("n_x_launch", C.c_int32), ("n_y_launch", C.c_int32), ("n_z_launch", C.c_int32),
        ("n_ky_launch", C.c_int32), ("n_kz_launch", C.c_int32), ("pad_", C.c_int32),
        ("x_launch0", C.c_double), ("dx_launch", C.c_double), ("y_launch0", C.c_double),
        ("dy_launch", C.c_double), ("slab_z_launch0", C.c_double),
        ("rindex_y0", C.c_double), ("delta_rindex_y0", C.c_double),
        ("rindex_z0", C.c_double), ("delta_rindex_z0", C.c_double),
    ]


class RaysParams(C.Structure):
    """ctypes image of ``rays_params_t`` (include/rays_hip.h)."""

    _fields_ = [
        ("abi_version", C.c_int32), ("nv", C.c_int32), ("nspec", C.c_int32),
        ("nstep_max", C.c_int32), ("ode_solver", C.c_int32), ("ray_deriv", C.c_int32),
        ("ray_param", C.c_int32), ("equilib_model", C.c_int32),
        ("integrate_eq_gradients", C.c_int32), ("pad_", C.c_int32 * 3),
        ("ds", C.c_double), ("s_max", C.c_double),
        ("omgrf", C.c_double), ("k0", C.c_double),
        ("clight", C.c_double), ("eps0", C.c_double),
        ("dispersion_resid_limit", C.c_double),
        ("rel_err0", C.c_double), ("abs_err0", C.c_double), ("SG_error_limit", C.c_double),
        ("qs", C.c_double * NS0), ("ms", C.c_double * NS0),
        ("n0s", C.c_double * NS0), ("t0s", C.c_double * NS0), ("eta", C.c_double * NS0),
        ("slab", SlabParams), ("solovev", SolovevParams),
        ("damping_model", C.c_int32), ("multi_spec_damping", C.c_int32),
        ("total_damping_limit", C.c_double),
        ("axisym", AxisymParams),
    ]


def _f32(x: float) -> float:
    """A default-real (single precision) Fortran literal widened to double."""
    return float(np.float32(x))


class Constants:
    """constants_m.f90:36-60 -- note the single-precision literals."""

    pi = _f32(3.1415926535897932385)          # :39
    clight = _f32(2.997930e8)                 # :42
    mu0 = pi * _f32(4.0e-7)                   # :43
    eps0 = 1.0 / (mu0 * (clight * clight))    # :44
    me = _f32(9.1094e-31)                     # :46
    mp = _f32(1.6726e-27)                     # :47
    e = _f32(1.6022e-19)                      # :48


# species_m.f90:31-35
SPEC_NAME0 = ["electron", "hydrogen", "deuterium", "tritium", "3He", "alpha"]
QS0 = [-1.0, 1.0, 1.0, 1.0, 2.0, 2.0]
MS0 = [1.0, 1836.0, 3670.0, 5497.0, 5496.0, 7294.0]


def _arr(val: Any, n: int, default: Any) -> list:
    """Namelist array value (scalar | list | {index: v}) -> python list of length n (0-based)."""
    out = [default] * n
    if val is None:
        return out
    if isinstance(val, dict):
        for k, v in val.items():
            if 0 <= k < n:
                out[k] = v
    elif isinstance(val, (list, tuple)):
        for i, v in enumerate(val[:n]):
            out[i] = v
    else:
        out[0] = val
    return out


class ConfigError(ValueError):
    """A configuration the reference rejects with `stop 1` (or that the device path lacks)."""


def _lookup(table: Dict[str, int], name: Any, what: str) -> int:
    key = str(name).strip()
    if key not in table:
        raise ConfigError(f"invalid {what} = {key!r}")
    return table[key]


def params_from_namelist(nml: Dict[str, Dict[str, Any]], axisym_tables: Dict[str, Any] = None) -> RaysParams:
    """Build ``rays_params_t`` the way `initialize(read_input=.true.)` builds module state.

    equilib_model = 'axisym_toroid' needs the host-built spline tables (+ the eqdsk-derived box and
    psiB scalars) in ``axisym_tables``: coefficient generation is host-side initialisation in RAYS
    (initialize_eqdsk_magnetics_spline_interp) and is not re-implemented in this Python mirror."""
    p = RaysParams()
    p.abi_version = ABI_VERSION
    diag = nml.get("diagnostics_list", {})
    sp = nml.get("species_list", {})
    rf = nml.get("rf_list", {})
    damp = nml.get("damping_list", {})
    eql = nml.get("equilibrium_list", {})
    ode = nml.get("ode_list", {})
    sg = nml.get("sg_ode_list", {})

    # ---- species_m.f90:97-168 ------------------------------------------------------------
    eta = [float(x) for x in _arr(sp.get("eta"), NS0, 0.0)]
    names = [str(x).strip() for x in _arr(sp.get("spec_name"), NS0, "")]
    qs = [float(x) for x in _arr(sp.get("qs"), NS0, 0.0)]
    ms = [float(x) for x in _arr(sp.get("ms"), NS0, 0.0)]
    t0s_ev = [float(x) for x in _arr(sp.get("t0s_ev"), NS0, 0.0)]
    names[0], ms[0], qs[0], eta[0] = "electron", 1.0, -1.0, 1.0
    nspec = 0
    for i in range(1, NS0):
        if eta[i] > 0.0:
            nspec += 1
            for j in range(1, NS0):
                if names[nspec] == SPEC_NAME0[j]:
                    ms[nspec], qs[nspec] = MS0[j], QS0[j]
    charge = 0.0
    for i in range(nspec + 1):  # dot_product(qs(:nspec), eta(:nspec))
        charge += qs[i] * eta[i]
    if abs(charge) > float(sp.get("neutrality", _f32(1.0e-10))):
        raise ConfigError(f"charge neutrality violated, charge = {charge}")
    n0 = float(sp.get("n0", 0.0))
    p.nspec = nspec
    for i in range(NS0):
        p.ms[i] = Constants.me * ms[i]
        p.qs[i] = Constants.e * qs[i]
        p.n0s[i] = eta[i] * n0
        p.t0s[i] = Constants.e * t0s_ev[i]
        p.eta[i] = eta[i]

    # ---- rf_m.f90:57-95 ------------------------------------------------------------------
    frf = float(rf.get("frf", 0.0))
    if frf <= 0.0:
        raise ConfigError(f"initialize_rf: frf = {frf}")
    if str(rf.get("ray_dispersion_model", "cold")).strip() != "cold":
        raise ConfigError("check_save: unimplemented ray_dispersion_model")
    p.omgrf = 2.0 * Constants.pi * frf
    p.k0 = p.omgrf / Constants.clight
    p.clight, p.eps0 = Constants.clight, Constants.eps0
    p.ray_param = _lookup(RAY_PARAM, rf.get("ray_param", "arcl"), "ray parameter")
    p.dispersion_resid_limit = float(rf.get("dispersion_resid_limit", 0.0))

    # ---- damping_m.f90:46-68 ---------------------------------------------------------------
    p.damping_model = _lookup(DAMPING, damp.get("damping_model", "no_damp"), "damping model")
    p.multi_spec_damping = 1 if bool(damp.get("multi_spec_damping", False)) else 0
    if p.multi_spec_damping and not p.damping_model:
        # nv would grow by 1 + nspec rows that eqn_ray never sets (eqn_ray.f90:196-213 is inside the damping branch)
        raise ConfigError("multi_spec_damping needs a damping model (damping_model = 'no_damp' leaves its rows undefined)")
    p.total_damping_limit = float(damp.get("total_damping_limit", _f32(0.99)))

    # ---- equilibrium ---------------------------------------------------------------------
    p.equilib_model = _lookup(EQUILIB, eql.get("equilib_model", ""), "equilibrium model")
    if p.equilib_model == EQUILIB["slab"]:
        s = nml.get("slab_eq_list", {})
        q = p.slab
        q.bx_prof_model = _lookup(SLAB_BX, s.get("bx_prof_model", ""), "bx_prof_model")
        q.by_prof_model = _lookup(SLAB_BY, s.get("by_prof_model", ""), "by_prof_model")
        q.bz_prof_model = _lookup(SLAB_BZ, s.get("bz_prof_model", ""), "bz_prof_model")
        q.dens_prof_model = _lookup(SLAB_N, s.get("dens_prof_model", ""), "dens_prof_model")
        tm = _arr(s.get("t_prof_model"), NS0, "")
        for i in range(nspec + 1):
            q.t_prof_model[i] = _lookup(SLAB_T, tm[i], "t_prof_model")
        for name in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax", "rmaj", "rmin", "x0", "bx0",
                     "by0", "bz0", "LBy_shear_scale", "LBz_scale", "dBzdx", "Ln_scale", "dndx",
                     "alphan1", "alphan2", "n_min", "LT_scale", "dtdx"):
            setattr(q, name, float(s.get(name.lower(), 0.0)))
        for name in ("alphat1", "alphat2", "T_min"):
            vals = _arr(s.get(name.lower()), NS0, 0.0)
            for i in range(NS0):
                getattr(q, name)[i] = float(vals[i])
    elif p.equilib_model == EQUILIB["axisym_toroid"]:
        s = nml.get("axisym_toroid_eq_list", {})
        q = p.axisym
        q.magnetics_model = _lookup(AXI_MAGNETICS, s.get("magnetics_model", ""), "magnetics model")
        if q.magnetics_model == AXI_MAGNETICS["solovev_magnetics"]:
            # /solovev_magnetics_list/ (solovev_magnetics_m.f90:60-125): travels in p.solovev; the box and psiB
            # it hands to axisym_toroid_eq are repeated in p.axisym
            m = nml.get("solovev_magnetics_list", {})
            v = p.solovev
            for name in ("rmaj", "kappa", "bphi0", "iota0", "box_rmin", "box_rmax", "box_zmin", "box_zmax"):
                setattr(v, name, float(m.get(name, 0.0)))
            v.outer_bound = float(m.get("outer_boundary", 0.0))
            if v.outer_bound < v.rmaj or v.outer_bound >= math.sqrt(2.0) * v.rmaj:
                raise ConfigError("Inner boundary complex, outer_bound >=  sqrt2*rmaj")  # :99-103 (`stop`)
            bp0 = v.bphi0 * v.iota0
            t = v.outer_bound * v.outer_bound - v.rmaj * v.rmaj
            v.psiB = 0.5 * bp0 * (t * t) / (v.rmaj * v.rmaj) / 4.0                       # :106
            axisym_tables = dict(axisym_tables or {}, box_rmin=v.box_rmin, box_rmax=v.box_rmax,
                                 box_zmin=v.box_zmin, box_zmax=v.box_zmax, psiB=v.psiB)
        elif axisym_tables is None:
            raise ConfigError("equilib_model='axisym_toroid' needs axisym_tables (host-built spline tables)")
        q.density_prof_model = _lookup(AXI_N, s.get("density_prof_model", ""), "density_prof_model")
        tm = _arr(s.get("temperature_prof_model"), NS0, " ")
        for i in range(nspec + 1):
            q.t_prof_model[i] = _lookup(AXI_T, tm[i], "temperature_prof_model")
        q.plasma_psi_limit = float(s.get("plasma_psi_limit", 1.0))
        for name in ("alphan1", "alphan2", "d_scrape_off"):
            setattr(q, name, float(s.get(name, 0.0)))
        q.T_scrape_off = float(s.get("t_scrape_off", 0.0))
        for name in ("alphat1", "alphat2"):
            vals = _arr(s.get(name), NS0, 0.0)
            for i in range(NS0):
                getattr(q, name)[i] = float(vals[i])
        for name in ("box_rmin", "box_rmax", "box_zmin", "box_zmax", "psiB"):
            setattr(q, name, float(axisym_tables[name]))
    else:
        s = nml.get("solovev_eq_list", {})
        q = p.solovev
        q.dens_prof_model = _lookup(SOLOVEV_N, s.get("dens_prof_model", ""), "dens_prof_model")
        tm = _arr(s.get("t_prof_model"), NS0, "")
        for i in range(nspec + 1):
            q.t_prof_model[i] = _lookup(SOLOVEV_T, tm[i], "t_prof_model")
        for name in ("rmaj", "kappa", "bphi0", "iota0", "outer_bound", "alphan1", "alphan2",
                     "box_rmin", "box_rmax", "box_zmin", "box_zmax"):
            setattr(q, name, float(s.get(name, 0.0)))
        for name in ("alphat1", "alphat2"):
            vals = _arr(s.get(name), NS0, 0.0)
            for i in range(NS0):
                getattr(q, name)[i] = float(vals[i])
        # solovev_eq_m.f90:89-92
        bp0 = q.bphi0 * q.iota0
        t = q.outer_bound * q.outer_bound - q.rmaj * q.rmaj
        q.psiB = 0.5 * bp0 * (t * t) / (q.rmaj * q.rmaj) / 4.0

    # ---- ode_m.f90:113-178, SG_ode_m.f90:37-69 ---------------------------------------------
    p.ode_solver = _lookup(ODE, ode.get("ode_solver_name", ""), "ode solver")
    p.ray_deriv = _lookup(DERIV, ode.get("ray_deriv_name", ""), "ray_deriv_name")
    p.nstep_max = int(ode.get("nstep_max", 0))
    p.s_max = float(ode.get("s_max", 0.0))
    p.ds = float(ode.get("ds", 0.0))
    p.integrate_eq_gradients = 1 if bool(diag.get("integrate_eq_gradients", False)) else 0
    p.nv = (7 + (1 if p.damping_model else 0) + (1 + p.nspec if p.multi_spec_damping else 0)
            + (5 if p.integrate_eq_gradients else 0))  # ode_m.f90:160-173
    p.rel_err0 = float(sg.get("rel_err0", 0.0))
    p.abs_err0 = float(sg.get("abs_err0", 0.0))
    p.SG_error_limit = float(sg.get("sg_error_limit", _f32(0.1)))
    if p.ode_solver == ODE["SG_ODE"]:
        lim = _f32(1.0e-10)  # SG_ode_m.f90:63
        if p.rel_err0 < lim or p.abs_err0 < lim:
            raise ConfigError("initialize_SG_ode: rel_err0, abs_err0 too small")
    return p


def params_bytes(p: RaysParams) -> bytes:
    return bytes(memoryview(p))


def copy_params(p: RaysParams) -> RaysParams:
    q = RaysParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(RaysParams))
    return q
