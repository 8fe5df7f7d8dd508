"""Shared helpers for the parity tests."""
from __future__ import annotations

import os

import numpy as np

from rays_amd.namelist import read_namelist
from rays_amd.params import STOP_CODE, params_from_namelist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GOLDEN_CASES = ["cfg1_slab16_rk4", "cfg2_solovev1024_rk4", "gold_solovev64_rk4_num", "gold_solovev64_pow_rk4",
                "gold_solovev_evanescent_rk4",
                "gold_solovev64_sg_cold", "gold_solovev64_sg_num", "gold_solovev64_arcl_grad_sg", "gold_solovev64_slow_sg", "gold_slab16_fast_rk4", "gold_slab16_damp_rk4", "gold_slab_box_exits_rk4", "gold_slab_negative_temp_rk4", "gold_slab_negative_dens_rk4",
                "gold_solovev64_damp_rk4", "gold_solovev64_damp_sg", "gold_axisym64_eqdsk_damp_rk4",
                "gold_axisym64_eqdsk_damp_sg", "gold_axisym64_eqdsk_tspline_rk4_num", "gold_axisym16_eqdsk_zexit_rk4",
                "gold_slab_toroid_parab_arcl_grad_rk4", "gold_slab_shear_gauss_3spec_sg_num", "gold_slab_lin2_rk4_num",
                "gold_slab_ns1_rk4", "gold_solovev64_damp_grad_rk4", "gold_slab_6spec_sg", "gold_solovev64_4spec_rk4_num",
                "gold_solovev64_damp_multi_sg", "gold_slab16_damp_multi_grad_rk4",
                "gold_axisym64_solmag_damp_rk4", "gold_axisym64_solmag_sg_num",
                "gold_axisym64_solmag_splines_grad_rk4",
                "gold_axisym64_eqlin_damp_rk4", "gold_axisym64_eqlin_tspline_sg_num",
                "gold_axisym64_eqdsk129_tspline_damp_rk4", "gold_axisym64_eqdsk129_tspline_damp_sg",
                "gold_slab_one_ray_rk4", "gold_solovev_file_rays_damp_rk4"]

# launchers that take single rays by position and direction: host-side in every build (the reference's own routine under
# the Fortran host, rays_amd/ray_init.py under the Python one); rays_hip_ray_init has no model for them
HOST_ONLY_LAUNCHERS = ("one_ray_init_XYZ_n_direction", "file_input_ray_init")


def launcher_model(nml):
    return str(nml.get("ray_init_list", {}).get("ray_init_model", "")).strip()


def load_golden(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    nml = read_namelist(os.path.join(ROOT, "configs", str(g["config"])))
    tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
    # eqdsk equilibrium (or an analytic magnetics model with splined profiles): hand the host-built spline tables
    # to every implementation under test
    if any(np.size(tab.get(k, ())) for k in ("r_grid", "ne_grid", "te_grid", "ti_grid")):   # ("lin_psi": bilinear eqdsk model)
        from rays_amd import hip
        from tests import emul_lib, oracle_lib

        oracle_lib.set_axisym_tables(tab)
        emul_lib.set_axisym_tables(tab)
        hip.set_axisym_tables(tab)
    return g, nml, params_from_namelist(nml, tab or None)


def stop_codes(flags):
    return np.array([STOP_CODE[str(f)] for f in flags], dtype=np.int32)


_LIBM_OK = None
_LIBM_CHECK_RAN = False


def host_libm_is_the_variant_the_fixtures_were_cut_with():
    """The bit-exact bars of the libm users (profiles with exp / pow, the Z function, the SG step-size update) hold
    where the reference binary's libm is glibc's x86-64 FMA build, whose exp / pow rays_amd/csrc/rays_libm.hpp restates
    and with which tests/golden was generated.  A quick sample of tests/test_cpu_libm.py's comparison (2e5 arguments);
    cached.  False (another glibc, a CPU without FMA): the comparisons of the C ORACLE (which calls the host's libm) with
    the fixtures fall back to the documented tolerance bars instead of failing bit-wise on environment drift;
    the GPU tier asserts that it is True (tests/test_gpu_parity.py::test_bit_exact_bar_is_the_bar_applied), so a green GPU
    run means the bit-exact bars were the ones applied."""
    global _LIBM_OK, _LIBM_CHECK_RAN
    if _LIBM_OK is None:
        try:
            import ctypes as C
            import subprocess
            d = os.path.join(ROOT, "tests", "libm_check")
            lib = os.path.join(d, "liblibm_check.so")
            srcs = [os.path.join(d, "libm_check.cpp"), os.path.join(ROOT, "rays_amd", "csrc", "rays_libm.hpp"),
                    os.path.join(ROOT, "rays_amd", "csrc", "rays_libm_tables.inc")]
            if not os.path.exists(lib) or any(os.path.getmtime(x) > os.path.getmtime(lib) for x in srcs):
                subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", srcs[0],
                                       "-o", lib, "-lm"])
            l = C.CDLL(lib)
            dp = C.POINTER(C.c_double)
            l.check_exp_uniform.restype = l.check_pow_uniform.restype = C.c_longlong
            l.check_exp_uniform.argtypes = [C.c_longlong, C.c_uint64, C.c_double, C.c_double, dp]
            l.check_pow_uniform.argtypes = [C.c_longlong, C.c_uint64] + [C.c_double] * 4 + [dp]
            bad = (C.c_double * 2)()
            _LIBM_OK = (l.check_exp_uniform(100000, 1, -100.0, 30.0, bad) == 0 and
                        l.check_pow_uniform(100000, 2, 0.0, 1.0, 0.0, 3.0, bad) == 0)
            _LIBM_CHECK_RAN = True
        except Exception as e:   # no compiler here: assume the image the fixtures were cut on
            print(f"[tests.common] libm variant check unavailable ({e}); assuming glibc's FMA build")
            _LIBM_OK = True
        if not _LIBM_OK:
            print("[tests.common] this host's libm is not glibc's x86-64 FMA build: exact comparisons of fixtures fall "
                  "back to the tolerance bars (1e-10 per point on r, k; 1e-6 on the absorbed-power row)")
    return _LIBM_OK


def host_libm_check_ran():
    """True if the comparison above was really made on this host (False: no compiler, the answer was assumed)."""
    host_libm_is_the_variant_the_fixtures_were_cut_with()
    return _LIBM_CHECK_RAN


def assert_matches_golden(out, g, p, rel_tol=1e-10, resid_atol=1e-12, exact=False, calls_host_libm=False):
    """Parity bar (BASELINE.json north_star): exact ray counts / step indices / stop flags,
    trajectories within 1e-10 relative per step (norm-wise on r and k);
    residual with an absolute tolerance (it is a cancellation remainder)."""
    np.testing.assert_array_equal(out["npoints"], g["npoints"])
    np.testing.assert_array_equal(out["stop_code"], stop_codes(g["stop_flag"]))
    keep = g["ray_vec"].shape[1]
    rv, ref = out["ray_vec"][:, :keep, :], g["ray_vec"]
    assert not out["ray_vec"][:, keep:, :].any()
    # Only an implementation that calls THIS host's libm (the C oracle) may fall back to the tolerance bars when that
    # libm is not the variant the fixtures were cut with.  The HIP kernels and their host emulation carry their own
    # exp / pow (rays_libm.hpp): for them exact means exact on every host.
    if exact and calls_host_libm and not host_libm_is_the_variant_the_fixtures_were_cut_with():
        exact = False
    if exact:
        np.testing.assert_array_equal(rv, ref)
        np.testing.assert_array_equal(out["residual"][:, :keep], g["residual"])
        np.testing.assert_array_equal(out["end_ray_vec"], g["end_ray_vec"])
        return 0.0
    worst = 0.0
    for sl in (slice(0, 3), slice(3, 6)):
        num = np.linalg.norm(rv[..., sl] - ref[..., sl], axis=-1)
        den = np.linalg.norm(ref[..., sl], axis=-1)
        mask = den > 0
        worst = max(worst, float((num[mask] / den[mask]).max()))
    assert worst <= rel_tol, f"trajectory rel err {worst:.3e} > {rel_tol}"
    d7 = np.abs(rv[..., 6] - ref[..., 6])
    assert (d7 <= rel_tol * np.maximum(np.abs(ref[..., 6]), 1e-30) + 1e-300).all()
    damp = bool(p.damping_model)
    if damp:
        # v(8) = absorbed power fraction.  The reference carries k_i through single-precision COMPLEX
        # temporaries (damp_fund_ECH.f90:36: D_WARM, DELTA), and Im Z = sqrt(pi) exp(-xi^2) is a libm exp:
        # an ulp of difference between ocml's and glibc's exp can flip the float rounding, i.e. move k_i by
        # a single-precision ulp (6e-8).  Hence 1e-6 on this row only (observed: bit-identical for RK4).
        np.testing.assert_allclose(rv[..., 7], ref[..., 7], rtol=1e-6, atol=1e-9)
    g0 = 8 if damp else 7
    if rv.shape[-1] > g0:
        # integrate_eq_gradients rows (eqn_ray.f90:217-229): plain double precision like r and k, so the same
        # bar: 1e-10 relative to the row's magnitude along the ray (they cross zero, so not element-wise)
        d = np.abs(rv[..., g0:] - ref[..., g0:])
        scale = np.maximum(np.abs(ref[..., g0:]).max(axis=1, keepdims=True), 1e-300)
        gw = float((d / scale).max())
        assert gw <= rel_tol, f"equilibrium-gradient rows rel err {gw:.3e} > {rel_tol}"
        worst = max(worst, gw)
    np.testing.assert_allclose(out["residual"][:, :keep], g["residual"], rtol=0, atol=resid_atol)
    # zero beyond npoints (ray_results_m.f90:154-164)
    for r, n in enumerate(g["npoints"]):
        assert not out["ray_vec"][r, n:, :].any() and not out["residual"][r, n:].any()
    return worst


def padded_full_trajectories(g, p):
    """The full fan's trajectories of a fixture that carries them (dep_ray_vec_full), padded to the
    reference layout [nray][nstep_max+1][nv]."""
    rv = g["dep_ray_vec_full"]
    out = np.zeros((rv.shape[0], p.nstep_max + 1, p.nv))
    out[:, :rv.shape[1], :] = rv
    return out
