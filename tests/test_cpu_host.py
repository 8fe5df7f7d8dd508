"""CPU tier: host logic (namelist, module-state image, ray launchers) and the C-ABI library."""
import ctypes as C
import os

import numpy as np
import pytest

from rays_amd import hip
from rays_amd.namelist import parse_namelist
from rays_amd.params import STOP_FLAG_TEXT, ConfigError, RaysParams, params_from_namelist
from rays_amd.ray_init import initialize_ray_init
from tests.common import GOLDEN_CASES, ROOT, load_golden


def test_namelist_parser():
    nml = parse_namelist("""
 &a_list  x=1, y = 2.5e3 name='it''s' ! comment
   flag=.true. arr(0)= 1. arr(2) = 3.d0 rep=2*'zero' dup = 1 dup = 2
 /
 &b_list z = -1 /
""")
    a = nml["a_list"]
    assert a["x"] == 1 and a["y"] == 2500.0 and a["name"] == "it's" and a["flag"] is True
    assert a["arr"] == {0: 1.0, 2: 3.0} and a["rep"] == ["zero", "zero"] and a["dup"] == 2
    assert nml["b_list"]["z"] == -1


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_module_state_equals_reference(name):
    """constants_m / rf_m / species_m / solovev psiB as the reference computed them (bitwise)."""
    g, nml, p = load_golden(name)
    omgrf, k0, clight, eps0 = g["consts"]
    assert (p.omgrf, p.k0, p.clight, p.eps0) == (omgrf, k0, clight, eps0)
    for key in ("qs", "ms", "n0s", "t0s"):
        np.testing.assert_array_equal(np.array(getattr(p, key)[:]), g[key])
    if p.equilib_model == 1:
        assert p.solovev.psiB == float(g["psiB"])
    if p.equilib_model == 2:
        assert p.nv == 8 + 5 * p.integrate_eq_gradients - (p.damping_model == 0) and p.axisym.psiB == float(g["axi_psiB"])
        if p.axisym.magnetics_model == 1:   # 'solovev_magnetics': psiB of solovev_magnetics_m.f90:106
            assert p.solovev.psiB == float(g["axi_psiB"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_ray_init_equals_reference(name):
    """simple_slab / solovev n_theta x n_phi launchers: same rays, same order, same bits."""
    g, nml, p = load_golden(name)
    if p.equilib_model == 2 and p.axisym.magnetics_model != 0:
        pytest.skip("'solovev_magnetics' / 'eqdsk_magnetics_lin_interp' have no Python host mirror: the fan comes from the device launcher "
                    "(test_cpu_kernel_emul.py::test_ray_init_source_on_host_equals_reference, test_gpu_parity.py)")
    tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
    r0, n0, _ = initialize_ray_init(p, nml, tab or None, base_dir=os.path.join(ROOT, "configs"))
    assert len(r0) == int(g["nray_full"])
    np.testing.assert_array_equal(r0, g["rvec0_full"])
    np.testing.assert_array_equal(n0, g["rindex_vec0_full"])


def test_evanescent_launches_are_dropped():
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    import copy
    nml = copy.deepcopy(nml)
    nml["solovev_ray_init_nphi_ktheta_list"].update(n_rindex_phi=8, rindex_phi0=0.5, delta_rindex_phi=0.2)
    r0, n0, _ = initialize_ray_init(p, nml)
    assert 0 < len(r0) < 32 * 8  # large n_phi is evanescent at the launch point


def test_config_errors():
    g, nml, p = load_golden("cfg1_slab16_rk4")
    import copy
    bad = copy.deepcopy(nml)
    bad["ode_list"]["ode_solver_name"] = "EULER"
    with pytest.raises(ConfigError):
        params_from_namelist(bad)
    bad = copy.deepcopy(nml)
    bad["damping_list"]["damping_model"] = "damp_landau"   # damping_m.f90:103-106 `stop 1`
    with pytest.raises(ConfigError):
        params_from_namelist(bad)


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    lib = hip.load()
    header = open(os.path.join(ROOT, "include", "rays_hip.h")).read()
    import re
    declared = set(re.findall(r"\b(rays_hip_\w+)\s*\(", header))
    assert declared == set(hip.EXPORTED_SYMBOLS), declared ^ set(hip.EXPORTED_SYMBOLS)
    for sym in declared:
        getattr(lib, sym)
    assert lib.rays_hip_sizeof_params() == C.sizeof(RaysParams)


def test_c_abi_host_side_without_compute():
    """Parameter validation, kernel lookup and the stop-flag table need no GPU."""
    for name in GOLDEN_CASES:
        g, nml, p = load_golden(name)
        hip.check_params(p)
        assert hip.kernel_name(p).startswith(("rk4_trace_kernel<", "sg_trace_kernel<", "sg_group_kernel<"))
    # kernel selection: unit profile exponents -> EQ | 4 (no pow code), else the general kernels
    assert hip.kernel_name(load_golden("cfg2_solovev1024_rk4")[2]) == "rk4_trace_kernel<5, 2, 0, 7>"
    assert hip.kernel_name(load_golden("gold_solovev64_pow_rk4")[2]) == "rk4_trace_kernel<1, 2, 0, 7>"
    assert hip.kernel_name(load_golden("gold_solovev64_sg_num")[2]) == "sg_group_kernel<5, 2, 4>"
    assert hip.kernel_name(load_golden("gold_axisym64_eqdsk_damp_rk4")[2]) == "rk4_trace_kernel<6, 2, 0, 8>"
    for code, text in STOP_FLAG_TEXT.items():
        assert hip.stop_flag_text(code) == text
    assert hip.stop_flag_text(2) == " nstep > nstep_max"  # leading blank (ray_tracing.f90:152)
    g, nml, p = load_golden("cfg1_slab16_rk4")
    from rays_amd.params import copy_params
    q = copy_params(p)
    q.nv = 9
    with pytest.raises(hip.RaysHipError):
        hip.check_params(q)
    q = copy_params(p)
    q.abi_version = 99
    with pytest.raises(hip.RaysHipError):
        hip.check_params(q)


def test_product_fails_loudly_without_gpu():
    lib = hip.load()
    if lib.rays_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    g, nml, p = load_golden("cfg1_slab16_rk4")
    with pytest.raises(hip.RaysHipError):
        hip.trace_host(p, g["rvec0"], g["rindex_vec0"])


def test_scan_values_match_scanner_m():
    """ray_scan parameter sequences (scanner_m.f90:115-160)."""
    from rays_amd.scan import scan_values
    np.testing.assert_array_equal(scan_values("fixed_increment", 4, p_start=2.5e-11, p_incr=1.25e-11),
                                  2.5e-11 + np.arange(4) * 1.25e-11)
    d = float(np.float32(1.0e-14))
    np.testing.assert_array_equal(scan_values("pwr_of_2", 3, p_max=1.0), np.array([0.25, 0.5, 1.0]) - d)
    np.testing.assert_array_equal(scan_values("integer_divide", 3, p_max=1.0, max_divide=4),
                                  1.0 / np.array([4.0, 3.0, 2.0]))
    with pytest.raises(ValueError):
        scan_values("nope", 2)


def test_ray_launcher_and_deposition_config_errors_need_no_gpu():
    """rays_hip_ray_init / rays_hip_deposition_device reject what the reference's launchers and
    post-processor `stop 1` on, before touching a device."""
    import ctypes as C
    from rays_amd.params import copy_params
    from rays_amd.ray_init import fan_from_namelist
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    fan, nray_max = fan_from_namelist(nml)
    bad = type(fan).from_buffer_copy(fan)
    bad.wave_mode = 7
    with pytest.raises(hip.RaysHipError, match="wave_mode"):
        hip.ray_init_host(p, bad, nray_max)
    with pytest.raises(hip.RaysHipError, match="improper number of rays"):
        hip.ray_init_host(p, fan, 10)                      # nray_max < n_r*n_theta*n_ntheta*n_nphi
    slab = type(fan).from_buffer_copy(fan)
    slab.model = 2                                         # simple_slab launcher on a Solovev equilibrium
    with pytest.raises(hip.RaysHipError, match="slab"):
        hip.ray_init_host(p, slab, nray_max)
    # deposition: the post-processor knows slab and axisym_toroid runs (deposition_profiles_m.f90:129-222) with damping
    with pytest.raises(hip.RaysHipError, match="unimplemented equilib_model"):
        hip.deposition_device(p, "Ptotal_psi", 100, 1, 1, 1, 1, 1, None, 1)
    gs, nmls, ps = load_golden("gold_slab16_damp_rk4")
    with pytest.raises(hip.RaysHipError, match="unimplemented profile"):
        hip.deposition_device(ps, "Ptotal_rho", 100, 1, 1, 1, 1, 1, None, 1)
    ga, nmla, pa = load_golden("gold_axisym64_eqdsk_damp_rk4")
    q = copy_params(pa)
    q.nv, q.damping_model = 7, 0
    with pytest.raises(hip.RaysHipError, match="damping"):
        hip.deposition_device(q, "Ptotal_psi", 100, 1, 1, 1, 1, 1, None, 1)
    gm, nmlm, pm = load_golden("gold_axisym64_solmag_damp_rk4")   # analytic magnetics: no rho(psiN) in the reference
    with pytest.raises(hip.RaysHipError, match="rho is only implemented"):
        hip.deposition_device(pm, "Ptotal_rho", 100, 1, 1, 1, 1, 1, None, 1)


def test_bench_gpus_n_is_never_a_silent_one_gpu_run():
    """`bench.py --gpus N` (VERDICT r01 / ADVICE): without a launcher it starts its own N ranks and refuses when
    fewer than N devices are visible; under a launcher that started another number of ranks it refuses too."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RAYS_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must start exactly --gpus ranks" in (r.stderr + r.stdout)


def test_eqdsk_reader_builds_the_reference_lin_interp_tables():
    """rays_amd/eqdsk.py (ReadgFile + initialize_eqdsk_magnetics_lin_interp restated for the Python host): grids, dR,
    dZ, Psi - PSIAXIS, T, box and psiB equal the reference's eqdsk_utilities_m arrays (dumped into the fixture) bit for
    bit, and RaysRun-style loading yields the fixture's parameters."""
    from rays_amd.eqdsk import eqdsk_lin_tables
    g, nml, p = load_golden("gold_axisym64_eqlin_damp_rk4")
    t = eqdsk_lin_tables(os.path.join(ROOT, "configs", "solovev_65x65.geqdsk"))
    for k in ("r_grid", "z_grid", "lin_psi", "lin_t", "lin_dR", "lin_dZ", "box_rmin", "box_rmax", "box_zmin", "box_zmax", "psiB"):
        np.testing.assert_array_equal(np.asarray(t[k]), np.asarray(g["axi_" + k]), err_msg=k)
    from rays_amd.trace import load_axisym_tables
    tab = load_axisym_tables(os.path.join(ROOT, "configs", str(g["config"])), nml)
    np.testing.assert_array_equal(tab["ne_fspl"], g["axi_ne_fspl"])     # the splined density rides along from the tables file
    q = params_from_namelist(nml, tab)
    assert bytes(q) == bytes(p)
