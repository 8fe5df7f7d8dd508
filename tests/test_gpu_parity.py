"""GPU parity tests proper: HIP path (through the C ABI) vs the reference golden vectors and vs
the CPU oracle on the same inputs."""
import numpy as np
import pytest

from rays_amd import hip
from tests import oracle_lib
from tests.common import GOLDEN_CASES, assert_matches_golden, load_golden

pytestmark = pytest.mark.gpu

# Every case is held to BIT equality with the reference: the kernels follow the reference's operation order
# in IEEE binary64 without contraction, and exp / pow are glibc's algorithms (rays_amd/csrc/rays_libm.hpp), so
# not even the libm users (gold_solovev64_pow_rk4, the Gaussian slab profile, the Z function, the
# Shampine-Gordon step-size update) differ in the last bit.
RK4_CASES = ["cfg1_slab16_rk4", "cfg2_solovev1024_rk4", "gold_solovev64_rk4_num", "gold_solovev64_pow_rk4",
             "gold_solovev_evanescent_rk4",
             "gold_solovev64_damp_rk4", "gold_axisym64_eqdsk_damp_rk4",
             "gold_slab_toroid_parab_arcl_grad_rk4", "gold_slab_lin2_rk4_num",
             "gold_axisym64_eqdsk_tspline_rk4_num", "gold_slab16_fast_rk4",
             "gold_slab_box_exits_rk4", "gold_slab_negative_temp_rk4", "gold_axisym16_eqdsk_zexit_rk4",
             "gold_slab_negative_dens_rk4", "gold_slab16_damp_rk4",
             "gold_slab_ns1_rk4", "gold_solovev64_damp_grad_rk4", "gold_solovev64_4spec_rk4_num",
             "gold_slab16_damp_multi_grad_rk4", "gold_axisym64_solmag_damp_rk4",
             "gold_axisym64_solmag_splines_grad_rk4", "gold_axisym64_eqlin_damp_rk4",
             "gold_slab_one_ray_rk4", "gold_solovev_file_rays_damp_rk4", "gold_axisym64_eqdsk129_tspline_damp_rk4"]
SG_CASES = ["gold_solovev64_sg_cold", "gold_solovev64_sg_num", "gold_solovev64_damp_sg", "gold_axisym64_eqdsk_damp_sg",
            "gold_solovev64_arcl_grad_sg", "gold_solovev64_slow_sg",
            "gold_slab_shear_gauss_3spec_sg_num", "gold_slab_6spec_sg", "gold_solovev64_damp_multi_sg",
            "gold_axisym64_solmag_sg_num", "gold_axisym64_eqlin_tspline_sg_num", "gold_axisym64_eqdsk129_tspline_damp_sg"]

def test_bit_exact_bar_is_the_bar_applied():
    """A green GPU run must mean the bit-exact bars were applied.  The kernels carry their own exp / pow, so their
    comparisons with the fixtures are exact on any host; the ORACLE (tests vs the oracle, smoke) calls the host's libm and
    is the reference's equal only where that is glibc's x86-64 FMA build -- asserted here, measured, not assumed."""
    from tests.common import host_libm_check_ran, host_libm_is_the_variant_the_fixtures_were_cut_with
    assert host_libm_check_ran(), "the host-libm comparison could not be built on this box"
    assert host_libm_is_the_variant_the_fixtures_were_cut_with(), \
        "this host's libm is not the variant the fixtures were cut with: oracle comparisons are not bit-exact here"


@pytest.mark.parametrize("name", RK4_CASES)
def test_rk4_matches_reference_golden(name):
    g, nml, p = load_golden(name)
    out = hip.trace_host(p, g["rvec0"], g["rindex_vec0"], ngpu=1)
    assert_matches_golden(out, g, p, exact=True)


@pytest.mark.parametrize("name", SG_CASES)
def test_sg_matches_reference_golden(name):
    """Shampine-Gordon adaptive stepper (+ numerical dD): npoints / stop flags incl. 'equations stiff',
    'ODE total error' and box exits, trajectories, residuals and summaries bit-identical to the reference."""
    g, nml, p = load_golden(name)
    out = hip.trace_host(p, g["rvec0"], g["rindex_vec0"], ngpu=1)
    assert_matches_golden(out, g, p, exact=True)


@pytest.mark.parametrize("name", SG_CASES)
def test_sg_full_fan_matches_oracle(name):
    g, nml, p = load_golden(name)
    out = hip.trace_host(p, g["rvec0_full"], g["rindex_vec0_full"], ngpu=1)
    np.testing.assert_array_equal(out["npoints"], g["npoints_full"])
    ora = oracle_lib.trace(p, g["rvec0_full"], g["rindex_vec0_full"])
    for k in ("stop_code", "npoints", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals"):
        np.testing.assert_array_equal(out[k], ora[k], err_msg=k)   # the whole fan, bit for bit


@pytest.mark.parametrize("name", ["cfg1_slab16_rk4", "cfg2_solovev1024_rk4", "gold_solovev64_pow_rk4",
                                  "gold_axisym64_eqdsk_damp_rk4"])
def test_device_functions_match_reference_probes(name):
    g, nml, p = load_golden(name)
    pr = g["probes"]
    from rays_amd.params import copy_params
    q = copy_params(p)          # the probe kernel evaluates the nv = 7 rows
    q.nv, q.damping_model = 7, 0
    dev = hip.probe(q, pr["v"][:, :7])
    for key in ("cold", "num", "dvds"):
        ref = pr[key][:, :7]
        np.testing.assert_array_equal(dev[key], ref, err_msg=key)   # NaN == NaN here
    np.testing.assert_array_equal(dev["resid"], pr["resid"])


def test_full_fan_matches_oracle():
    """Full cfg2 fan (1024 rays): exact counts/flags vs the reference, values vs the oracle."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    out = hip.trace_host(p, g["rvec0_full"], g["rindex_vec0_full"], ngpu=1)
    np.testing.assert_array_equal(out["npoints"], g["npoints_full"])
    ora = oracle_lib.trace(p, g["rvec0_full"], g["rindex_vec0_full"])
    np.testing.assert_array_equal(out["npoints"], ora["npoints"])
    np.testing.assert_array_equal(out["stop_code"], ora["stop_code"])
    d = np.abs(out["ray_vec"] - ora["ray_vec"])
    scale = np.maximum(np.abs(ora["ray_vec"]), 1e-30)
    print("cfg2 full fan: bitwise" if np.array_equal(out["ray_vec"], ora["ray_vec"]) else
          f"cfg2 full fan: max abs diff {d.max():.3e}")
    rel = np.linalg.norm(out["ray_vec"][..., :3] - ora["ray_vec"][..., :3], axis=-1) / \
        np.maximum(np.linalg.norm(ora["ray_vec"][..., :3], axis=-1), 1e-30)
    assert rel.max() < 1e-10
    np.testing.assert_allclose(out["end_ray_vec"], ora["end_ray_vec"], rtol=1e-10, atol=0, equal_nan=True)
    np.testing.assert_allclose(out["end_residuals"], ora["end_residuals"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["max_residuals"], ora["max_residuals"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_device_ray_init_matches_reference(name):
    """rays_hip_ray_init: the fan built on the GPU equals the reference launcher's
    rvec0 / rindex_vec0 bit for bit, in the same ray order."""
    from rays_amd.ray_init import fan_from_namelist, initialize_ray_init
    from tests.common import HOST_ONLY_LAUNCHERS, launcher_model
    g, nml, p = load_golden(name)
    if launcher_model(nml) in HOST_ONLY_LAUNCHERS:
        pytest.skip("host-side launcher (no rays_hip_ray_init model)")
    fan, nray_max = fan_from_namelist(nml)
    r0, n0, w = hip.ray_init_host(p, fan, nray_max)
    np.testing.assert_array_equal(r0, g["rvec0_full"])
    np.testing.assert_array_equal(n0, g["rindex_vec0_full"])
    tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
    _, _, w_host = initialize_ray_init(p, nml, tab or None)
    np.testing.assert_array_equal(w, w_host)


def test_device_ray_init_device_pointers_and_trace():
    """Device-pointer form feeding the trace directly: the fan never visits the host."""
    import torch
    from rays_amd.ray_init import fan_from_namelist
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    fan, nray_max = fan_from_namelist(nml)
    d_r = torch.zeros((nray_max, 3), dtype=torch.float64, device="cuda")
    d_n = torch.zeros((nray_max, 3), dtype=torch.float64, device="cuda")
    nray = hip.ray_init_device(p, fan, nray_max, d_r.data_ptr(), d_n.data_ptr())
    assert nray == int(g["nray_full"])
    np.testing.assert_array_equal(d_r[:nray].cpu().numpy(), g["rvec0_full"])
    np.testing.assert_array_equal(d_n[:nray].cpu().numpy(), g["rindex_vec0_full"])


def test_device_deposition_profiles_match_reference():
    """rays_hip_deposition_device on the trajectories of a device trace: work(n_bins, nray),
    the profiles and Q_sum equal the reference post-processor's bit for bit; splitting the fan in two
    blocks and chaining the partial sums (the multi-GPU exchange) gives the same bits."""
    import torch
    g, nml, p = load_golden("gold_axisym64_eqdsk_damp_rk4")
    from rays_amd.trace import DeviceTrace
    tr = DeviceTrace(p, g["rvec0_full"], g["rindex_vec0_full"])
    tr.launch()
    torch.cuda.synchronize()
    nray, nb = tr.nray, int(g["dep_n_bins"])
    np.testing.assert_array_equal(tr.npoints.cpu().numpy(), g["npoints_full"])
    hip.set_rho_table(g["dep_rho_grid"], g["dep_rho_fspl"])
    power = torch.as_tensor(g["dep_power"], device="cuda")
    for which, name in enumerate(("Ptotal_psi", "Ptotal_rho")):
        work = torch.zeros((nb, nray), dtype=torch.float64, device="cuda")   # bin-major
        prof = torch.zeros(nb, dtype=torch.float64, device="cuda")
        hip.deposition_device(p, name, nb, nray, tr.ray_vec.data_ptr(), tr.npoints.data_ptr(), power.data_ptr(),
                              work.data_ptr(), None, prof.data_ptr())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(work.cpu().numpy().T, g["dep_work"][which])
        np.testing.assert_array_equal(prof.cpu().numpy(), g["dep_profile"][which])
        q = 0.0
        for x in prof.cpu().numpy():
            q = q + x
        assert q == g["dep_q_sum"][which]
        # two consecutive ray blocks, partial sums chained
        h = nray // 2
        part, tot = torch.zeros(nb, dtype=torch.float64, device="cuda"), torch.zeros(nb, dtype=torch.float64, device="cuda")
        w2 = torch.zeros((nb, nray), dtype=torch.float64, device="cuda")
        hip.deposition_device(p, name, nb, h, tr.ray_vec.data_ptr(), tr.npoints.data_ptr(), power.data_ptr(),
                              w2.data_ptr(), None, part.data_ptr())
        hip.deposition_device(p, name, nb, nray - h, tr.ray_vec[h:].data_ptr(), tr.npoints[h:].data_ptr(),
                              power[h:].data_ptr(), w2.data_ptr(), part.data_ptr(), tot.data_ptr())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(tot.cpu().numpy(), g["dep_profile"][which])


@pytest.mark.parametrize("name", ["gold_axisym64_solmag_damp_rk4", "gold_axisym64_eqlin_damp_rk4"])
def test_device_other_magnetics_deposition_profile_matches_reference(name):
    """'Ptotal_psi' of an axisym_toroid run with the analytic 'solovev_magnetics' field / the bilinear
    'eqdsk_magnetics_lin_interp' model: the host entry rays_hip_deposition on the device trace's arrays equals the
    reference post-processor bit for bit ('Ptotal_rho' does not exist for these models in the reference)."""
    g, nml, p = load_golden(name)
    out = hip.trace_host(p, g["rvec0_full"], g["rindex_vec0_full"], ngpu=1)
    np.testing.assert_array_equal(out["npoints"], g["npoints_full"])
    work, prof = hip.deposition_host(p, "Ptotal_psi", int(g["dep_n_bins"]), out["ray_vec"], out["npoints"], g["dep_power"])
    np.testing.assert_array_equal(work, g["dep_work"][0])
    np.testing.assert_array_equal(prof, g["dep_profile"][0])
    with pytest.raises(hip.RaysHipError, match="rho is only implemented"):
        hip.deposition_host(p, "Ptotal_rho", int(g["dep_n_bins"]), out["ray_vec"], out["npoints"], g["dep_power"])


def test_device_slab_deposition_profile_matches_reference():
    """'Ptotal_x' (slab run with damping): work, profile and Q_sum of the reference post-processor bit for
    bit from the trajectories of a device trace; an axisym profile name is refused for a slab run."""
    import torch
    from rays_amd.trace import DeviceTrace
    g, nml, p = load_golden("gold_slab16_damp_rk4")
    tr = DeviceTrace(p, g["rvec0_full"], g["rindex_vec0_full"])
    tr.launch()
    torch.cuda.synchronize()
    nray, nb = tr.nray, int(g["dep_n_bins"])
    np.testing.assert_array_equal(tr.npoints.cpu().numpy(), g["npoints_full"])
    power = torch.as_tensor(g["dep_power"], device="cuda")
    work = torch.zeros((nb, nray), dtype=torch.float64, device="cuda")
    prof = torch.zeros(nb, dtype=torch.float64, device="cuda")
    hip.deposition_device(p, "Ptotal_x", nb, nray, tr.ray_vec.data_ptr(), tr.npoints.data_ptr(), power.data_ptr(),
                          work.data_ptr(), None, prof.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(work.cpu().numpy().T, g["dep_work"][0])
    np.testing.assert_array_equal(prof.cpu().numpy(), g["dep_profile"][0])
    q = 0.0
    for x in prof.cpu().numpy():
        q = q + x
    assert q == g["dep_q_sum"][0]
    with pytest.raises(hip.RaysHipError):
        hip.deposition_device(p, "Ptotal_psi", nb, nray, tr.ray_vec.data_ptr(), tr.npoints.data_ptr(), power.data_ptr(),
                              work.data_ptr(), None, prof.data_ptr())


def test_device_ray_init_all_evanescent_and_missing_rho_table():
    """Launcher error behaviour on the device path: a fan entirely past the cutoff ends with the
    reference's 'No successful ray initializations'."""
    from rays_amd.ray_init import fan_from_namelist
    g, nml, p = load_golden("gold_solovev_evanescent_rk4")
    fan, nray_max = fan_from_namelist(nml)
    far = type(fan).from_buffer_copy(fan)
    far.rindex_theta0, far.delta_rindex_theta = 5.0, 0.1     # |n_theta| >> 1 everywhere: evanescent
    with pytest.raises(hip.RaysHipError, match="No successful ray initializations"):
        hip.ray_init_host(p, far, nray_max)


def test_result_files_equal_reference(tmp_path):
    """namelist -> device trace -> finalize_run writes run_results.<label>; every array
    in it equals the reference's own file for the same namelist (tests/golden/run_results.gold_slab4_ld)."""
    import os
    from rays_amd import results as R
    from rays_amd.trace import RaysRun
    from tests.common import ROOT
    run = RaysRun.from_namelist(os.path.join(ROOT, "configs", "gold_slab4_results_ld.in"))
    run.finalize_run(run.trace_rays(), str(tmp_path))
    mine = R.read_results_LD(str(tmp_path / "run_results.ld4"))
    ref = R.read_results_LD(os.path.join(ROOT, "tests", "golden", "run_results.gold_slab4_ld"))
    for k in ("npoints", "initial_ray_power", "end_ray_parameter", "end_residuals", "max_residuals",
              "start_ray_vec", "end_ray_vec", "residual", "ray_vec"):
        np.testing.assert_array_equal(mine[k], ref[k], err_msg=k)
    assert mine["ray_stop_flag"] == ref["ray_stop_flag"]
    nc = R.read_results_NC(str(tmp_path / "run_results.ld4.nc"))
    np.testing.assert_array_equal(nc["ray_vec"], ref["ray_vec"])
    np.testing.assert_array_equal(nc["npoints"], ref["npoints"])


def test_deposition_of_the_last_trace_from_its_device_image():
    """rays_hip_keep_last_result + rays_hip_deposition_last: the profiles of the rays rays_hip_trace has just traced,
    binned from the slabs that call left on the device, against rays_hip_deposition on the host arrays of the same call
    (itself bit-identical to the reference post-processor) -- several blocks per device, so the running sums are carried
    from block to block; and the refusals: nothing held, another shape."""
    g, nml, p = load_golden("gold_axisym64_eqdsk_damp_rk4")
    r0, n0 = g["rvec0_full"], g["rindex_vec0_full"]
    power, nb = g["dep_power"], int(g["dep_n_bins"])
    hip.keep_last_result(False)
    assert hip.deposition_last(p, "Ptotal_psi", nb, power) is None          # nothing held
    prev = hip.keep_last_result(True)
    try:
        hip.init_devices([0, 0, 0])                                         # three blocks on the one device
        out = hip.trace_host(p, r0, n0, ngpu=None)
        np.testing.assert_array_equal(out["npoints"], g["npoints_full"])
        for which in ("Ptotal_psi", "Ptotal_rho"):
            if which == "Ptotal_rho":
                hip.set_rho_table(g["dep_rho_grid"], g["dep_rho_fspl"])
            w_host, prof_host = hip.deposition_host(p, which, nb, out["ray_vec"], out["npoints"], power)
            w_dev, prof_dev = hip.deposition_last(p, which, nb, power)
            np.testing.assert_array_equal(prof_dev, prof_host)
            np.testing.assert_array_equal(w_dev, w_host)
            i = list(g["dep_names"]).index(which)
            np.testing.assert_array_equal(prof_dev, g["dep_profile"][i])   # = the reference post-processor's
        assert hip.deposition_last(p, "Ptotal_psi", nb, power[:-1]) is None  # another nray: refused, not misread
    finally:
        hip.keep_last_result(prev)
        hip.load().rays_hip_init(1)
