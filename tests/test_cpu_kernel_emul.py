"""CPU tier: the PRODUCT kernel source (rays_amd/csrc/rays_rk4.hpp, rays_sg.hpp, rays_device.hpp)
compiled for the host with a one-lane HIP emulation (tests/hip_emul) and compared with the
reference golden vectors bit for bit.  This covers the integrator state machines, stop logic,
the SG storage tiers, point recording and the summary fields without a GPU."""
import numpy as np
import pytest

from tests import emul_lib
from tests.common import GOLDEN_CASES, assert_matches_golden, load_golden


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_kernel_source_on_host_equals_reference(name):
    g, nml, p = load_golden(name)
    out = emul_lib.trace(p, g["rvec0"], g["rindex_vec0"])
    assert_matches_golden(out, g, p, exact=True)


@pytest.mark.parametrize("offsets", [(0, 0), (1, 7), (3, 4), (5, 2), (7, 1)])
def test_rk4_point_window_any_sector_alignment(offsets):
    """The RK4 kernel (nv = 7) passes recorded points through a per-lane window and writes whole 64-byte
    sectors of the GLOBAL address (rays_trace.hpp: PointWindow); a ray's slab starts anywhere within a
    sector.  Every placement of the two arrays must give the same trajectories, with nothing written
    outside a ray's points (1024 rays: all per-ray phases, all tail lengths, rays that stop at once)."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][:256].copy(), g["rindex_vec0_full"][:256].copy()
    r0[5, 0] = 10.0     # launched outside the box: npoints = 1
    n0[9] *= 3.0        # off the dispersion surface: stops at the initial check
    ref = emul_lib.trace(p, r0, n0)
    out = emul_lib.trace(p, r0, n0, vec_offset=offsets[0], res_offset=offsets[1])
    for k in ("ray_vec", "residual", "npoints", "stop_code", "end_ray_vec"):
        np.testing.assert_array_equal(out[k], ref[k])
    assert ref["npoints"][5] == 1 and ref["npoints"][9] == 1
    live = np.arange(p.nstep_max + 1)[None, :] < ref["npoints"][:, None]
    assert not out["ray_vec"][~live].any() and not out["residual"][~live].any()
    sub = [int(i) for i in g["ray_index"] if i < 256]
    np.testing.assert_array_equal(out["npoints"][sub], g["npoints"][:len(sub)])
    keep = g["ray_vec"].shape[1]
    np.testing.assert_array_equal(out["ray_vec"][sub, :keep], g["ray_vec"][:len(sub)])


@pytest.mark.parametrize("nstep_max", [0, 1, 6, 7, 8, 9, 15, 16, 17, 64, 200])
def test_rk4_point_window_group_boundaries(nstep_max):
    """Rays ending before, on and after the window's eight-point groups (rays_trace.hpp: PointWindow),
    against the C restatement, for two placements of the arrays within a sector."""
    from rays_amd.params import copy_params
    from tests import oracle_lib
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][:96].copy(), g["rindex_vec0_full"][:96].copy()
    r0[3, 0] = 10.0
    q = copy_params(p)
    q.nstep_max = nstep_max
    ora = oracle_lib.trace(q, r0, n0)
    for off in ((0, 0), (3, 5)):
        out = emul_lib.trace(q, r0, n0, vec_offset=off[0], res_offset=off[1])
        for k in ("ray_vec", "residual", "npoints", "stop_code", "end_ray_vec"):
            np.testing.assert_array_equal(out[k], ora[k])   # NaN == NaN here (rays ending on 'infinite_Vg')


@pytest.mark.parametrize("name", ["gold_solovev64_sg_cold", "gold_solovev64_sg_num", "gold_solovev64_damp_sg"])
def test_sg_storage_tiers(name):
    """The SG kernel keeps its coefficient vectors and divided differences in tiers (LDS / registers
    for the low orders, private memory above; rays_sg.hpp).  Built with the fast tiers shrunk to
    2 entries / 2 + 1 rows, every ray crosses every tier boundary; the result must not change.
    The full 64-ray fan is used: a few of its rays reach order k = 10."""
    g, nml, p = load_golden(name)
    out = emul_lib.trace(p, g["rvec0"], g["rindex_vec0"], small_tiers=True)
    assert_matches_golden(out, g, p, exact=True)
    full_a = emul_lib.trace(p, g["rvec0_full"], g["rindex_vec0_full"])
    full_b = emul_lib.trace(p, g["rvec0_full"], g["rindex_vec0_full"], small_tiers=True)
    for k in ("ray_vec", "residual", "npoints", "stop_code", "end_ray_vec"):
        np.testing.assert_array_equal(full_a[k], full_b[k])
    np.testing.assert_array_equal(full_a["npoints"], g["npoints_full"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_ray_init_source_on_host_equals_reference(name):
    """Device-side ray initialisation: fan_member / solve_n1_vs_n2_n3 of
    rays_ray_init.hpp compiled for the host reproduce the reference launcher's fan bit for bit
    (rvec0, rindex_vec0 and the ray numbering after evanescent launches are dropped)."""
    from rays_amd.ray_init import fan_from_namelist
    from tests.common import HOST_ONLY_LAUNCHERS, launcher_model
    g, nml, p = load_golden(name)
    if launcher_model(nml) in HOST_ONLY_LAUNCHERS:
        pytest.skip("host-side launcher (no rays_hip_ray_init model)")
    fan, nray_max = fan_from_namelist(nml)
    r0, n0 = emul_lib.ray_init(p, fan, nray_max)
    np.testing.assert_array_equal(r0, g["rvec0_full"])
    np.testing.assert_array_equal(n0, g["rindex_vec0_full"])


def test_deposition_source_on_host_equals_reference():
    """Deposition profiles: deposit_ray (bin_a_ray + binner_real + the psi / rho
    evaluators) and the ray-ordered sum reproduce the reference's work(n_bins, nray), profile and Q_sum
    for Ptotal_psi and Ptotal_rho bit for bit."""
    from tests.common import padded_full_trajectories
    g, nml, p = load_golden("gold_axisym64_eqdsk_damp_rk4")
    rv = padded_full_trajectories(g, p)
    for which, name in enumerate(g["dep_names"]):
        assert str(name) == ("Ptotal_psi", "Ptotal_rho")[which]
        work, prof = emul_lib.deposition(p, which, int(g["dep_n_bins"]), rv, g["npoints_full"], g["dep_power"],
                                         g["dep_rho_grid"], g["dep_rho_fspl"])
        np.testing.assert_array_equal(work, g["dep_work"][which])
        np.testing.assert_array_equal(prof, g["dep_profile"][which])
        q = 0.0
        for x in prof:
            q = q + x
        assert q == g["dep_q_sum"][which]


@pytest.mark.parametrize("name", ["gold_axisym64_solmag_damp_rk4", "gold_axisym64_eqlin_damp_rk4"])
def test_other_magnetics_deposition_source_on_host_equals_reference(name):
    """'Ptotal_psi' of an axisym_toroid run with magnetics_model = 'solovev_magnetics' (axisym_toroid_psi ->
    solovev_magnetics_psi, solovev_magnetics_m.f90:199-244) or 'eqdsk_magnetics_lin_interp'
    (eqdsk_magnetics_lin_interp_psi: GetPsi / PSIBOUND): work, profile and Q_sum bit for bit.  ('Ptotal_rho'
    does not exist for these magnetics models in the reference, axisym_toroid_eq_m.f90:398-430.)"""
    from tests.common import padded_full_trajectories
    g, nml, p = load_golden(name)
    assert [str(n) for n in g["dep_names"]] == ["Ptotal_psi"]
    rv = padded_full_trajectories(g, p)
    work, prof = emul_lib.deposition(p, 0, int(g["dep_n_bins"]), rv, g["npoints_full"], g["dep_power"],
                                     np.zeros(1), np.zeros(4))
    np.testing.assert_array_equal(work, g["dep_work"][0])
    np.testing.assert_array_equal(prof, g["dep_profile"][0])
    q = 0.0
    for x in prof:
        q = q + x
    assert q == g["dep_q_sum"][0] and q > 0.1


def test_slab_deposition_source_on_host_equals_reference():
    """'Ptotal_x' of a slab run with damping (Ptotal_x_slab_evaluator, deposition_profiles_m.f90:438-452;
    grid = the slab box in x): work, profile and Q_sum bit for bit."""
    from tests.common import padded_full_trajectories
    g, nml, p = load_golden("gold_slab16_damp_rk4")
    assert [str(n) for n in g["dep_names"]] == ["Ptotal_x"]
    rv = padded_full_trajectories(g, p)
    z = np.zeros(1)
    work, prof = emul_lib.deposition(p, 2, int(g["dep_n_bins"]), rv, g["npoints_full"], g["dep_power"], z, np.zeros(4))
    np.testing.assert_array_equal(work, g["dep_work"][0])
    np.testing.assert_array_equal(prof, g["dep_profile"][0])
    q = 0.0
    for x in prof:
        q = q + x
    assert q == g["dep_q_sum"][0] and q > 0.05   # 16 rays x 0.99 absorbed x weight 1/256 (App. A-11)


def test_fused_scan_launch_equals_oracle_per_run():
    """rays_hip_scan_device's launch (one fan of n_runs x nray rays, the run sets ds; rays_trace.hpp: start_ray)
    through the kernel source on the host: every run equals the C restatement's trace with that ds."""
    from rays_amd.params import copy_params
    from rays_amd.scan import scan_values
    from tests import oracle_lib
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][::64], g["rindex_vec0_full"][::64]
    vals = scan_values("fixed_increment", 3, p_start=float(p.ds) * 0.5, p_incr=float(p.ds) * 0.5)
    out = emul_lib.scan(p, r0, n0, vals)
    assert len({int(out["npoints"][r].sum()) for r in range(3)}) > 1
    for r, v in enumerate(vals):
        q = copy_params(p)
        q.ds = float(v)
        ora = oracle_lib.trace(q, r0, n0)
        for k in ("ray_vec", "residual", "npoints", "stop_code", "end_ray_vec", "end_residuals", "max_residuals"):
            np.testing.assert_array_equal(out[k][r], ora[k], err_msg=f"run {r}: {k}")


@pytest.mark.parametrize("name", ["cfg2_solovev1024_rk4", "gold_solovev64_sg_cold", "gold_solovev64_damp_rk4",
                                  "gold_solovev64_arcl_grad_sg"])
def test_one_step_restart_reproduces_the_next_reference_point(name):
    """rays_hip_ode_step_device's launch on the host (glibc libm, like the reference): restarted from any
    recorded reference point with its ray parameter, one output step lands on the next recorded point bit for
    bit -- RK4 and SG (whose integrator is restarted every output interval, SG_ode_m.f90:106-122)."""
    g, nml, p = load_golden(name)
    ref, npts = g["ray_vec"], g["npoints"]
    v0, v1, s0 = [], [], []
    for r in range(min(len(npts), 6)):
        n = int(npts[r])
        if n < 2:
            continue
        s = np.concatenate([[0.0], np.cumsum(np.full(n - 1, float(p.ds)))])
        v0.append(ref[r, :n - 1]); v1.append(ref[r, 1:n]); s0.append(s[:n - 1])
    v0, v1, s0 = np.concatenate(v0), np.concatenate(v1), np.concatenate(s0)
    got, resid, code = emul_lib.ode_step(p, v0, s0)
    assert (code == 0).all()
    np.testing.assert_array_equal(got, v1)


def test_eqdsk_lin_interp_outermost_cells_are_memory_safe():
    """GetPsi does not bound its cell index (eqdsk_utilities_m.f90:152-153): the central differences of a point in
    the outermost cells reach one cell beyond R_grid / Z_grid, where the reference reads the neighbouring column of
    Psi through the storage order and, past the last row, memory outside the array.  The kernel source and the C
    restatement follow the storage order and clamp only what leaves the array: same results, no fault, at the four
    corners and edge midpoints of the box."""
    from tests import oracle_lib
    g, nml, p = load_golden("gold_axisym64_eqlin_damp_rk4")
    a = p.axisym
    eps = 1e-4
    pts = [(r, z) for r in (a.box_rmin + eps, 0.5 * (a.box_rmin + a.box_rmax), a.box_rmax - eps)
           for z in (a.box_zmin + eps, 0.0, a.box_zmax - eps)]
    r0 = np.array([[r, 0.0, z] for r, z in pts])
    n0 = np.tile(g["rindex_vec0"][0], (len(pts), 1))
    ora = oracle_lib.trace(p, r0, n0)
    out = emul_lib.trace(p, r0, n0)
    for k in ("ray_vec", "residual", "npoints", "stop_code", "end_ray_vec"):
        np.testing.assert_array_equal(out[k], ora[k])
    assert (ora["npoints"] >= 1).all()


def test_rk4_hand_over_of_ill_conditioned_steps_runs_and_changes_nothing():
    """The tolerance kernels do not commit a step whose stage ran into dD/dw -> 0: the ray ends for them with an internal
    stop code and rk4_resume_kernel (an exact translation unit) takes the step again and runs the ray to its end
    (rays_rk4_body.inc: kStopResumeExact; rays_rk4.hpp: rk4_resume_ray).  On the host both arithmetics are the
    reference's, so the whole fan must come out bit-identical to the oracle -- and the path must have RUN: every ray of
    the cfg 2 fan ends in such a step.  (The GPU tier measures what it buys: tests/test_gpu_numerics_full_fans.py.)"""
    import ctypes
    from tests import oracle_lib
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    lib = emul_lib.lib()
    lib.rays_emul_redo_steps.restype = ctypes.c_longlong
    lib.rays_emul_redo_steps(1)
    r0, n0 = g["rvec0_full"], g["rindex_vec0_full"]
    out = emul_lib.trace(p, r0, n0)
    n_handed = lib.rays_emul_redo_steps(1)
    ora = oracle_lib.trace(p, r0, n0, nthreads=8)
    for k in ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals"):
        np.testing.assert_array_equal(out[k], ora[k], err_msg=k)
    assert (out["stop_code"] < 1000).all(), "the internal hand-over code leaked"
    assert 0.5 * len(r0) < n_handed <= len(r0), f"{n_handed} of {len(r0)} rays handed over"
    print(f"rays handed over to rk4_resume_ray: {n_handed} of {len(r0)}")


def test_rk4_resume_traces_whole_rays_like_the_reference(tmp_path):
    """rk4_resume_ray (what rk4_resume_kernel runs) is a third statement of trace_rays' loop (after the kernels' state
    machine and the oracle).  Built with a hand-over ratio of 2 every step's last stage "collapses", so every ray is handed
    over at its FIRST step and the resume code traces it from point 1 to its end: all seven result arrays must be the
    oracle's, bit for bit -- Solovev fan (rays ending at the mode coalescence, in the box wall, at nstep_max), a slab
    fixture with damping rows (nv = 8), and a fused `ds` scan's per-run steps."""
    import ctypes
    from tests import oracle_lib
    out_lib = str(tmp_path / "librays_emul_handover_all.so")
    emul_lib.build(out=out_lib, defs=["-DRAYS_RK4_HANDOVER_RATIO=2.0"])
    saved = emul_lib._lib
    emul_lib._lib = emul_lib._load(out_lib)
    try:
        lib = emul_lib._lib
        lib.rays_emul_redo_steps.restype = ctypes.c_longlong
        for name, take in (("cfg2_solovev1024_rk4", slice(0, 1024, 8)), ("gold_slab16_damp_rk4", slice(None)),
                           ("gold_solovev_evanescent_rk4", slice(None))):
            g, nml, p = load_golden(name)
            r0 = g["rvec0_full"][take] if "rvec0_full" in g.files else g["rvec0"][take]
            n0 = g["rindex_vec0_full"][take] if "rindex_vec0_full" in g.files else g["rindex_vec0"][take]
            lib.rays_emul_redo_steps(1)
            out = emul_lib.trace(p, r0, n0)
            handed = lib.rays_emul_redo_steps(1)
            ora = oracle_lib.trace(p, r0, n0, nthreads=8)
            for k in ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals"):
                np.testing.assert_array_equal(out[k], ora[k], err_msg=f"{name}: {k}")
            started = int((ora["npoints"] > 1).sum())
            assert handed >= started, f"{name}: {handed} rays handed over, {started} take at least one step"
    finally:
        emul_lib._lib = saved
