"""GPU tier: edge cases the reference exercises (empty / ragged fans, rays that never start, step
and length limits), the lane-refill path (more rays than resident lanes), and size-independent
properties on the full 64k-ray BASELINE fan."""
import numpy as np
import pytest

from rays_amd import hip
from rays_amd.params import copy_params
from tests import oracle_lib
from tests.common import load_golden

pytestmark = pytest.mark.gpu


def _same(out, ora):
    np.testing.assert_array_equal(out["npoints"], ora["npoints"])
    np.testing.assert_array_equal(out["stop_code"], ora["stop_code"])
    np.testing.assert_array_equal(out["ray_vec"], ora["ray_vec"])      # RK4/cold: bit-identical
    np.testing.assert_array_equal(out["residual"], ora["residual"])
    np.testing.assert_array_equal(out["end_ray_vec"], ora["end_ray_vec"])
    np.testing.assert_array_equal(out["end_residuals"], ora["end_residuals"])
    np.testing.assert_array_equal(out["max_residuals"], ora["max_residuals"])


def test_empty_fan():
    g, nml, p = load_golden("cfg1_slab16_rk4")
    out = hip.trace_host(p, np.zeros((0, 3)), np.zeros((0, 3)), ngpu=1)
    assert out["npoints"].shape == (0,)


@pytest.mark.parametrize("nray", [1, 63, 65, 257])
def test_ragged_fan_sizes(nray):
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][:nray], g["rindex_vec0_full"][:nray]
    _same(hip.trace_host(p, r0, n0, ngpu=1), oracle_lib.trace(p, r0, n0))


def test_rays_that_never_start_and_limits():
    g, nml, p = load_golden("cfg1_slab16_rk4")
    r0, n0 = g["rvec0"].copy(), g["rindex_vec0"].copy()
    r0[3, 0] = 10.0          # launched outside the box
    n0[5] *= 3.0             # k far off the dispersion surface: initial check_save stops the ray
    _same(hip.trace_host(p, r0, n0, ngpu=1), oracle_lib.trace(p, r0, n0))
    q = copy_params(p)
    q.nstep_max = 0          # ' nstep > nstep_max' on the first trip
    _same(hip.trace_host(q, r0, n0, ngpu=1), oracle_lib.trace(q, r0, n0))
    q = copy_params(p)
    q.s_max = 10.5 * p.ds    # 'sout > s_max' after 10 steps
    out = hip.trace_host(q, r0, n0, ngpu=1)
    _same(out, oracle_lib.trace(q, r0, n0))
    assert (out["stop_code"][[0, 1, 2]] == 1).all() and (out["npoints"][[0, 1, 2]] == 11).all()


@pytest.mark.parametrize("nstep_max", [6, 7, 8, 9, 16, 65])
def test_point_window_group_boundaries(nstep_max):
    """The RK4 kernel's LDS point window (rays_trace.hpp: PointWindow) with rays ending before, on and
    after its eight-point groups; 1024 rays cover every sector phase of ray_vec and residual."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    q = copy_params(p)
    q.nstep_max = nstep_max
    r0, n0 = g["rvec0_full"].copy(), g["rindex_vec0_full"].copy()
    r0[3, 0] = 10.0
    out, ora = hip.trace_host(q, r0, n0, ngpu=1), oracle_lib.trace(q, r0, n0)
    for k in ("ray_vec", "residual", "npoints", "stop_code", "end_ray_vec"):
        np.testing.assert_array_equal(out[k], ora[k])


def test_arcl_parameter_and_eq_gradients():
    """ray_param = 'arcl' and integrate_eq_gradients (nv = 12): v(7) tracks s, gradient rows track B."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    q = copy_params(p)
    q.ray_param = 0
    q.ds = 0.01
    q.s_max = 3.0
    q.nstep_max = 200
    r0, n0 = g["rvec0"][:8], g["rindex_vec0"][:8]
    out, ora = hip.trace_host(q, r0, n0, ngpu=1), oracle_lib.trace(q, r0, n0)
    _same(out, ora)
    n = out["npoints"][0]
    np.testing.assert_allclose(out["ray_vec"][0, :n, 6], np.arange(n) * q.ds, rtol=1e-6, atol=1e-9)
    q.integrate_eq_gradients = 1
    q.nv = 12
    out, ora = hip.trace_host(q, r0, n0, ngpu=1), oracle_lib.trace(q, r0, n0)
    np.testing.assert_array_equal(out["npoints"], ora["npoints"])
    np.testing.assert_array_equal(out["ray_vec"], ora["ray_vec"])


def test_lane_refill_more_rays_than_resident_lanes():
    """70000 rays > 65536 resident lanes: finished lanes pull new rays from the global counter."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    q = copy_params(p)
    q.nstep_max = 40
    reps = 70000 // 1024 + 1
    r0 = np.tile(g["rvec0_full"], (reps, 1))[:70000]
    n0 = np.tile(g["rindex_vec0_full"], (reps, 1))[:70000]
    out = hip.trace_host(q, r0, n0, ngpu=1)
    ora = oracle_lib.trace(q, r0[:1024], n0[:1024])
    for b in range(0, 70000, 1024):
        m = min(1024, 70000 - b)
        np.testing.assert_array_equal(out["npoints"][b:b + m], ora["npoints"][:m])
        np.testing.assert_array_equal(out["ray_vec"][b:b + m], ora["ray_vec"][:m])


def test_full_64k_fan_properties():
    """BASELINE headline size (65536-ray Solovev fan, RK4, cold): size-independent properties."""
    import os
    from rays_amd.trace import DeviceTrace, RaysRun
    from tests.common import ROOT

    run = RaysRun.from_namelist(os.path.join(ROOT, "configs", "cfg3b_solovev64k_rk4.in"))
    assert run.nray == 65536
    p = run.params
    tr = DeviceTrace(p, run.rvec0, run.rindex_vec0)
    tr.launch()
    a = tr.results()
    tr.launch(zero_fill=False)      # deterministic: a second pass over the same buffers is identical
    b = tr.results()
    np.testing.assert_array_equal(a.ray_vec, b.ray_vec)
    np.testing.assert_array_equal(a.npoints, b.npoints)
    npt = a.npoints
    assert npt.min() >= 2 and npt.max() <= p.nstep_max + 1
    assert set(np.unique(a.stop_code)) <= {2, 20, 21, 30, 40, 41}
    idx = np.arange(p.nstep_max + 1)[None, :]
    live = idx < npt[:, None]
    assert not a.ray_vec[~live].any() and not a.residual[~live].any()        # zero past npoints
    assert np.isfinite(a.ray_vec[live]).all()
    assert a.residual[live].max() <= p.dispersion_resid_limit                # recorded points pass check_save
    s7 = a.ray_vec[..., 6]
    assert (np.diff(s7, axis=1)[live[:, 1:]] > 0).all()                      # arc length grows
    assert a.total_steps == int((npt - 1).sum())
    # every 128th ray against the CPU oracle, bit for bit
    sel = np.arange(0, 65536, 128)
    ora = oracle_lib.trace(p, run.rvec0[sel], run.rindex_vec0[sel])
    np.testing.assert_array_equal(a.npoints[sel], ora["npoints"])
    np.testing.assert_array_equal(a.stop_code[sel], ora["stop_code"])
    np.testing.assert_array_equal(a.ray_vec[sel], ora["ray_vec"])
    np.testing.assert_array_equal(a.end_ray_vec[sel], ora["end_ray_vec"])


def test_pack_unpack_roundtrip_on_device():
    """The multi-GPU exchange payload: unpack(pack(slab)) == slab, packed size = sum(npoints)."""
    import torch
    from rays_amd.trace import DeviceTrace

    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    tr = DeviceTrace(p, g["rvec0_full"][:300], g["rindex_vec0_full"][:300])
    tr.launch()
    torch.cuda.synchronize()
    n, nv = tr.nray, p.nv
    npts = tr.npoints
    off = torch.cumsum(npts, 0, dtype=torch.int64) - npts
    total = int(npts.sum().item())
    pv = torch.full((total, nv), float("nan"), dtype=torch.float64, device="cuda")
    pr = torch.full((total,), float("nan"), dtype=torch.float64, device="cuda")
    hip.pack_device(n, nv, p.nstep_max, npts.data_ptr(), off.data_ptr(), tr.ray_vec.data_ptr(),
                    tr.residual.data_ptr(), pv.data_ptr(), pr.data_ptr())
    rv2, rs2 = torch.zeros_like(tr.ray_vec), torch.zeros_like(tr.residual)
    hip.unpack_device(n, nv, p.nstep_max, npts.data_ptr(), off.data_ptr(), pv.data_ptr(), pr.data_ptr(),
                      rv2.data_ptr(), rs2.data_ptr())
    torch.cuda.synchronize()
    assert not torch.isnan(pv).any() and not torch.isnan(pr).any()
    assert torch.equal(rv2, tr.ray_vec) and torch.equal(rs2, tr.residual)
    # packed rows are the rays' points in order
    r = 7
    o, m = int(off[r]), int(npts[r])
    assert torch.equal(pv[o:o + m], tr.ray_vec[r, :m])


def test_ode_step_returns_what_ode_solver_left_in_v():
    """rays_hip_ode_step_device (the ode_m interface): for a step that check_save refuses, v1 is the ADVANCED state
    (`call ode_solver` has updated v; trace_rays then does not record it: ray_tracing.f90:214-234) with the stop code
    -- not zeros.  dispersion_resid_limit is set between the residual at the launch points (~1e-16) and the
    residual one step on (~1e-12), so every ray passes its initial check and is refused after its first step; the
    oracle's end_ray_vec of a one-step trace is that state."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][:200], g["rindex_vec0_full"][:200]
    q = copy_params(p)
    q.nstep_max = 1
    base = oracle_lib.trace(q, r0, n0)
    assert (base["npoints"] == 2).all()
    lim = float(np.sqrt(base["residual"][:, 1].min() * 1e-16))      # between the two
    q.dispersion_resid_limit = lim
    ora = oracle_lib.trace(q, r0, n0)
    refused = (ora["npoints"] == 1) & (ora["stop_code"] == 40)      # 'dispersion_residual' after the step
    assert refused.sum() > 100
    v0 = base["ray_vec"][:, 0, :]
    got, resid, code = hip.ode_step(q, v0)
    np.testing.assert_array_equal(code, ora["stop_code"])
    np.testing.assert_array_equal(got[refused], ora["end_ray_vec"][refused])
    assert np.abs(got[refused] - v0[refused]).max() > 0             # advanced, not v0 and not zeros
    taken = code == 0
    np.testing.assert_array_equal(got[taken], base["ray_vec"][taken, 1, :])
