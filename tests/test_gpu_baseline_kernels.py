"""GPU tier: the kernel builds the BASELINE configurations actually dispatch, at sizes where their
size-dependent machinery is in play, compared with the CPU oracle (never HIP with HIP):

  * cfg 4 / large fans: `rk4_trace_kernel_w2<...>` (two waves per SIMD, from 131072 rays on), slab and Solovev;
  * cfg 3 / cfg 5: the Shampine-Gordon kernels with more rays than resident lanes (lane refill);
  * the host entry with several slots per device (one host thread, stream and staging area per slot);
  * the fused `ds` scan (one launch) run by run against the oracle;
  * per-step parity: one output step restarted from every recorded reference point (north_star's
    "1e-10 relative per step") for the fixtures whose accumulated trajectories carry libm noise.
"""
import os

import numpy as np
import pytest

from rays_amd import hip
from rays_amd.namelist import read_namelist
from rays_amd.params import copy_params, params_from_namelist
from rays_amd.ray_init import fan_from_namelist
from tests import oracle_lib
from tests.common import ROOT, load_golden

pytestmark = pytest.mark.gpu

ARRAYS = ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals")


def _fan(cfg, overrides, tables=None):
    """Namelist -> (params, rvec0, rindex_vec0) with the fan built by the device launcher (bit-identical to
    the reference's, tests/test_gpu_parity.py::test_device_ray_init_matches_reference)."""
    nml = read_namelist(os.path.join(ROOT, "configs", cfg))
    for group, kv in overrides.items():
        nml[group].update(kv)
    p = params_from_namelist(nml, tables)
    fan, nray_max = fan_from_namelist(nml)
    r0, n0, _ = hip.ray_init_host(p, fan, nray_max)
    return p, r0, n0


def _assert_same(out, ora, keys=ARRAYS):
    for k in keys:
        np.testing.assert_array_equal(out[k], ora[k], err_msg=k)


def test_w2_slab_fan_matches_oracle():
    """BASELINE config 4's kernel: the cfg 4 slab fan at 363 x 363 = 131769 distinct rays (>= two waves per
    SIMD, ragged: 131769 = 514 * 256 + 185), 40 steps, bit for bit against the oracle."""
    p, r0, n0 = _fan("cfg4_slab1M_rk4.in", {
        "simple_slab_ray_init_list": dict(n_ky_launch=363, n_kz_launch=363, delta_rindex_y0=0.2 / 363,
                                          delta_rindex_z0=0.2 / 363),
        "ode_list": dict(nstep_max=40)})
    assert len(r0) == 363 * 363
    assert hip.kernel_name(p, len(r0)) == "rk4_trace_kernel_w2<4, 2, 0, 7>"
    out = hip.trace_host(p, r0, n0, ngpu=1)
    ora = oracle_lib.trace(p, r0, n0, nthreads=os.cpu_count() or 1)
    _assert_same(out, ora)
    assert out["npoints"].min() >= 2 and len(np.unique(out["ray_vec"][:, 1, 0])) > 1000


def test_w2_solovev_fan_matches_oracle():
    """The Solovev large-fan build `rk4_trace_kernel_w2<5, 2, 0, 7>` on a 363 x 363 fan of the headline
    configuration, 24 steps, bit for bit against the oracle."""
    p, r0, n0 = _fan("cfg3b_solovev64k_rk4.in", {
        "solovev_ray_init_nphi_ktheta_list": dict(n_rindex_theta=363, n_rindex_phi=363,
                                                  delta_rindex_theta=0.31 / 363, delta_rindex_phi=0.3875 / 363),
        "ray_init_list": dict(nray_max=363 * 363),
        "ode_list": dict(nstep_max=24)})
    assert len(r0) >= 131072
    assert hip.kernel_name(p, len(r0)) == "rk4_trace_kernel_w2<5, 2, 0, 7>"
    out = hip.trace_host(p, r0, n0, ngpu=1)
    ora = oracle_lib.trace(p, r0, n0, nthreads=os.cpu_count() or 1)
    _assert_same(out, ora)


def _device_trace_sampled(p, r0, n0, stride):
    """Trace the whole fan on the device, bring back only every `stride`-th ray (the full arrays of these fans are
    6 - 34 GB) plus npoints / stop codes of all rays."""
    import torch
    from rays_amd.trace import DeviceTrace
    tr = DeviceTrace(p, r0, n0)
    tr.launch()
    torch.cuda.synchronize()
    sel = np.arange(0, len(r0), stride)
    idx = torch.as_tensor(sel, device=tr.device)
    out = {k: getattr(tr, k).index_select(0, idx).cpu().numpy() for k in ARRAYS}
    npts, codes = tr.npoints.cpu().numpy(), tr.stop_code.cpu().numpy()
    # zero past npoints and finite before it, over the whole fan, checked on the device
    live = torch.arange(p.nstep_max + 1, device=tr.device)[None, :] < tr.npoints[:, None]
    assert not bool((tr.residual * (~live)).any()) and not bool((tr.ray_vec * (~live)[..., None]).any())
    assert bool(torch.isfinite(tr.residual).all())
    del tr
    torch.cuda.empty_cache()
    return sel, out, npts, codes


def test_w2_solovev_ragged_fan_with_refills_matches_oracle():
    """The two-waves-per-SIMD build with what the lock-step fans above never reach: rays of their natural, ragged
    lengths (99 .. 472 steps, nothing cut by nstep_max), more rays than resident lanes (512 x 400 = 204800 > 131072),
    so lanes are refilled in the middle of their neighbours' steps and the delayed start of rays_rk4_body.inc (a
    fresh ray waits for the lanes under way to come round to stage 3) is taken.  Every 256th ray against the
    oracle, bit for bit."""
    p, r0, n0 = _fan("cfg3b_solovev64k_rk4.in", {
        "solovev_ray_init_nphi_ktheta_list": dict(n_rindex_theta=512, n_rindex_phi=400,
                                                  delta_rindex_theta=0.32 / 512, delta_rindex_phi=0.4 / 400),
        "ray_init_list": dict(nray_max=512 * 400),
        "ode_list": dict(nstep_max=500)})
    assert len(r0) >= 200000
    assert hip.kernel_name(p, len(r0)) == "rk4_trace_kernel_w2<5, 2, 0, 7>"
    sel, out, npts, codes = _device_trace_sampled(p, r0, n0, 256)
    assert npts.max() - npts.min() > 200 and (codes == 2).mean() < 0.01       # natural lengths (ragged), not cut by nstep_max
    ora = oracle_lib.trace(p, r0[sel], n0[sel], nthreads=os.cpu_count() or 1)
    _assert_same(out, ora)


def test_cfg4_full_size_w2_sampled_against_oracle():
    """BASELINE config 4 at its full single-GPU size: the 1024 x 1024 = 1048576-ray slab fan, nstep_max = 500
    (33.5 GB of trajectories, sixteen rays per resident lane) on `rk4_trace_kernel_w2<4, 2, 0, 7>`; every 4096th
    ray against the oracle, bit for bit."""
    p, r0, n0 = _fan("cfg4_slab1M_rk4.in", {})
    assert len(r0) == 1024 * 1024
    assert hip.kernel_name(p, len(r0)) == "rk4_trace_kernel_w2<4, 2, 0, 7>"
    sel, out, npts, codes = _device_trace_sampled(p, r0, n0, 4096)
    assert int((npts.astype(np.int64) - 1).sum()) == 524242966               # steps per pass
    ora = oracle_lib.trace(p, r0[sel], n0[sel], nthreads=os.cpu_count() or 1)
    _assert_same(out, ora)


def _tiled(r0, n0, nray):
    reps = nray // len(r0) + 1
    return np.tile(r0, (reps, 1))[:nray].copy(), np.tile(n0, (reps, 1))[:nray].copy()


@pytest.mark.parametrize("which", ["cfg3", "cfg5"])
def test_sg_lane_refill_matches_oracle(which):
    """More rays than the SG kernels keep resident (65536 lanes): finished lanes pull new rays from the refill
    counter in the middle of other lanes' Adams steps.  163840 rays = 160 tiles of a 1024-ray fan, each tile
    compared with the oracle's trace of that fan.  cfg3: `sg_group_kernel<5, 2, 4>` (Solovev, finite-
    difference dD: one ray per group of four lanes, 163840 rays on <= 98304 resident groups); cfg5: `sg_trace_kernel<6, 2, 0, 8>` (eqdsk splines + ECH damping)."""
    if which == "cfg3":
        p, r0, n0 = _fan("cfg3_solovev64k_sg_num.in", {
            "solovev_ray_init_nphi_ktheta_list": dict(n_rindex_theta=32, n_rindex_phi=32,
                                                      delta_rindex_theta=0.01, delta_rindex_phi=0.0125),
            "ode_list": dict(nstep_max=5)})
        want = "sg_group_kernel<5, 2, 4>"
    else:
        g, nml0, p0 = load_golden("gold_axisym64_eqdsk129_tspline_damp_sg")   # hands cfg 5's eqdsk spline tables to hip + oracle
        tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
        p, r0, n0 = _fan("cfg5_axisym256k_sg_damp.in", {
            "axisym_toroid_ray_init_r_z_nphi_ntheta_list": dict(n_rindex_theta=32, n_rindex_phi=32,
                                                                delta_rindex_theta=0.01, delta_rindex_phi=0.0125),
            "ode_list": dict(nstep_max=5)}, tables=tab)
        want = "sg_trace_kernel<6, 2, 0, 8>"
    nml_rays = len(r0)
    assert nml_rays >= 512
    assert hip.kernel_name(p, 163840) == want
    R0, N0 = _tiled(r0, n0, 163840)
    out = hip.trace_host(p, R0, N0, ngpu=1)
    ora = oracle_lib.trace(p, r0, n0)
    assert (ora["npoints"] > 1).any()
    for b in range(0, 163840, nml_rays):
        m = min(nml_rays, 163840 - b)
        np.testing.assert_array_equal(out["npoints"][b:b + m], ora["npoints"][:m])
        np.testing.assert_array_equal(out["stop_code"][b:b + m], ora["stop_code"][:m])
        for k in ("ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals"):
            np.testing.assert_array_equal(out[k][b:b + m], ora[k][:m], err_msg=f"tile at ray {b}: {k}")


def test_host_entry_with_several_slots_per_device():
    """rays_hip_trace with more than one slot (rays_hip_init_devices): one host thread, stream, device buffer
    cache and pinned staging area per slot, all live at once.  Three slots on device 0, a ragged fan; equal
    to the one-slot result and to the oracle."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][:1001], g["rindex_vec0_full"][:1001]
    one = hip.trace_host(p, r0, n0, ngpu=1)
    ora = oracle_lib.trace(p, r0, n0)
    _assert_same(one, ora)
    for slots in ([0, 0, 0], [0, 0, 0, 0, 0, 0, 0]):
        hip.init_devices(slots)
        for _ in range(2):   # second call: cached buffers, streams and staging areas of every slot reused
            many = hip.trace_host(p, r0, n0, ngpu=None)
            _assert_same(many, ora)
    hip.load().rays_hip_init(1)


def test_fused_scan_one_launch_matches_oracle():
    """rays_hip_scan_device: five `ds` values x 128 rays in ONE launch; every run against
    the oracle's trace with that ds."""
    from rays_amd.scan import RayScan, scan_values
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][::8], g["rindex_vec0_full"][::8]
    vals = scan_values("fixed_increment", 5, p_start=float(p.ds) * 0.5, p_incr=float(p.ds) * 0.25)
    scan = RayScan(p, r0, n0, vals)
    scan.launch()
    res = scan.results()
    assert len({int(r.npoints.sum()) for r in res}) > 1
    for v, r in zip(vals, res):
        q = copy_params(p)
        q.ds = float(v)
        ora = oracle_lib.trace(q, r0, n0)
        for k in ARRAYS:
            np.testing.assert_array_equal(getattr(r, k), ora[k], err_msg=f"ds={v}: {k}")


# The libm users (profiles with exp / pow, the Z function, the SG step-size update) and the finite-difference dD
# that amplifies an ulp by 1e8: with the device library's own exp / pow two of these missed the per-step bar
# (2.7e-8, 2.7e-10); with rays_libm.hpp (glibc's algorithms) every restarted step is bit-identical.
PER_STEP_CASES = ["gold_slab_shear_gauss_3spec_sg_num", "gold_axisym64_eqdsk_tspline_rk4_num",
                  "gold_solovev64_sg_num", "gold_solovev64_arcl_grad_sg", "gold_solovev64_damp_sg",
                  "gold_axisym64_eqdsk_damp_sg", "gold_solovev64_pow_rk4", "gold_solovev64_rk4_num",
                  "gold_axisym64_eqdsk129_tspline_damp_sg"]


@pytest.mark.parametrize("name", PER_STEP_CASES)
def test_per_step_parity_from_reference_points(name):
    """north_star: "within 1e-10 relative per step".  Restart the HIP path at EVERY recorded reference point k
    (state and ray parameter from the fixture) for one output step (rays_hip_ode_step_device = ode_solver +
    check_save) and compare with the reference's point k+1, norm-wise on r and k like SURVEY App. A.  (SG:
    rel_err / abs_err at their initial values; the fixture rays never inflate them.)"""
    g, nml, p = load_golden(name)
    ref, npts = g["ray_vec"], g["npoints"]
    v0, v1, s0 = [], [], []
    for r in range(len(npts)):
        n = int(npts[r])
        if n < 2:
            continue
        s = np.concatenate([[0.0], np.cumsum(np.full(n - 1, float(p.ds)))])  # sout = sout + ds, sequentially
        v0.append(ref[r, :n - 1])
        v1.append(ref[r, 1:n])
        s0.append(s[:n - 1])
    v0, v1, s0 = np.concatenate(v0), np.concatenate(v1), np.concatenate(s0)
    got, resid, code = hip.ode_step(p, v0, s0)
    assert (code == 0).all(), f"{(code != 0).sum()} of {len(code)} restarted steps stopped: {np.unique(code)}"
    np.testing.assert_array_equal(got, v1)   # every restarted step lands on the reference's next point, bit for bit


@pytest.mark.parametrize("cfg,stride,kernel", [
    ("cfg3_solovev64k_sg_num.in", 128, "sg_group_kernel<5, 2, 4>"),
    ("cfg5_axisym256k_sg_damp.in", 512, "sg_trace_kernel<6, 2, 0, 8>"),
    ("cfg5b_axisym256k_rk4_damp.in", 512, "rk4_trace_kernel<6, 2, 0, 8>")])
def test_baseline_config_at_full_size_sampled_against_oracle(cfg, stride, kernel):
    """BASELINE configs 3 and 5 at their full sizes (65536 rays SG + finite-difference dD; 262144 rays eqdsk
    splines + damping, SG and RK4): every `stride`-th ray of the device result against the oracle's trace of
    just those rays, bit for bit, plus the size-independent properties."""
    from rays_amd.trace import DeviceTrace
    tab = None
    if "axisym" in cfg:
        g, _, _ = load_golden("gold_axisym64_eqdsk129_tspline_damp_sg")   # cfg 5's equilibrium file and profile splines
        tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
    p, r0, n0 = _fan(cfg, {}, tables=tab)
    assert hip.kernel_name(p, len(r0)) == kernel
    tr = DeviceTrace(p, r0, n0)
    tr.launch()
    a = tr.results()
    sel = np.arange(0, len(r0), stride)
    ora = oracle_lib.trace(p, r0[sel], n0[sel], nthreads=os.cpu_count() or 1)
    for k in ARRAYS:
        np.testing.assert_array_equal(getattr(a, k)[sel], ora[k], err_msg=k)
    live = np.arange(p.nstep_max + 1)[None, :] < a.npoints[:, None]
    assert not a.ray_vec[~live].any() and not a.residual[~live].any()
    assert np.isfinite(a.ray_vec[live]).all()


@pytest.mark.parametrize("cfg,overrides", [
    ("cfg5b_axisym256k_rk4_damp.in", {}),
    ("cfg3b_solovev64k_rk4.in", {"solovev_ray_init_nphi_ktheta_list": dict(n_rindex_theta=300, n_rindex_phi=333,
                                                                       delta_rindex_theta=0.32 / 300,
                                                                       delta_rindex_phi=0.4 / 333),
                                 "ray_init_list": dict(nray_max=300 * 333), "ode_list": dict(nstep_max=300)})])
def test_ray_hand_out_order_changes_nothing(monkeypatch, cfg, overrides):
    """Fans of more rays than lanes: the RK4 kernels hand the rays out "long rays first" (pilots, then two sweeps over
    their neighbourhoods: rays_trace.hpp take_rays).  A scheduling property, so it is checked as one: every ray is
    traced (npoints >= 1 everywhere -- a ray handed out to nobody would keep 0), and the complete result arrays are
    the same bits for index order and for neighbourhoods of 2 (the default), 4 and 8 rays -- incl. a fan whose size
    (99900) is no multiple of the block of 64 x 4 rays.  The oracle comparison of the same kernels at these sizes is
    test_baseline_config_at_full_size_sampled_against_oracle / the w2 tests above."""
    import torch
    from rays_amd.trace import DeviceTrace
    tab = None
    if "axisym" in cfg:
        g, _, _ = load_golden("gold_axisym64_eqdsk129_tspline_damp_sg")   # cfg 5's equilibrium file and profile splines
        tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
    p, r0, n0 = _fan(cfg, overrides, tables=tab)
    assert len(r0) > 65536 and "w2" not in hip.kernel_name(p, len(r0))   # more rays than the 65536 lanes of the launch
    ref = None
    for order in ("index", "pilot", "pilot4", "pilot8"):
        monkeypatch.setenv("RAYS_HIP_RAY_ORDER", order)
        tr = DeviceTrace(p, r0, n0)
        tr.launch()
        torch.cuda.synchronize()
        assert int(tr.npoints.min()) >= 1, order
        got = {k: getattr(tr, k) for k in ARRAYS}
        if ref is None:
            ref = got
        else:
            for k in ARRAYS:
                assert torch.equal(got[k].view(torch.int64) if got[k].dtype == torch.float64 else got[k],
                                   ref[k].view(torch.int64) if ref[k].dtype == torch.float64 else ref[k]), (order, k)
        del tr


def test_trace_gather_device_resident_result():
    """rays_hip_trace_gather: the library's own multi-GPU entry (blocks per device, RCCL gather to the root, result
    left in device memory).  On the one-GPU box it runs with one device: the root traces into the global arrays and
    no peer exists, so this covers the entry, the result block, its re-use across calls and rays_hip_result_to_host;
    a repeated device in the list is refused (RCCL: one rank per device)."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][:777], g["rindex_vec0_full"][:777]
    ora = oracle_lib.trace(p, r0, n0)
    hip.load().rays_hip_init(1)
    for _ in range(2):
        res, out = hip.trace_gather(p, r0, n0)
        assert res.nray == 777 and res.device == 0 and res.ray_vec
        _assert_same(out, ora)
    hip.init_devices([0, 0])
    with pytest.raises(hip.RaysHipError, match="must not repeat a device"):
        hip.trace_gather(p, r0, n0)
    hip.load().rays_hip_init(1)


@pytest.mark.parametrize("mapping,kernel", [("1", "sg_group_kernel<5, 2, 4>"), ("0", "sg_trace_kernel<5, 2, 1, 7>")])
def test_sg_num_both_mappings_match_oracle(monkeypatch, mapping, kernel):
    """SG + finite-difference dD (nv = 7) exists in two mappings: one ray per group of four lanes (rays_sg_group.hpp,
    the default: deriv_num's central differences and the ODE components spread over the group) and one ray per lane
    (rays_sg.hpp, RAYS_HIP_SG_GROUP=0).  3000 rays of the cfg 3 fan for twelve output intervals, incl. rays that
    never start, against the oracle, bit for bit -- both.  (The full 64k fan of both against each other and the
    oracle: tools/sg_group_check.py, profiles/r03/measurements/.)"""
    monkeypatch.setenv("RAYS_HIP_SG_GROUP", mapping)
    p, r0, n0 = _fan("cfg3_solovev64k_sg_num.in", {
        "solovev_ray_init_nphi_ktheta_list": dict(n_rindex_theta=60, n_rindex_phi=50,
                                                  delta_rindex_theta=0.32 / 60, delta_rindex_phi=0.4 / 50),
        "ode_list": dict(nstep_max=12)})
    assert hip.kernel_name(p, len(r0)) == kernel
    r0, n0 = r0.copy(), n0.copy()
    r0[7, 0] = 10.0      # outside the box
    n0[11] *= 3.0        # stops at the initial check_save
    out = hip.trace_host(p, r0, n0, ngpu=1)
    ora = oracle_lib.trace(p, r0, n0, nthreads=os.cpu_count() or 1)
    _assert_same(out, ora)


def test_sg_group_kernel_limits_and_scan():
    """The lane-group SG kernel at the limits trace_rays tests on every trip (ray_tracing.f90:118-172): nstep_max = 0
    (' nstep > nstep_max' before the first step), s_max inside the second interval ('sout > s_max'), and as the
    kernel of a fused `ds` scan (every group reads its run's ds where its ray starts) -- against the oracle."""
    from rays_amd.scan import RayScan, scan_values
    g, nml, p = load_golden("gold_solovev64_sg_num")
    r0, n0 = g["rvec0_full"][:40], g["rindex_vec0_full"][:40]
    assert hip.kernel_name(p, len(r0)) == "sg_group_kernel<5, 2, 4>"
    for change in (dict(nstep_max=0), dict(nstep_max=5, s_max=1.5 * float(p.ds)), dict(nstep_max=3)):
        q = copy_params(p)
        for k, v in change.items():
            setattr(q, k, v)
        out = hip.trace_host(q, r0, n0, ngpu=1)
        ora = oracle_lib.trace(q, r0, n0)
        _assert_same(out, ora)
    q = copy_params(p)
    q.nstep_max = 4
    vals = scan_values("fixed_increment", 3, p_start=float(p.ds) * 0.5, p_incr=float(p.ds) * 0.5)
    scan = RayScan(q, r0, n0, vals)
    scan.launch()
    for v, r in zip(vals, scan.results()):
        qq = copy_params(q)
        qq.ds = float(v)
        ora = oracle_lib.trace(qq, r0, n0)
        for k in ARRAYS:
            np.testing.assert_array_equal(getattr(r, k), ora[k], err_msg=f"ds={v}: {k}")
