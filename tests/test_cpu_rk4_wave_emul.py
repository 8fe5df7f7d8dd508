"""CPU tier: the one-ray-per-lane RK4 kernel on whole emulated waves (tests/hip_emul/hip/hip_wave_emul.h: 64 lanes as
fibers, ballots are rendezvous).  A single emulated lane cannot exercise what the wave does with lanes whose ray has
ended (rays_rk4_body.inc: they are parked, and written out / refilled in one batched pass of the wave), so this is
where that control flow is compared with the oracle, bit for bit, for every setting of the batching threshold."""
import numpy as np
import pytest

from rays_amd.params import copy_params
from tests import group_emul_lib as ge
from tests import oracle_lib
from tests.common import load_golden

ARRAYS = ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals")
# RAYS_REFILL_EVENT_COST (idle lane-trips that trigger a pass): the default and "at the next stage-3 trip"
VARIANTS = {"default": [], "cost0": ["-DRAYS_REFILL_EVENT_COST=0"]}


def _lib(variant):
    return ge.lib() if variant == "default" else ge.lib_variant(variant, VARIANTS[variant])


def _tables(g, library):
    tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
    if any(np.size(tab.get(k, ())) for k in ("r_grid", "ne_grid", "te_grid", "ti_grid")):
        ge.set_axisym_tables(tab, library)


def _check(out, ora):
    for k in ARRAYS:
        np.testing.assert_array_equal(out[k], ora[k], err_msg=k)


@pytest.mark.parametrize("stride", [0, 4, 3, 16])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_solovev_fan_with_refills(variant, stride):
    """200 rays of the Solovev fan at their natural, ragged lengths (99..380 steps) on ONE wave: every lane is
    refilled two or three times, rays end on different trips, the last pass finds the counter dry.  stride > 1: the rays
    are handed out "long rays first" (rays_trace.hpp: take_rays -- pilots, then two sweeps over their neighbourhoods; with
    stride 16 there are fewer pilots than lanes and most lanes start in the first sweep), every ray exactly once."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    r0, n0 = g["rvec0_full"][::5][:200].copy(), g["rindex_vec0_full"][::5][:200].copy()
    n0[7] *= 3.0   # far off the dispersion surface: stops at its initial check
    ora = oracle_lib.trace(p, r0, n0)
    assert len(set(ora["npoints"].tolist())) > 20 and ora["npoints"][7] == 1
    _check(ge.trace_rk4_waves(p, r0, n0, nwaves=1, library=_lib(variant), stride=stride), ora)
    if stride == 0:  # the two-waves-per-SIMD build's body (its own loop structure and hand-out)
        _check(ge.trace_rk4_waves(p, r0, n0, nwaves=1, library=_lib(variant), w2_body=True), ora)


@pytest.mark.parametrize("variant", ["default", "cost0"])
def test_no_more_rays_than_lanes(variant):
    """50 rays on one wave, 100 on two: nothing to refill, ended rays stay parked until their wave has no lane under
    way; 14 lanes (28 of the second wave) never hold a ray."""
    g, nml, p = load_golden("cfg2_solovev1024_rk4")
    q = copy_params(p)
    q.nstep_max = 150
    for n, nw in ((50, 1), (100, 2)):
        r0, n0 = g["rvec0_full"][::9][:n], g["rindex_vec0_full"][::9][:n]
        _check(ge.trace_rk4_waves(q, r0, n0, nwaves=nw, library=_lib(variant)), oracle_lib.trace(q, r0, n0))


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_slab_box_exits_two_waves(variant):
    """Rays that leave the box after a few steps (and one that starts outside it), tiled to 300 rays on two waves."""
    g, nml, p = load_golden("gold_slab_box_exits_rk4")
    reps = -(-300 // len(g["rvec0_full"]))
    r0, n0 = np.tile(g["rvec0_full"], (reps, 1))[:300], np.tile(g["rindex_vec0_full"], (reps, 1))[:300]
    ora = oracle_lib.trace(p, r0, n0)
    assert len(set(ora["stop_code"].tolist())) > 1
    _check(ge.trace_rk4_waves(p, r0, n0, nwaves=2, library=_lib(variant)), ora)
    _check(ge.trace_rk4_waves(p, r0, n0, nwaves=2, library=_lib(variant), stride=4), ora)
    _check(ge.trace_rk4_waves(p, r0, n0, nwaves=2, library=_lib(variant), w2_body=True), ora)


def test_eqdsk_damping_fan_with_refills():
    """nv = 8 (absorbed power row; residual(:) alone passes through the LDS window): 150 short rays on one wave."""
    g, nml, p = load_golden("gold_axisym64_eqdsk_damp_rk4")
    library = _lib("default")
    _tables(g, library)
    reps = -(-150 // len(g["rvec0_full"]))
    r0, n0 = np.tile(g["rvec0_full"], (reps, 1))[:150], np.tile(g["rindex_vec0_full"], (reps, 1))[:150]
    q = copy_params(p)
    q.nstep_max = min(q.nstep_max, 60)
    ora = oracle_lib.trace(q, r0, n0)
    _check(ge.trace_rk4_waves(q, r0, n0, nwaves=1, library=library), ora)
    _check(ge.trace_rk4_waves(q, r0, n0, nwaves=1, library=library, stride=4), ora)
