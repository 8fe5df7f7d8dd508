"""GPU tier of tests/test_cpu_profile_grids.py: the profile splines on three grids, on two (Te and Ti share one) and on one,
through the HIP library (both numerics flavours of the eqdsk RK4 kernel and the Shampine-Gordon kernel) against the oracle."""
import numpy as np
import pytest

from rays_amd import hip
from tests import oracle_lib
from tests.common import load_golden
from tests.test_cpu_profile_grids import ARRAYS, _resample, _tables

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fixture", ["gold_axisym64_eqdsk129_tspline_damp_rk4", "gold_axisym64_eqdsk129_tspline_damp_sg"])
@pytest.mark.parametrize("mode", ["three_grids", "te_ti_share", "one_grid"])
def test_profile_lookups_whatever_grids_they_share(mode, fixture):
    g, nml, p = load_golden(fixture)
    tab = _tables(g)
    fine = np.linspace(0.0, 1.0, 41)
    grids = {"three_grids": (tab["ne_grid"], tab["te_grid"], np.linspace(0.0, 1.0, 17)),
             "te_ti_share": (tab["ne_grid"], tab["te_grid"], tab["ti_grid"]),
             "one_grid": (fine, fine, fine)}[mode]
    for name, grid in zip(("ne", "te", "ti"), grids):
        tab[name + "_fspl"] = _resample(tab[name + "_grid"], tab[name + "_fspl"], grid)
        tab[name + "_grid"] = np.ascontiguousarray(grid, dtype=np.float64)
    prev = hip.get_numerics()
    try:
        oracle_lib.set_axisym_tables(tab)
        hip.set_axisym_tables(tab)
        r0, n0 = g["rvec0_full"], g["rindex_vec0_full"]
        ora = oracle_lib.trace(p, r0, n0)
        assert ora["npoints"].max() > 5
        hip.set_numerics("exact")
        out = hip.trace_host(p, r0, n0, ngpu=1)
        for k in ARRAYS:
            np.testing.assert_array_equal(out[k], ora[k], err_msg=k)
        if p.ode_solver == 0:   # RK4: the tolerance flavour -- counts exact, points within its bar of the exact trajectory
            hip.set_numerics("tolerance")
            tol = hip.trace_host(p, r0, n0, ngpu=1)
            np.testing.assert_array_equal(tol["npoints"], ora["npoints"])
            np.testing.assert_array_equal(tol["stop_code"], ora["stop_code"])
            live = np.arange(ora["ray_vec"].shape[1])[None, :] < ora["npoints"][:, None]
            a, b = tol["ray_vec"][live][:, :6], ora["ray_vec"][live][:, :6]
            assert np.max(np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)) < 1e-9
    finally:
        hip.set_numerics(prev)
        load_golden(fixture)   # the fixture's own tables back into the libraries


def test_spline_lookups_on_and_beside_the_grid_lines():
    """States whose R or Z sits ON a grid line of the psi spline or one ulp beside it -- where the cell estimate of the one-trip
    search (rays_device_arith.inc: spl_guess) can be a cell off and the lookup settles and fetches again -- through the device
    RHS probe against the oracle's, bit for bit (dD/dx, dD/dk, dD/dw, dv/ds, residual, codes)."""
    from rays_amd.params import copy_params
    g, nml, p = load_golden("gold_axisym64_eqdsk129_tspline_damp_rk4")
    tab = _tables(g)
    q = copy_params(p)          # the probe kernel evaluates the nv = 7 rows
    q.nv, q.damping_model = 7, 0
    base = np.array(g["ray_vec"][0, 3, :7], dtype=np.float64)   # a state inside the plasma
    r_axis, z_axis = float(np.hypot(base[0], base[1])), float(base[2])
    states = []
    for grid, which in ((tab["r_grid"], 0), (tab["z_grid"], 2)):
        inner = np.asarray(grid[2:-2], dtype=np.float64)
        for x in np.concatenate([inner, np.nextafter(inner, -np.inf), np.nextafter(inner, np.inf)]):
            v = base.copy()
            if which == 0:
                v[0], v[1] = x, 0.0        # R = |x| exactly
                v[2] = z_axis
            else:
                v[0], v[1] = r_axis, 0.0
                v[2] = x
            states.append(v)
    v = np.array(states)
    dev = hip.probe(q, v)
    n_inside = 0
    for i in range(len(v)):
        ora = oracle_lib.probe(q, v[i])
        np.testing.assert_array_equal(dev["codes"][i], ora["codes"], err_msg=f"codes of state {i}: {v[i][:3]}")
        if ora["codes"][0] != 0:
            continue   # the equilibrium refuses the point (outside the plasma / box): the fields are undefined in the reference
        n_inside += 1
        for key in ("cold", "dvds"):
            np.testing.assert_array_equal(dev[key][i], ora[key][:7] if key == "cold" else ora[key],
                                          err_msg=f"{key} of state {i}: {v[i][:3]}")
        np.testing.assert_array_equal(dev["resid"][i], ora["resid"])
    assert n_inside > 100   # most of these points lie inside the plasma: the lookups were really made
