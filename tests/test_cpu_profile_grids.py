"""CPU tier: the profile splines n(psi), Te(psi), Ti(psi) of the axisym_toroid equilibrium on three grids, on two (Te and Ti
share one) and on one -- the kernel source looks tables that share a grid up with ONE cell search (rays_device_arith.inc:
spl1_ax3, DevParams::a_profiles_one_grid, decided on the host in rays_dev_params.inc: set_spline_axes).  Whatever is shared,
the trajectories are the oracle's, bit for bit.  (The reference fixtures have Te and Ti on one grid and n on another.)"""
import numpy as np
import pytest

from tests import emul_lib, oracle_lib
from tests.common import load_golden

ARRAYS = ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals")


def _resample(grid, fspl, new_grid):
    """The piecewise cubic (grid, fspl(4, n)) written on `new_grid` (each new knot's piece is the old piece it lies in,
    shifted to the knot): the same function up to rounding wherever the new knots refine the old ones."""
    X, c = np.asarray(grid, dtype=np.float64), np.asarray(fspl, dtype=np.float64).reshape(-1, 4)
    x = np.asarray(new_grid, dtype=np.float64)
    i = np.clip(np.searchsorted(X, x, side="right") - 1, 0, len(X) - 2)
    d = x - X[i]
    c0, c1, c2, c3 = c[i, 0], c[i, 1], c[i, 2], c[i, 3]
    out = np.stack([c0 + d * (c1 + d * (c2 + d * c3)), c1 + d * (2 * c2 + 3 * c3 * d), c2 + 3 * c3 * d, c3], axis=1)
    return np.ascontiguousarray(out.reshape(-1))


def _tables(g):
    return {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k].copy()) for k in g.files if k.startswith("axi_")}


@pytest.mark.parametrize("mode", ["three_grids", "te_ti_share", "one_grid"])
def test_profile_lookups_whatever_grids_they_share(mode):
    g, nml, p = load_golden("gold_axisym64_eqdsk129_tspline_damp_rk4")
    tab = _tables(g)
    fine = np.linspace(0.0, 1.0, 41)   # refines both fixture grids (11 and 9 knots on [0, 1])
    assert len(tab["ne_grid"]) == 11 and len(tab["te_grid"]) == 9 and np.array_equal(tab["te_grid"], tab["ti_grid"])
    grids = {"three_grids": (tab["ne_grid"], tab["te_grid"], np.linspace(0.0, 1.0, 17)),
             "te_ti_share": (tab["ne_grid"], tab["te_grid"], tab["ti_grid"]),
             "one_grid": (fine, fine, fine)}[mode]
    for name, grid in zip(("ne", "te", "ti"), grids):
        tab[name + "_fspl"] = _resample(tab[name + "_grid"], tab[name + "_fspl"], grid)
        tab[name + "_grid"] = np.ascontiguousarray(grid, dtype=np.float64)
    try:
        oracle_lib.set_axisym_tables(tab)
        emul_lib.set_axisym_tables(tab)
        r0, n0 = g["rvec0_full"][:24], g["rindex_vec0_full"][:24]
        ora = oracle_lib.trace(p, r0, n0)
        out = emul_lib.trace(p, r0, n0)
        assert ora["npoints"].max() > 5
        for k in ARRAYS:
            np.testing.assert_array_equal(out[k], ora[k], err_msg=k)
    finally:
        load_golden("gold_axisym64_eqdsk129_tspline_damp_rk4")   # the fixture's own tables back into both libraries
