"""CPU tier: the lane-group Shampine-Gordon kernel (rays_amd/csrc/rays_sg_group.hpp: one ray per group of G lanes,
deriv_num's central differences and the ODE components spread over the group, __shfl between its lanes) on the host
WAVE emulator (tests/hip_emul/hip/hip_wave_emul.h: 64 lanes per wave as fibers, cross-lane operations are
rendezvous), against the oracle and the reference fixtures, bit for bit."""
import numpy as np
import pytest

from rays_amd.params import copy_params
from tests import group_emul_lib as ge
from tests import oracle_lib
from tests.common import load_golden

ARRAYS = ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals")
# SG + finite-difference dD.  The two axisym fixtures carry damping (nv = 8: the one-ray-per-lane kernel's shape); their
# configurations are run here with damping switched off (nv = 7), against the oracle.
CASES = ["gold_solovev64_sg_num", "gold_slab_shear_gauss_3spec_sg_num", "gold_axisym64_solmag_sg_num",
         "gold_axisym64_eqlin_tspline_sg_num"]


def _tables(g):
    tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
    if any(np.size(tab.get(k, ())) for k in ("r_grid", "ne_grid", "te_grid", "ti_grid")):
        ge.set_axisym_tables(tab)


@pytest.mark.parametrize("G", [8, 4, 16])
@pytest.mark.parametrize("name", CASES)
def test_group_kernel_equals_oracle(name, G):
    """19 rays of every finite-difference SG fixture for six output intervals, more rays than the launched groups
    hold (two blocks of 256 / G groups at G = 16; at G = 4, 8 one partly filled block), incl. a ray that never
    starts and one that stops at its initial check."""
    g, nml, p = load_golden(name)
    _tables(g)
    q = copy_params(p)
    q.nstep_max = 6
    if q.damping_model:
        q.damping_model, q.nv = 0, 7
    n = min(19, len(g["rvec0_full"]))
    r0, n0 = g["rvec0_full"][:n].copy(), g["rindex_vec0_full"][:n].copy()
    if n > 4:
        r0[2, 0] = 10.0     # outside the box
        n0[3] *= 3.0        # far off the dispersion surface
    ora = oracle_lib.trace(q, r0, n0)
    out = ge.trace(q, r0, n0, G=G, resident_blocks=1)
    for k in ARRAYS:
        np.testing.assert_array_equal(out[k], ora[k], err_msg=k)


def test_group_kernel_refills_groups():
    """More rays than resident groups: 80 rays on ONE block of 32 groups (G = 8); finished groups pull the rest."""
    g, nml, p = load_golden("gold_solovev64_sg_num")
    q = copy_params(p)
    q.nstep_max = 3
    r0 = np.tile(g["rvec0_full"], (2, 1))[:80]
    n0 = np.tile(g["rindex_vec0_full"], (2, 1))[:80]
    ora = oracle_lib.trace(q, r0, n0)
    out = ge.trace(q, r0, n0, G=8, resident_blocks=1)
    for k in ARRAYS:
        np.testing.assert_array_equal(out[k], ora[k], err_msg=k)


def test_group_kernel_whole_ray_of_the_reference_fixture():
    """One ray of the reference fixture from launch to its end ('equations stiff' after 178 intervals)."""
    g, nml, p = load_golden("gold_solovev64_sg_num")
    out = ge.trace(p, g["rvec0"][:1], g["rindex_vec0"][:1], G=8)
    n = int(g["npoints"][0])
    assert out["npoints"][0] == n
    np.testing.assert_array_equal(out["ray_vec"][0, :n], g["ray_vec"][0, :n])
    np.testing.assert_array_equal(out["residual"][0, :n], g["residual"][0, :n])


@pytest.mark.parametrize("G", [8, 4])
def test_group_kernel_workspace_rows(G):
    """phi rows above the register tier live in the launch's workspace (rays_sg_group.hpp: GPhi).  Built with two
    register rows instead of six, every step of an ordinary ray crosses the tier boundary: same results."""
    g, nml, p = load_golden("gold_solovev64_sg_num")
    q = copy_params(p)
    q.nstep_max = 8
    r0, n0 = g["rvec0_full"][:11], g["rindex_vec0_full"][:11]
    ora = oracle_lib.trace(q, r0, n0)
    out = ge.trace(q, r0, n0, G=G, resident_blocks=1, small_tier=True)
    for k in ARRAYS:
        np.testing.assert_array_equal(out[k], ora[k], err_msg=k)
