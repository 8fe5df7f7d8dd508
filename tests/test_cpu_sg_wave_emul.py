"""CPU tier: the one-ray-per-lane Shampine-Gordon kernel (rays_sg.hpp: sg_trace_kernel) on whole emulated waves
(tests/hip_emul/hip/hip_wave_emul.h).  A single emulated lane votes alone; here 64 lanes vote on the phase and the
interval each trip serves (some sit trips out), rays end on different trips, and lanes whose ray has ended pull the
next one while their neighbours are in the middle of a step.  None of that may change a bit of any ray."""
import numpy as np

from rays_amd.params import copy_params
from tests import group_emul_lib as ge
from tests import oracle_lib
from tests.common import load_golden
from tests.test_cpu_rk4_wave_emul import ARRAYS, _tables


def _check(out, ora):
    for k in ARRAYS:
        np.testing.assert_array_equal(out[k], ora[k], err_msg=k)


def test_solovev_fan_with_refills():
    """160 rays of the Solovev fan on ONE wave: every lane is refilled once or twice, the last lanes find the counter dry."""
    g, nml, p = load_golden("gold_solovev64_sg_cold")
    reps = -(-160 // len(g["rvec0_full"]))
    r0 = np.tile(g["rvec0_full"], (reps, 1))[:160].copy()
    n0 = np.tile(g["rindex_vec0_full"], (reps, 1))[:160].copy()
    n0[5] *= 3.0   # far off the dispersion surface: stops at its initial check
    q = copy_params(p)
    q.nstep_max = min(q.nstep_max, 40)
    ora = oracle_lib.trace(q, r0, n0)
    assert ora["npoints"][5] == 1
    _check(ge.trace_sg_waves(q, r0, n0, nwaves=1), ora)


def test_eqdsk_damping_fan_two_waves():
    """The kernel of BASELINE config 5 (eqdsk magnetics, damping: nv = 8): 300 short rays on two waves."""
    g, nml, p = load_golden("gold_axisym64_eqdsk129_tspline_damp_sg")
    library = ge.lib()
    _tables(g, library)
    reps = -(-300 // len(g["rvec0_full"]))
    r0, n0 = np.tile(g["rvec0_full"], (reps, 1))[:300], np.tile(g["rindex_vec0_full"], (reps, 1))[:300]
    q = copy_params(p)
    q.nstep_max = min(q.nstep_max, 12)
    ora = oracle_lib.trace(q, r0, n0)
    _check(ge.trace_sg_waves(q, r0, n0, nwaves=2, library=library), ora)
