"""CPU tier: the PRODUCT kernel source (rays_amd/csrc/rays_rk4.hpp, rays_sg.hpp, rays_device.hpp)
compiled for the host with a one-lane HIP emulation (tests/hip_emul) and compared with the
reference golden vectors bit for bit.  This covers the integrator state machines, stop logic,
LDS staging/flush indexing and the summary fields without a GPU."""
import numpy as np
import pytest

from tests import emul_lib
from tests.common import GOLDEN_CASES, assert_matches_golden, load_golden


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_kernel_source_on_host_equals_reference(name):
    g, nml, p = load_golden(name)
    out = emul_lib.trace(p, g["rvec0"], g["rindex_vec0"])
    assert_matches_golden(out, g, p, exact=True)
