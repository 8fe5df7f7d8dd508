"""The product kernel source under AddressSanitizer + UBSan (GPU ASan is not available on the pool):
tests/hip_emul compiles rays_rk4.hpp / rays_sg.hpp / rays_ray_init.hpp / rays_deposition.hpp for the
host; this runs every golden fixture through that build, incl. the SG kernel with shrunken storage
tiers.

    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
      ASAN_OPTIONS=detect_leaks=0 python tools/sanitize_emulation.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import emul_lib
emul_lib._LIB = '/tmp/librays_emul_asan.so'
emul_lib._TIERS_LIB = '/tmp/librays_emul_tiers_asan.so'
emul_lib.build(sanitize=True)
emul_lib.build(sanitize=True, out=emul_lib._TIERS_LIB, defs=emul_lib._TIERS_DEFS)
from tests.common import GOLDEN_CASES, load_golden, assert_matches_golden, padded_full_trajectories
from rays_amd.ray_init import fan_from_namelist
import rays_amd.hip as hip
for name in GOLDEN_CASES:
    g, nml, p = load_golden(name)
    out = emul_lib.trace(p, g["rvec0"], g["rindex_vec0"])
    assert_matches_golden(out, g, p, exact=True)
    if p.ode_solver == 1:
        out = emul_lib.trace(p, g["rvec0_full"], g["rindex_vec0_full"], small_tiers=True)
        assert np.array_equal(out["npoints"], g["npoints_full"])
    fan, nmax = fan_from_namelist(nml)
    r0, n0 = emul_lib.ray_init(p, fan, nmax)
    assert np.array_equal(r0, g["rvec0_full"])
    print(name, "ok", flush=True)
g, nml, p = load_golden("gold_axisym64_eqdsk_damp_rk4")
rv = padded_full_trajectories(g, p)
for w in (0, 1):
    work, prof = emul_lib.deposition(p, w, int(g["dep_n_bins"]), rv, g["npoints_full"], g["dep_power"], g["dep_rho_grid"], g["dep_rho_fspl"])
    assert np.array_equal(prof, g["dep_profile"][w])
print("sanitized emulation run: all ok")
