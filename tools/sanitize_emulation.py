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
        tab = {k[4:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("axi_")}
        if any(np.size(tab.get(k, ())) for k in ("r_grid", "ne_grid", "te_grid", "ti_grid")):
            emul_lib.set_axisym_tables(tab, small_tiers=True)   # (the shrunken-tier build keeps its own copy)
        out = emul_lib.trace(p, g["rvec0_full"], g["rindex_vec0_full"], small_tiers=True)
        assert np.array_equal(out["npoints"], g["npoints_full"])
    if str(nml.get("ray_init_list", {}).get("ray_init_model", "")).strip() not in ("one_ray_init_XYZ_n_direction", "file_input_ray_init"):
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
# the bilinear eqdsk model at the corners and edges of its grid (GetPsi's unbounded cell index, rays_device.hpp)
g, nml, p = load_golden("gold_axisym64_eqlin_damp_rk4")
a = p.axisym
pts = [(r, z) for r in (a.box_rmin + 1e-4, 0.5 * (a.box_rmin + a.box_rmax), a.box_rmax - 1e-4, a.box_rmax, a.box_rmin)
       for z in (a.box_zmin + 1e-4, 0.0, a.box_zmax - 1e-4, a.box_zmax, a.box_zmin)]
r0 = np.array([[r, 0.0, z] for r, z in pts])
out = emul_lib.trace(p, r0, np.tile(g["rindex_vec0"][0], (len(pts), 1)))
print("eqdsk_magnetics_lin_interp edge cells: ok", out["stop_code"].tolist())
