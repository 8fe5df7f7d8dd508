"""Drop-in test: the REFERENCE host program (initialize / ray_init / ray_results_m, compiled from
the reference sources) with `trace_rays` replaced by fortran/trace_rays_hip.f90, the three ray
launchers by fortran/{solovev,simple_slab,axisym_toroid}_ray_init_hip.f90 and the deposition binning by
fortran/deposition_profiles_hip.f90 -> C ABI -> HIP,
against the unmodified reference binary on the same namelist.  Both binaries are built by
oracle/build_ref.sh where /root/reference is available and travel to the GPU box prebuilt."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from tests.common import ROOT
from tests.refdump import read_dump

pytestmark = pytest.mark.gpu

REF = os.path.join(ROOT, "oracle", "_ref", "rays_ref_dump")
HIPBIN = os.path.join(ROOT, "oracle", "_ref", "rays_hip_dropin")


def _run(binary, cfg, d, extra_env=None):
    os.makedirs(d, exist_ok=True)
    shutil.copy(os.path.join(ROOT, "configs", cfg), os.path.join(d, "rays.in"))
    for f in os.listdir(os.path.join(ROOT, "configs")):
        if f.endswith(".geqdsk") or f.startswith("ray_init_"):   # (file_input_ray_init reads ray_init_<run_label>.in)
            shutil.copy(os.path.join(ROOT, "configs", f), d)
    env = dict(os.environ, RAYS_DUMP_FILE="dump.bin", RAYS_DUMP_PROBE="0", **(extra_env or {}))
    r = subprocess.run([binary], cwd=d, env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    _run.last_stderr = r.stderr
    return read_dump(os.path.join(d, "dump.bin"))


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIPBIN)),
                    reason="reference binaries not built (oracle/build_ref.sh needs /root/reference)")
@pytest.mark.parametrize("cfg", ["cfg1_slab16_rk4.in", "cfg2_solovev1024_rk4.in", "gold_slab_ns1_rk4.in",
                                 "gold_axisym64_eqdsk_damp_sg.in", "gold_slab16_damp_rk4.in",
                                 "gold_solovev64_sg_cold.in", "gold_solovev64_rk4_num.in",
                                 "gold_solovev64_damp_rk4.in", "gold_axisym64_eqdsk_damp_rk4.in",
                                 "gold_slab_lin2_rk4_num.in", "gold_slab_negative_dens_rk4.in",
                                 "gold_solovev64_slow_sg.in", "gold_axisym64_solmag_damp_rk4.in",
                                 "gold_axisym64_solmag_splines_grad_rk4.in", "gold_axisym64_eqlin_damp_rk4.in",
                                 "gold_slab_one_ray_rk4.in", "gold_solovev_file_rays_damp_rk4.in",
                                 "gold_axisym64_eqdsk129_tspline_damp_rk4.in", "gold_axisym64_eqdsk129_tspline_damp_sg.in"])
def test_fortran_dropin_equals_reference_binary(cfg):
    with tempfile.TemporaryDirectory() as d:
        ref = _run(REF, cfg, os.path.join(d, "ref"))
        hipr = _run(HIPBIN, cfg, os.path.join(d, "hip"))
    assert hipr["nray"] == ref["nray"]
    np.testing.assert_array_equal(hipr["rvec0"], ref["rvec0"])
    np.testing.assert_array_equal(hipr["rindex_vec0"], ref["rindex_vec0"])  # every launcher runs on the GPU
    np.testing.assert_array_equal(hipr["npoints"], ref["npoints"])
    assert hipr["stop_flag"] == ref["stop_flag"]          # the exact strings, leading blank included
    np.testing.assert_array_equal(hipr["ray_vec"], ref["ray_vec"])  # bit-identical trajectories, every row
    np.testing.assert_array_equal(hipr["residual"], ref["residual"])
    np.testing.assert_array_equal(hipr["end_ray_vec"], ref["end_ray_vec"])


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIPBIN)),
                    reason="reference binaries not built (oracle/build_ref.sh needs /root/reference)")
@pytest.mark.parametrize("cfg", ["gold_axisym64_eqdsk_damp_rk4.in", "gold_slab16_damp_rk4.in",
                                 "gold_axisym64_solmag_damp_rk4.in", "gold_axisym64_eqlin_damp_rk4.in",
                                 "gold_axisym64_eqdsk129_tspline_damp_rk4.in"])
def test_fortran_deposition_dropin_equals_reference_post_processor(cfg):
    """fortran/deposition_profiles_hip.f90 (rays_hip_deposition on the ray_results_m arrays of the drop-in run)
    against the reference's calculate_deposition_profiles on the reference run: work(n_bins, nray), the profiles
    and Q_sum of 'Ptotal_psi', 'Ptotal_rho' (eqdsk) / 'Ptotal_x' (slab), bit for bit."""
    from tests.refdump import read_deposition
    outs, files = {}, {}
    for tag, binary in (("ref", REF), ("hip", HIPBIN)):
        with tempfile.TemporaryDirectory() as d:
            _run(binary, cfg, d, extra_env={"RAYS_DUMP_DEPOSITION": "dep.bin", "RAYS_DUMP_DEPOSITION_LD": "1",
                                            "RAYS_HIP_TIMING": "1"})
            if tag == "hip":
                # trace -> profiles in one process (RAYS_P.f90:19-44): binned from the image rays_hip_trace left on the
                # device (rays_hip_deposition_last), the trajectories are never uploaded again
                assert "[rays_hip_deposition_last]" in _run.last_stderr and "no trajectory upload" in _run.last_stderr, \
                    _run.last_stderr[-800:]
            outs[tag] = read_deposition(os.path.join(d, "dep.bin"))
            # the profile file post_process_RAYS / graphics_RAYS read, written by the REFERENCE's own
            # write_deposition_profiles_LD (deposition_profiles_m.f90:296-331) from the reference's sums / the GPU's
            f = [x for x in os.listdir(d) if x.startswith("deposition_profiles.")]
            files[tag] = open(os.path.join(d, f[0])).read() if f else None
    assert files["hip"] == files["ref"]          # byte for byte (None for both where only one profile exists)
    assert outs["hip"]["names"] == outs["ref"]["names"] and len(outs["ref"]["names"]) >= 1
    for k in ("work", "profile"):
        for a, b in zip(outs["hip"][k], outs["ref"][k]):
            np.testing.assert_array_equal(a, b, err_msg=k)
    assert outs["hip"]["q_sum"] == outs["ref"]["q_sum"]
