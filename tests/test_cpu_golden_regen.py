"""The committed fixtures are what the committed generator produces: `make_golden.py --check` regenerates every
fixture from the reference binary into a scratch directory and compares it with tests/golden/ and the host
tables under configs/ bit for bit (the results file: up to its date / wall-time records).  Runs where the
reference binary exists (this container; it is built from /root/reference by oracle/build_ref.sh)."""
import os
import sys

import pytest

from tests.common import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "rays_ref_dump")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/rays_ref_dump not built (no /root/reference here)")
def test_committed_fixtures_are_what_the_generator_produces():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    try:
        import make_golden
    finally:
        sys.path.pop(0)
    diffs = make_golden.check()
    assert not diffs, "\n".join(diffs)


def test_host_tables_are_the_union_of_the_eqdsk_cases():
    """configs/<eqdsk>.tables.npz carries every profile spline an eqdsk namelist of this repo selects on that file
    (65 x 65: ne_* for the damping fixtures, te_* / ti_* for the splined-temperature ones; 129 x 129, BASELINE config
    5's file: all three): no case's tables overwrite another's."""
    import numpy as np
    for f, nr in (("solovev_65x65.geqdsk", 65), ("solovev_129x129.geqdsk", 129)):
        z = np.load(os.path.join(ROOT, "configs", f + ".tables.npz"))
        for k in ("r_grid", "z_grid", "psi_fspl", "rb_grid", "rb_fspl", "ne_grid", "ne_fspl", "te_grid", "te_fspl",
                  "ti_grid", "ti_fspl", "rho_grid", "rho_fspl"):
            assert k in z.files and z[k].size, (f, k)
        assert len(z["r_grid"]) == nr and len(z["z_grid"]) == nr


def test_cfg5_equilibrium_file_is_what_the_references_own_tool_writes(tmp_path):
    """BASELINE config 5: "eqdsk 129 x 129 written by solovev_2_eqdsk".  configs/solovev_129x129.geqdsk
    must be the output of oracle/_ref/solovev_2_eqdsk (the reference's program, compiled by oracle/build_ref.sh) on
    configs/solovev_2_eqdsk_129.in, byte for byte; cfg 5 and cfg 5b must name it and the splined temperature model."""
    import shutil
    import subprocess
    from rays_amd.namelist import read_namelist
    for cfg in ("cfg5_axisym256k_sg_damp.in", "cfg5b_axisym256k_rk4_damp.in"):
        nml = read_namelist(os.path.join(ROOT, "configs", cfg))
        assert nml["eqdsk_magnetics_spline_interp_list"]["eqdsk_file_name"].strip() == "solovev_129x129.geqdsk"
        assert str(nml["axisym_toroid_eq_list"]["temperature_prof_model"][0]).strip() == "temperature_spline_interp" \
            if isinstance(nml["axisym_toroid_eq_list"]["temperature_prof_model"], (list, tuple, dict)) else True
    tool = os.path.join(ROOT, "oracle", "_ref", "solovev_2_eqdsk")
    if not os.path.exists(tool):
        pytest.skip("oracle/_ref/solovev_2_eqdsk not built (needs /root/reference)")
    shutil.copy(os.path.join(ROOT, "configs", "solovev_2_eqdsk_129.in"), tmp_path / "rays.in")
    subprocess.run([tool], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert (tmp_path / "solovev_129x129.geqdsk").read_bytes() == open(os.path.join(ROOT, "configs", "solovev_129x129.geqdsk"), "rb").read()
