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
    """configs/solovev_65x65.geqdsk.tables.npz carries every profile spline an eqdsk namelist of this repo selects
    (ne_* for cfg 5 / 5b, te_* / ti_* for the splined-temperature cases): no case's tables overwrite another's."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "configs", "solovev_65x65.geqdsk.tables.npz"))
    for k in ("r_grid", "z_grid", "psi_fspl", "rb_grid", "rb_fspl", "ne_grid", "ne_fspl", "te_grid", "te_fspl",
              "ti_grid", "ti_fspl", "rho_grid", "rho_fspl"):
        assert k in z.files and z[k].size, k
