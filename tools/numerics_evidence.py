#!/usr/bin/env python3
"""The full-fan numerics survey of tests/numerics_survey.py for one configuration and flavour, as a tool (A/B of
libraries: RAYS_HIP_LIB).   python tools/numerics_evidence.py configs/cfg3b_solovev64k_rk4.in [tolerance|exact] [ray stride]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rays_amd import hip  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.numerics_survey import survey  # noqa: E402

cfg = sys.argv[1]
flavour = sys.argv[2] if len(sys.argv) > 2 else "tolerance"
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hip.set_numerics(flavour)
nml, p, r0, n0 = bench.build_fan(cfg, 1, 1, None)
if bench.build_fan.tables is not None:
    oracle_lib.set_axisym_tables(bench.build_fan.tables)
batch = 65536 if "w2" not in hip.kernel_name(p, len(r0)) else 262144
res = survey(p, r0, n0, ray_stride=stride, restart_batch=batch)
res.pop("worst_steps_states", None)
print(json.dumps(res))
