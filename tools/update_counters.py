#!/usr/bin/env python3
"""Merge per-configuration counter entries (pmc_summary.json files written by tools/summarize_profile.py and
copied under profiles/rNN/) into profiles/counters.json, which bench.py reads for roofline.traffic and
roofline_fp64.   python tools/update_counters.py profiles/r02/*_pmc_summary.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402
path = os.path.join(ROOT, "profiles", "counters.json")
db = json.load(open(path)) if os.path.exists(path) else {}
for f in sys.argv[1:]:
    if "_index_order" in os.path.basename(f):  # comparison profiles (RAYS_HIP_RAY_ORDER=index): not what bench.py runs
        continue
    s = json.load(open(f))
    c, d, b = s["counters"], s["derived"], s["bench_line"]
    m = lambda k: c[k]["mean_per_launch"] if k in c else None
    # keyed by workload AND kernel (a workload has an exact and a tolerance flavour); source_hash = the kernel sources
    # the counters were collected from (written by summarize_profile.py on the GPU box, i.e. from the snapshot that ran)
    db[b["config"]["workload"] + "::" + b["config"]["kernel"]] = {
        "kernel": b["config"]["kernel"], "source_hash": s.get("source_hash") or kernel_source_hash(), "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"),
        "fp64_wave_insts": {k: m(f"SQ_INSTS_VALU_{k.upper()}_F64") for k in ("add", "mul", "fma", "trans")
                            if m(f"SQ_INSTS_VALU_{k.upper()}_F64") is not None},
        "valu_insts": m("SQ_INSTS_VALU"), "effective_clock_GHz": d.get("effective_clock_GHz"),
        "wave_time_parked_in_waitcnt": d.get("wave_time_parked_in_waitcnt"),
        "kernel_avg_ms_rocprof": s.get("kernel_stats", {}).get("avg_ns", 0.0) / 1e6,
        "source": os.path.relpath(os.path.abspath(f), ROOT) + " (rocprofv3 --pmc, separate passes; FETCH_SIZE x2 on gfx950)"}
json.dump(db, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(db, indent=1))
