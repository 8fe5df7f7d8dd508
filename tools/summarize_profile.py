#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/profile_bench.sh to pmc_summary.json (per-launch means of
the trace kernel's counters + the derived HBM traffic, corrected as MI355X_MICROARCH.md says)."""
import collections
import csv
import glob
import json
import os
import sys

out, args = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
kernel = None
bench = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
# the kernel of the bench line (the run also times the other numerics flavour of the same kernel: not this one's counters)
want = "rays::" + bench["config"]["kernel"]
is_it = lambda name: name.split("(")[0].replace("void ", "").strip() == want
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if is_it(r["Kernel_Name"]):
            kernel = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for k, v in sorted(agg.items())}
m = lambda k: c[k]["mean_per_launch"] if k in c else None
stats = {}
for r in csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))):
    if is_it(r["Name"]):
        stats = {"name": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                 "percentage": float(r["Percentage"])}
d = {}
if m("WRITE_SIZE") is not None:
    d["hbm_write_bytes_per_launch"] = m("WRITE_SIZE") * 1024.0          # KiB
if m("FETCH_SIZE") is not None:
    d["hbm_read_bytes_per_launch_x2_gfx950_correction"] = 2.0 * m("FETCH_SIZE") * 1024.0
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
d["algorithmic_bytes_per_launch"] = alg
if "hbm_write_bytes_per_launch" in d:
    d["write_amplification"] = d["hbm_write_bytes_per_launch"] / alg
    d["hbm_bytes_per_launch"] = d["hbm_write_bytes_per_launch"] + d.get("hbm_read_bytes_per_launch_x2_gfx950_correction", 0.0)
if m("SQ_INSTS_VALU") and m("SQ_WAVES"):
    d["valu_insts_per_wave"] = m("SQ_INSTS_VALU") / m("SQ_WAVES")
    f64 = sum(m(k) or 0.0 for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
    if f64:
        d["f64_arith_fraction_of_valu"] = f64 / m("SQ_INSTS_VALU")
if m("SQ_WAVE_CYCLES") and m("SQ_WAIT_ANY"):
    # SQ_WAIT_ANY: wave-cycles waiting for anything (memory counters, but also issue and dependencies): an upper bound on
    # memory waits -- the headline kernel, whose trip holds no s_waitcnt, shows 0.20 (profiles/README.md)
    d["wave_time_parked_in_waitcnt"] = m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES")
if m("GRBM_GUI_ACTIVE") and stats:
    d["effective_clock_GHz"] = m("GRBM_GUI_ACTIVE") / 8.0 / stats["avg_ns"]  # summed over 8 XCDs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402
summary = {"command": f"tools/profile_bench.sh ... {args}".strip(), "kernel": kernel, "kernel_stats": stats,
           "source_hash": kernel_source_hash(),
           "bench_line": bench, "counters": c, "derived": d,
           "note": "WRITE_SIZE/FETCH_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half)"}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
# the entry bench.py reads from profiles/counters.json (merged there by tools/update_counters.py)
entry = {"kernel": bench["config"]["kernel"], "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"),
         "fp64_wave_insts": {k: m(f"SQ_INSTS_VALU_{k.upper()}_F64") for k in ("add", "mul", "fma", "trans")
                             if m(f"SQ_INSTS_VALU_{k.upper()}_F64") is not None},
         "valu_insts": m("SQ_INSTS_VALU"), "effective_clock_GHz": d.get("effective_clock_GHz"),
         "wave_time_parked_in_waitcnt": d.get("wave_time_parked_in_waitcnt"),
         "kernel_avg_ms_rocprof": stats.get("avg_ns", 0.0) / 1e6 if stats else None}
json.dump({bench["config"]["workload"]: entry}, open(os.path.join(out, "counters_entry.json"), "w"), indent=1)
print(json.dumps({"kernel": kernel, "kernel_stats": stats, "derived": d}, indent=1))
