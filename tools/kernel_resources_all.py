#!/usr/bin/env python3
"""Register / scratch / spill figures of the kernels the BASELINE configs dispatch, as hipcc reports them
(-Rpass-analysis=kernel-resource-usage) with the tracked Makefile's flags (NS = 2 instantiations: -DRAYS_INST_FAST).
    python tools/kernel_resources_all.py > profiles/rNN/kernel_resources.txt"""
import os
import re
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rays_amd", "csrc")
BASE = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
        "-DRAYS_INST_FAST", "-Rpass-analysis=kernel-resource-usage", "-c", "rays_inst.hip", "-o", "/dev/null"]
EXACT = ["-ffp-contract=off"]
TOL = ["-ffp-contract=fast-honor-pragmas", "-fassociative-math", "-fno-signed-zeros", "-fno-trapping-math", "-DRAYS_TOL_FLAVOUR", "-DRAYS_INST_TOL=1"]


def group(solver, eq, deriv, ue, tol=False):
    defs = [f"-DRAYS_INST_SOLVER={solver}", f"-DRAYS_INST_EQ={eq}", f"-DRAYS_INST_DERIV={deriv}", f"-DRAYS_INST_UE={ue}",
            "-DRAYS_INST_MS=0", f"-DRAYS_INST_EQT={eq + 4 * ue + (16 if tol else 0)}"]
    out = subprocess.run(BASE + (TOL if tol else EXACT) + defs, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"^void |\(.*$", "", name)}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


print("# Kernel resources as hipcc reports them (-Rpass-analysis=kernel-resource-usage), tracked Makefile flags, NS = 2.")
print(f"# {'kernel':44s} VGPRs AGPRs scratch B/lane  SGPR spills  VGPR spills  waves/SIMD  static LDS B")
for label, args in (("exact build (-ffp-contract=off)", [(0, 0, 0, 1), (0, 1, 0, 1), (0, 2, 0, 1), (1, 1, 1, 1), (1, 2, 0, 1)]),
                    ("tolerance flavour (" + " ".join(TOL[:4]) + ")", [(0, 0, 0, 1, True), (0, 1, 0, 1, True), (0, 2, 0, 1, True)])):
    print("# " + label)
    for a in args:
        for r in group(*a):
            print(f"{r['name']:46s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('ScratchSize [bytes/lane]', '?'):>14s} "
                  f"{r.get('SGPRs Spill', '?'):>12s} {r.get('VGPRs Spill', '?'):>12s} {r.get('Occupancy [waves/SIMD]', '?'):>11s} "
                  f"{r.get('LDS Size [bytes/block]', '?'):>13s}")
