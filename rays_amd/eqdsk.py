"""Host mirror of the reference's g-eqdsk reader and of initialize_eqdsk_magnetics_lin_interp: the tables of
magnetics_model = 'eqdsk_magnetics_lin_interp' need no spline fit, so the Python host can build them itself
(the Fortran host passes its own eqdsk_utilities_m arrays, fortran/rays_hip_state_m.f90).

  ReadgFile                                  eqdsk_utilities_m.f90:52-105
  initialize_eqdsk_magnetics_lin_interp      eqdsk_magnetics_lin_interp_m.f90:62-144
"""
from __future__ import annotations

from typing import Any, Dict

import numpy as np


def _numbers(lines, start, count):
    """`count` reals in format (5e16.9) starting at line `start`: 16-character fields, five to a line."""
    out = []
    i = start
    while len(out) < count:
        line = lines[i].rstrip("\n")
        for c in range(0, min(len(line), 80), 16):
            field = line[c:c + 16].strip()
            if field and len(out) < count:
                out.append(float(field.replace("D", "E").replace("d", "e")))
        i += 1
    return np.array(out, dtype=np.float64), i


def read_gfile(path: str) -> Dict[str, Any]:
    """ReadgFile (eqdsk_utilities_m.f90:52-105): header, T, P, TTp, Pp, Psi(NRBOX, NZBOX), Q, boundary, limiter."""
    with open(path, "r") as f:
        lines = f.readlines()
    head = lines[0].rstrip("\n")                       # (a48, 3i4)
    nr, nz = int(head[52:56]), int(head[56:60])
    g: Dict[str, Any] = dict(NRBOX=nr, NZBOX=nz)
    v, i = _numbers(lines, 1, 20)
    (g["RBOXLEN"], g["ZBOXLEN"], g["R0"], g["RBOXLFT"], g["ZOFF"],
     g["RAXIS"], g["ZAXIS"], g["PSIAXIS"], g["PSIBOUND"], g["B0"], g["CURRENT"]) = (float(x) for x in v[:11])
    for name in ("T", "P", "TTp", "Pp"):
        g[name], i = _numbers(lines, i, nr)
    psi, i = _numbers(lines, i, nr * nz)
    g["Psi"] = psi                                     # ((Psi(i, j), i = 1, NRBOX), j = 1, NZBOX): Fortran order
    g["Q"], i = _numbers(lines, i, nr)
    return g


def eqdsk_lin_tables(path: str) -> Dict[str, Any]:
    """Tables + box + psiB as initialize_eqdsk_magnetics_lin_interp leaves them (:110-141), keyed as
    rays_amd.hip.set_axisym_tables / params_from_namelist expect them."""
    g = read_gfile(path)
    nr, nz = g["NRBOX"], g["NZBOX"]
    box_rmin = g["RBOXLFT"]
    box_rmax = box_rmin + g["RBOXLEN"]
    box_zmin = g["ZOFF"] - g["ZBOXLEN"] / 2.0
    box_zmax = g["ZOFF"] + g["ZBOXLEN"] / 2.0
    r_grid = np.array([box_rmin + (box_rmax - box_rmin) * float(i) / float(nr - 1) for i in range(nr)])   # :128-130
    z_grid = np.array([box_zmin + (box_zmax - box_zmin) * float(i) / float(nz - 1) for i in range(nz)])   # :132-134
    return dict(box_rmin=box_rmin, box_rmax=box_rmax, box_zmin=box_zmin, box_zmax=box_zmax,
                psiB=g["PSIBOUND"] - g["PSIAXIS"],                                   # :140-141
                r_grid=r_grid, z_grid=z_grid,
                lin_dR=float((r_grid[1] - r_grid[0]) / 2.0), lin_dZ=float((z_grid[1] - z_grid[0]) / 2.0),   # :137-138
                lin_psi=g["Psi"] - g["PSIAXIS"],                                     # :139
                lin_t=g["T"])
