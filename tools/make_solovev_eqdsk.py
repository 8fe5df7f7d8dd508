#!/usr/bin/env python3
"""Write a g-eqdsk file for the analytic Solovev equilibrium of solovev_eq_m (the reference ships
no eqdsk; its own solovev_2_eqdsk tool does the same thing).  Pure data: psi(R,Z) on a uniform
grid, F = R*Bphi, a flat q profile, a 5-point boundary.  Format = the reads of
RAYS_project/RAYS_lib/eqdsk_utilities_m.f90:52-105 (a48,3i4 / 5e16.9 / 2i5).

    python tools/make_solovev_eqdsk.py configs/solovev_65x65.geqdsk
"""
import sys
import numpy as np


def main(path, nr=65, nz=65):
    rmaj, kappa, bphi0, iota0, outer_bound = 1.0, 1.1, 3.3, 1.0e-5, 1.4
    rmin, rmax, zmin, zmax = 0.5, 1.5, -0.7, 0.7
    bp0 = bphi0 * iota0
    psiB = 0.5 * bp0 * (outer_bound**2 - rmaj**2) ** 2 / rmaj**2 / 4.0
    R = rmin + (rmax - rmin) * np.arange(nr) / (nr - 1)
    Z = zmin + (zmax - zmin) * np.arange(nz) / (nz - 1)
    RR, ZZ = np.meshgrid(R, Z, indexing="ij")
    psi = 0.5 * bp0 * ((RR * ZZ / (rmaj * kappa)) ** 2 + ((RR**2 - rmaj**2) ** 2) / rmaj**2 / 4.0)
    inner = np.sqrt(2.0 * rmaj**2 - outer_bound**2)
    r_zmax = (2.0 * outer_bound**2 * rmaj**2 - outer_bound**4) ** 0.25
    vert = kappa / (2.0 * r_zmax) * np.sqrt(outer_bound**4 + 2.0 * (r_zmax**2 - outer_bound**2) * rmaj**2 - r_zmax**4)
    rb = [inner, r_zmax, outer_bound, r_zmax, inner]
    zb = [0.0, vert, 0.0, -vert, 0.0]

    def rows(vals):
        vals = list(vals)
        return "".join("".join("%16.9e" % v for v in vals[i:i + 5]) + "\n" for i in range(0, len(vals), 5))

    with open(path, "w") as f:
        f.write("%-48s%4d%4d%4d\n" % ("  SOLOVEV rays-mi355x test equilibrium", 3, nr, nz))
        f.write(rows([rmax - rmin, zmax - zmin, rmaj, rmin, 0.0]))
        f.write(rows([rmaj, 0.0, 0.0, psiB, bphi0]))
        f.write(rows([0.0, 0.0, 0.0, rmaj, 0.0]))
        f.write(rows([0.0, 0.0, psiB, 0.0, 0.0]))
        f.write(rows([bphi0 * rmaj] * nr))          # T = R*Bphi
        f.write(rows([0.0] * nr))                   # P
        f.write(rows([0.0] * nr))                   # TTp
        f.write(rows([0.0] * nr))                   # Pp
        f.write(rows(psi.T.reshape(-1)))            # ((Psi(i,j), i=1,NR), j=1,NZ)
        f.write(rows(1.0 + 2.0 * np.arange(nr) / (nr - 1)))  # q: 1 -> 3
        f.write("%5d%5d\n" % (len(rb), 1))
        f.write(rows([v for pair in zip(rb, zb) for v in pair]))
        f.write(rows([0.0, 0.0]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "solovev.geqdsk")
