#!/usr/bin/env bash
# TEST / BENCH DATA: BASELINE config 5's equilibrium file as SURVEY 8(d) wrote it -- "eqdsk 129 x 129 written by the
# reference's own solovev_2_eqdsk".  Runs oracle/_ref/solovev_2_eqdsk (RAYS_project/solovev_2_eqdsk/solovev_2_eqdsk.f90
# compiled from /root/reference by oracle/build_ref.sh) on configs/solovev_2_eqdsk_129.in -- the Solovev equilibrium of
# every other config of this repo (rmaj 1, kappa 1.1, B0 3.3 T, outer boundary 1.4, box 0.5..1.5 x -0.7..0.7) -- and
# keeps its output, configs/solovev_129x129.geqdsk (a data file: psi(R, Z) on the grid, R B_phi, the boundary).
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/.." && pwd)
BIN=$HERE/_ref/solovev_2_eqdsk
[ -x "$BIN" ] || { echo "make_cfg5_eqdsk: $BIN missing -- run oracle/build_ref.sh where /root/reference is mounted"; exit 1; }
W=$(mktemp -d); trap 'rm -rf "$W"' EXIT
cp "$ROOT/configs/solovev_2_eqdsk_129.in" "$W/rays.in"
( cd "$W" && "$BIN" > solovev_2_eqdsk.log 2>&1 )
cp "$W/solovev_129x129.geqdsk" "$ROOT/configs/solovev_129x129.geqdsk"
echo "configs/solovev_129x129.geqdsk: $(head -c 60 "$ROOT/configs/solovev_129x129.geqdsk" | head -1)"
