#!/bin/bash
# Developer A/B builds in a minute instead of a full build: compiles only the six RK4 objects the BASELINE configs
# dispatch (Solovev / eqdsk / slab, unit exponents, exact + tolerance; electrons + one ion) from the source tree SRC
# (a copy of rays_amd/csrc, possibly with edits or extra -D switches) and links them with the other objects of
# rays_amd/csrc/build_fast (make -C rays_amd/csrc FAST=1 first) into rays_amd/lib/librays_hip_exp_NAME.so.
# usage: tools/variant_rk4_objects.sh NAME SRC [-D...]
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/var_$NAME; mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DRAYS_INST_FAST $*"
TOL="-ffp-contract=fast -fassociative-math -fno-signed-zeros -fno-trapping-math -DRAYS_TOL_FLAVOUR -DRAYS_INST_TOL=1"
cd $SRC
for e in 0 1 2; do
  /opt/rocm/bin/hipcc $BASE -ffp-contract=off -DRAYS_INST_SOLVER=0 -DRAYS_INST_EQ=$e -DRAYS_INST_DERIV=0 -DRAYS_INST_UE=1 -DRAYS_INST_MS=0 -DRAYS_INST_EQT=$((e + 4)) -c rays_inst.hip -o $OUT/inst_0_${e}_0_1_0.o &
  /opt/rocm/bin/hipcc $BASE $TOL -DRAYS_INST_SOLVER=0 -DRAYS_INST_EQ=$e -DRAYS_INST_DERIV=0 -DRAYS_INST_UE=1 -DRAYS_INST_MS=0 -DRAYS_INST_EQT=$((e + 20)) -c rays_inst.hip -o $OUT/tol_${e}_1.o &
done
wait
OBJS=""
for o in $ROOT/rays_amd/csrc/build_fast/*.o; do
  b=$(basename $o)
  if [ -f $OUT/$b ]; then OBJS="$OBJS $OUT/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/rays_amd/lib/librays_hip_exp_$NAME.so $OBJS -lpthread -ldl
echo built $ROOT/rays_amd/lib/librays_hip_exp_$NAME.so
