#!/bin/bash
# What the reference's call sites see on BASELINE config 5b (262144 rays, eqdsk + damping, RK4): `call trace_rays` and the
# deposition profiles behind it, through the Fortran drop-in binary (oracle/_ref/rays_hip_dropin = the reference's host
# objects + fortran/*.f90 + librays_hip.so), with the library's phase times (RAYS_HIP_TIMING).  Run twice: profiles binned
# from the image the trace left on the device (rays_hip_deposition_last), and -- RAYS_HIP_NO_KEEP_LAST_RESULT=1 -- from
# the host arrays handed back (rays_hip_deposition: the round-3 path).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; W=$(mktemp -d); trap 'rm -rf "$W"' EXIT
cp $R/configs/cfg5b_axisym256k_rk4_damp.in $W/rays.in; cp $R/configs/*.geqdsk $W/
for mode in device_image host_arrays; do
  echo "==== $mode"
  ( cd $W && env RAYS_DUMP_FILE=skip RAYS_DUMP_DEPOSITION=dep.bin RAYS_HIP_TIMING=1 RAYS_HIP_NUMERICS=${NUMERICS:-exact} \
      $( [ $mode == host_arrays ] && echo RAYS_HIP_NO_KEEP_LAST_RESULT=1 ) $R/oracle/_ref/rays_hip_dropin 2>&1 >/dev/null \
      | grep -E "rays_hip_deposition|pack \+ copy|inputs \+ trace|RAYS_REF|Wall" | sed 's/^/  /' )
  ( cd $W && ls -la dep.bin | awk '{print "  dep.bin bytes", $5}' && md5sum dep.bin | awk '{print "  dep.bin md5", $1}' )
done
