#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *reference* RAYS hot path (RAYS_project) from the
# sources where they lie under /root/reference into binaries under oracle/_ref/.
#
#   oracle/_ref/rays_ref_dump   reference initialize + trace_rays + raw-binary dump (CPU)
#   oracle/_ref/solovev_2_eqdsk the reference's own tool that writes the g-eqdsk of a Solovev equilibrium
#                               (RAYS_project/solovev_2_eqdsk/solovev_2_eqdsk.f90): BASELINE config 5's 129 x 129 file
#                               is its output (oracle/make_cfg5_eqdsk.sh)
#   oracle/_ref/rays_hip_dropin reference host (initialize/ray_results/...) with trace_rays
#                               REPLACED by fortran/trace_rays_hip.f90, the three ray launchers by
#                               fortran/{solovev,simple_slab,axisym_toroid}_ray_init_hip.f90 and the
#                               deposition binning by fortran/deposition_profiles_hip.f90 -> C-ABI
#
# Nothing from /root/reference is kept: sources are streamed through `sed` into a scratch
# directory that is deleted before the script exits (objects, .mod files too); only the two
# linked executables stay, and oracle/_ref/ is git-ignored.
#
# The reference builds with gfortran + NetCDF + cmake; this image has amdflang and no NetCDF,
# so the recipe applies the *non-arithmetic* edits SURVEY.md §8(c) lists (P1..P7):
#   P1 type(eq_point(nspec=nspec)) -> type(eq_point)   (type is not parameterised)
#   P2 drop `use ode_m, only : ode_stop` inside submodules of ode_m (flang rejects self-use)
#   P3 `module subroutine` -> `subroutine` in quick_cube_splines_m
#   P4 drop `use axisym_toroid_ray_init_nphi_ntheta_m` (module absent from the tree)
#   P5 drop the re-declaration of zfun0/zfun0_real_arg in damp_fund_ECH
#   P6 cut the NetCDF writer/reader (I/O only) from ray_results_m / finalize_run
#   P7 cut the 'multiple_mirror' equilibrium case (drags in NetCDF)
#   P8 (solovev_2_eqdsk only) ask get_unit_number() for the unit BEFORE opening rays.in with it (the program opens
#      an undefined unit number first and asks afterwards)
# No statement on the hot path (trace_rays, ode, eqn_ray, equilibrium, deriv_*, check_save)
# is touched.  No stand-in for NetCDF is written: the I/O code that needs it is removed.
set -euo pipefail

REF=${RAYS_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/.." && pwd)
OUT=$HERE/_ref
FC=${FC:-/opt/rocm/bin/amdflang}
FFLAGS=${FFLAGS:-"-O2 -fopenmp -ffp-contract=off -w"}

if [ ! -d "$REF/RAYS_project/RAYS_lib" ]; then
  echo "build_ref: $REF not present -- keeping prebuilt oracle/_ref (if any)"; exit 0
fi
if [ ! -x "$FC" ]; then echo "build_ref: $FC missing"; exit 0; fi

W=$OUT/.build
rm -rf "$W"; mkdir -p "$W" "$OUT"
trap 'rm -rf "$W"' EXIT

P=$REF/RAYS_project
L=$P/RAYS_lib
p1() { sed -e 's/type(eq_point(nspec=nspec))/type(eq_point)/g' "$1"; }

# ---- splines_lib + math_functions_lib pieces the active library needs -------------------
for f in bcspline ibc_ck v_spline bcspeval cspline cspeval splinck zonfind; do
  cp "$P/splines_lib/$f.f90" "$W/"; done
sed -e 's/^\( *\)module subroutine/\1subroutine/' "$P/splines_lib/quick_cube_splines_m.f90" \
  > "$W/quick_cube_splines_m.f90"                                                    # P3
for f in zfunctions_m quad_trapezoid_m bin_to_uniform_grid_m bisect_m \
         monotonic_function_inversion complete_elliptic_int_m vectors3_m; do
  cp "$P/math_functions_lib/$f.f90" "$W/"; done

# ---- RAYS_lib ------------------------------------------------------------------------------
for src in "$L"/*.f90; do
  b=$(basename "$src")
  case "$b" in
    XY_curves_netCDF_m.f90|multiple_mirror_eq_m.f90|mirror_magnetics_spline_interp_m.f90) continue;;
    ray_tracing.f90) p1 "$src" > "$W/ray_tracing_ref.f90"; continue;;   # kept apart: replaced in dropin
  esac
  p1 "$src" > "$W/$b"                                                                 # P1
done
sed -i '/^ *use ode_m, only : ode_stop/d' "$W/RK4_ode_m.f90" "$W/SG_ode_m.f90"        # P2
sed -i '/use axisym_toroid_ray_init_nphi_ntheta_m/d' "$W/ray_init_m.f90"              # P4
sed -i 's/complex(KIND=rkind) :: zf, zfun0, zfun0_real_arg/complex(KIND=rkind) :: zf/' \
  "$W/damp_fund_ECH.f90"                                                              # P5
# P6: NetCDF writer/reader/check out of ray_results_m; finalize_run no longer calls it
s=$(grep -n '^ *subroutine write_results_NC' "$W/ray_results_m.f90" | cut -d: -f1)
e=$(grep -n '^ *end subroutine check' "$W/ray_results_m.f90" | cut -d: -f1)
sed -i "${s},${e}d" "$W/ray_results_m.f90"
sed -i 's/procedure :: from_module, to_module, read_results_instance_NC/procedure :: from_module, to_module/' \
  "$W/ray_results_m.f90"
sed -i -e 's/& write_results_LD, write_results_NC, run_results/\& write_results_LD, run_results/' \
       -e '/call write_results_NC/d' "$W/finalize_run.f90"
# P7: multiple_mirror equilibrium out
sed -i -e '/use multiple_mirror_eq_m/d' \
       -e "/case ('multiple_mirror')/,/multiple_mirror_eq/d" "$W/equilibrium_m.f90"

# post_process_lib/deposition_profiles_m: the binning itself is plain Fortran; its
# NetCDF writer (I/O only) is cut like P6
s=$(grep -n '^ *subroutine write_deposition_profiles_NC' "$P/post_process_lib/deposition_profiles_m.f90" | cut -d: -f1)
e=$(grep -n '^ *end subroutine check' "$P/post_process_lib/deposition_profiles_m.f90" | cut -d: -f1)
sed -e "${s},${e}d" "$P/post_process_lib/deposition_profiles_m.f90" > "$W/deposition_profiles_m.f90"

MODS="constants_m diagnostics_m
 bcspline ibc_ck v_spline bcspeval cspline cspeval splinck zonfind quick_cube_splines_m
 zfunctions_m quad_trapezoid_m bin_to_uniform_grid_m bisect_m monotonic_function_inversion
 complete_elliptic_int_m vectors3_m
 species_m rf_m slab_eq_m solovev_eq_m solovev_magnetics_m eqdsk_utilities_m
 eqdsk_magnetics_lin_interp_m eqdsk_magnetics_spline_interp_m density_spline_interp_m
 temperature_spline_interp_m axisym_toroid_eq_m equilibrium_m suscep_m damping_m ode_m
 RK4_ode_m SG_ode_m dispersion_solvers_m simple_slab_ray_init_m solovev_ray_init_nphi_ntheta_m
 axisym_toroid_ray_init_R_Z_nphi_ntheta_m one_ray_init_XYZ_k_direction_m file_input_ray_init_m
 ray_init_m ray_results_m openmp_m deposition_profiles_m"
EXTS="check_save damp_fund_ECH deallocate deriv_cold deriv_num disp_solve_cold_n1sq_vs_n3
 disp_solve_cold_nsq_vs_theta disp_solve_n_vs_k_vec eqn_ray finalize_run get_unit_number
 initialize_ode_vector intialize ode_RAYS"

cd "$W"
for m in $MODS $EXTS; do
  $FC $FFLAGS -c "$m.f90" -o "$m.o"
done
LIBOBJS=$(for m in $MODS $EXTS; do echo "$m.o"; done)

# ---- (1) reference CPU path + dump driver ---------------------------------------------------
$FC $FFLAGS -c ray_tracing_ref.f90 -o ray_tracing_ref.o
$FC $FFLAGS -cpp -c "$HERE/ref_dump_driver.f90" -o ref_dump_driver.o
$FC $FFLAGS -o "$OUT/rays_ref_dump" ref_dump_driver.o ray_tracing_ref.o $LIBOBJS

# ---- (1b) the reference's solovev_2_eqdsk (SURVEY 8(d) cfg 5: "eqdsk 129 x 129 written by solovev_2_eqdsk") -----------
awk '/^ *open\(unit=input_unit, file=.rays.in./ { held = $0; next }
     /^ *input_unit = get_unit_number\(\)/ && held != "" { print; print held; held = ""; next }
     { print }' "$P/solovev_2_eqdsk/solovev_2_eqdsk.f90" > "$W/solovev_2_eqdsk.f90"                  # P8
$FC $FFLAGS -c solovev_2_eqdsk.f90 -o solovev_2_eqdsk.o
$FC $FFLAGS -o "$OUT/solovev_2_eqdsk" solovev_2_eqdsk.o $LIBOBJS

# ---- (2) drop-in: same host objects, trace_rays replaced by the HIP shim --------------------
LIBHIP=$ROOT/rays_amd/lib/librays_hip.so
if [ -f "$LIBHIP" ]; then
  $FC $FFLAGS -c "$ROOT/fortran/rays_hip_m.f90" -o rays_hip_m.o
  $FC $FFLAGS -c "$ROOT/fortran/rays_hip_state_m.f90" -o rays_hip_state_m.o
  $FC $FFLAGS -c "$ROOT/fortran/trace_rays_hip.f90" -o trace_rays_hip.o
  # the Solovev ray launcher replaced too: our module takes the place of the
  # reference's solovev_ray_init_nphi_ntheta_m, and ray_init_m is recompiled against it
  # (its own object directory: the replacement modules write .mod files of the same names as the reference's)
  mkdir -p dropin && cp *.mod dropin/ 2>/dev/null || true
  DF="$FFLAGS -module-dir dropin -Idropin"
  $FC $DF -c "$ROOT/fortran/solovev_ray_init_hip.f90" -o solovev_ray_init_hip.o
  $FC $DF -c "$ROOT/fortran/simple_slab_ray_init_hip.f90" -o simple_slab_ray_init_hip.o
  $FC $DF -c "$ROOT/fortran/axisym_toroid_ray_init_hip.f90" -o axisym_toroid_ray_init_hip.o
  $FC $DF -c ray_init_m.f90 -o ray_init_m_dropin.o
  $FC $DF -c "$ROOT/fortran/deposition_profiles_hip.f90" -o deposition_profiles_hip.o
  DROPOBJS=$(for m in $MODS $EXTS; do
      case "$m" in solovev_ray_init_nphi_ntheta_m) echo solovev_ray_init_hip.o;;
                   simple_slab_ray_init_m) echo simple_slab_ray_init_hip.o;;
                   axisym_toroid_ray_init_R_Z_nphi_ntheta_m) echo axisym_toroid_ray_init_hip.o;;
                   ray_init_m) echo ray_init_m_dropin.o;;
                   *) echo "$m.o";; esac; done)
  DROPOBJS="$DROPOBJS deposition_profiles_hip.o"
  $FC $DF -cpp -DRAYS_DROPIN -c "$HERE/ref_dump_driver.f90" -o ref_dump_driver2.o
  $FC $FFLAGS -o "$OUT/rays_hip_dropin" ref_dump_driver2.o trace_rays_hip.o rays_hip_state_m.o rays_hip_m.o $DROPOBJS \
     -L"$ROOT/rays_amd/lib" -lrays_hip -Wl,-rpath,'$ORIGIN/../../rays_amd/lib'
else
  echo "build_ref: $LIBHIP not built yet -- skipping rays_hip_dropin"
fi
echo "build_ref: done -> $(ls "$OUT")"
