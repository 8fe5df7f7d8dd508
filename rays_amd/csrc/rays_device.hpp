// rays_device.hpp -- gfx950 device functions of the RAYS ray-equation right-hand side.
//
// One ray per lane; everything here is straight-line FP64 scalar algebra per lane (no MFMA: the
// RHS is a cold-plasma dispersion determinant and its derivatives, not a contraction).  All
// configuration that the reference selects with strings inside its hot loop
// (ode_m.f90:238, eqn_ray.f90:106,148, equilibrium_m.f90:177, slab_eq_m.f90:172-300) is either a
// template parameter (equilibrium model, species count, derivative model) or a wave-uniform
// branch on kernel-argument data (profile models, ray_param), so lanes never diverge on it.
//
// Numerics contract: IEEE binary64, evaluated in the reference's operation order so that results
// are bit-identical to the Fortran CPU path wherever libm (pow/exp) is not involved.  This file
// must be compiled with -ffp-contract=off and without fast-math; NaN polarity of every comparison
// is part of the contract.
//
// Citations are RAYS_project/RAYS_lib/<file>:<line>.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/rays_hip.h"
#include "rays_libm.hpp"

namespace rays {

#define RAYS_DEV __device__ __forceinline__
// rare per-ray events (a ray starts, stops, crashes): block-frequency hint for the register allocator,
// which otherwise assumes 50 % and keeps their operands in reach on every trip
#define RAYS_RARE(x) __builtin_expect(!!(x), 0)
// First statement of a function body whose arithmetic decides an INDEX or a branch (spline cell searches): in the
// tolerance-flavour translation units (-fassociative-math, FMA contraction) it is evaluated as written all the same.
// Lexical: functions inlined into the body keep their own setting.  (Needs -ffp-contract=fast-honor-pragmas.)
#if defined(__clang__) && defined(RAYS_TOL_FLAVOUR) && !defined(RAYS_HOST_EMUL)
#define RAYS_FP_AS_WRITTEN _Pragma("clang fp reassociate(off) contract(off)")
#else
#define RAYS_FP_AS_WRITTEN
#endif

// ---------------------------------------------------------------------------------------------
// Device parameter block: rays_params_t trimmed to what the kernels read, plus values the
// reference recomputes from constants on every call (same IEEE operations, done once on the host
// by make_dev_params() in rays_capi.hip, compiled -ffp-contract=off).
// Passed by value as a kernel argument -> lives in the kernarg segment, fetched with scalar loads.
// ---------------------------------------------------------------------------------------------
// (in a namespace of its own so that argument-dependent lookup on a DevParams argument does not reach the functions
// of namespace rays: the tolerance translation units hold the arithmetic twice, in rays and in rays::exact)
namespace dev_types {
struct DevParams {
  int nspec, nstep_max, ray_param, nv;
  int damping_model, zf_nx;                 // damping_m.f90:30-40; Z-function spline grid size
  double total_damping_limit, zf_xmin, zf_xmax;
  const double* zf_fspl;                    // device pointer: fsplRe[nx][4] (zfunctions_m.f90)
  unsigned zf_lds;                          // LDS byte address of a staged copy of fsplRe (0 = none): the lookup sits
                                            // at the very end of the RHS' dependency chain, nothing hides its latency
  // axisym_toroid + eqdsk spline magnetics (axisym_toroid_eq_m.f90, eqdsk_magnetics_spline_interp_m.f90)
  int a_n_model, a_nr, a_nz, a_n_rb, a_n_ne, a_n_te, a_n_ti;
  int a_mag_model;                          // RAYS_AXI_MAG_*: eqdsk bicubic spline | analytic Solovev field | eqdsk bilinear
  double a_lin_dR, a_lin_dZ;                // 'eqdsk_magnetics_lin_interp': dR, dZ of eqdsk_utilities_m (half a grid cell);
                                            // its tables sit in a_r_grid, a_z_grid, a_psi_fspl = Psi(nr, nz), a_rb_fspl = T(nr)
  int a_t_model[RAYS_NS0];
  double a_box_rmin, a_box_rmax, a_box_zmin, a_box_zmax, a_psi_limit, a_psiB, a_inv_psiB;
  double a_an1, a_an2, a_d_scrape, a_T_scrape;
  double a_at1[RAYS_NS0], a_at2[RAYS_NS0];
  const double *a_r_grid, *a_z_grid, *a_psi_fspl, *a_rb_grid, *a_rb_fspl, *a_ne_grid, *a_ne_fspl,
      *a_te_grid, *a_te_fspl, *a_ti_grid, *a_ti_fspl;  // device pointers (L2-resident tables)
  // The 1-D tables (rb .. ti) are one contiguous block of a_tab1d_doubles doubles starting at a_rb_grid.  A
  // kernel with LDS to spare copies it there at its start and sets a_lds_tab to the copy's LDS byte address
  // (0 = not staged): the profile lookups n(psi), T(psi), RBphi(R) then cost an LDS access instead of the second
  // of two dependent trips to L2 per evaluation of the equilibrium.
  int a_tab1d_doubles;
  unsigned a_lds_tab;
  // likewise the two grids of the bicubic psi spline (r_grid then z_grid, 2 x ~65 doubles): the cell search reads
  // x(i-1), x(i) twice in a row, each a dependent trip to L2 otherwise
  unsigned a_lds_rz;
  // the six spline grids (r, z, rb, ne, te, ti: kAxR ..): x(1), x(n) and RN(1 / (x(n) - x(1))), so that the uniform-grid
  // estimate of the cell a point lies in needs no table access (rays_device_arith.inc: spl_guess; host: spline_axis)
  double a_axis[6][3];
  int a_tab_off[8];         // offsets (doubles) of rb_grid, rb_fspl, ne_grid, ne_fspl, te_grid, te_fspl, ti_grid, ti_fspl within the
                            // contiguous block of 1-D tables (a_rb_grid ..): what the staged LDS copy is indexed with
  int a_profiles_one_grid;  // 1: Te(psi) and Ti(psi) are tabulated on ONE grid (same length, same values), 2: n(psi) as well:
                            // one cell search serves the lookups that share a grid (host: set_spline_axes)
  double ds, s_max, omgrf, k0, clight, eps0, resid_limit;
  double omgrf2;               // omgrf**2                      equilibrium_m.f90:264
  double two_over_k0;          // 2./k0                         deriv_cold.f90:51
  double m2_over_omgrf;        // -2./omgrf                     deriv_cold.f90:73-74
  double m1_over_omgrf;        // -1./omgrf                     deriv_cold.f90:75
  double rel_err0, abs_err0, sg_error_limit;
  double qs[RAYS_NS0], ms[RAYS_NS0], n0s[RAYS_NS0], t0s[RAYS_NS0], eta[RAYS_NS0];
  double qs2[RAYS_NS0];        // qs**2                         equilibrium_m.f90:263
  double eps0ms[RAYS_NS0];     // eps0*ms                       equilibrium_m.f90:263
  // numerical-derivative constants (deriv_num.f90:37,72-80)
  double delta, two_delta, omgrf_p, omgrf_m, k0_p, k0_m, omgrf2_p, omgrf2_m, omgrf0_delta;
  // slab
  int by_model, bz_model, n_model;
  int t_model[RAYS_NS0];
  double xmin, xmax, ymin, ymax, zmin, zmax;
  double s_rmaj, s_rmin, x0, by0, bz0, LBy, LBz, dBzdx, Ln, dndx, s_an1, s_an2, n_min, LT, dtdx;
  double s_at1[RAYS_NS0], s_at2[RAYS_NS0], T_min[RAYS_NS0];
  double by0_over_LBy, bz0_over_LBz, one_over_Ln, one_over_LT, rmin2;
  // solovev
  int v_n_model;
  int v_t_model[RAYS_NS0];
  double rmaj, kappa, bphi0, psiB, v_an1, v_an2;
  double v_at1[RAYS_NS0], v_at2[RAYS_NS0];
  double box_rmin, box_rmax, box_zmin, box_zmax;
  double bp0, rk, rk2, rmaj2, bphi0_rmaj, half_bp0, bp0_2;
  // Correctly rounded reciprocals RN(1/d) of constant denominators, computed on the host with an
  // IEEE division (see Recip below).
  double inv_k0, inv_omgrf, inv_omgrf2, inv_k0_p, inv_k0_m, inv_omgrf_p, inv_omgrf_m, inv_omgrf2_p,
      inv_omgrf2_m;
  double inv_ms[RAYS_NS0], inv_eps0ms[RAYS_NS0];
  double inv_rk, inv_rk2, inv_rmaj, inv_rmaj2, inv_psiB;
  // tolerance flavour (rays_device_arith.inc: RAYS_TOL_FLAVOUR): gamma_s = |B| * gam_per_b[s], alpha_s = n_s * alp_per_n[s]
  // (equilibrium_m.f90:262-265 with the constant quotients folded on the host)
  double gam_per_b[RAYS_NS0], alp_per_n[RAYS_NS0];
};
}  // namespace dev_types
using dev_types::DevParams;

// ---------------------------------------------------------------------------------------------
// hot_params(): the trace kernels' working copy of the parameter block.
//
// The block is ~170 doubles and a wave has ~100 SGPRs.  LLVM keeps what fits in SGPRs and, for the
// rest, re-issues the scalar kernarg load inside the wave loop right where the value is used, i.e.
// s_load + s_waitcnt lgkmcnt(0) back to back (eight such pairs per RHS evaluation before this
// change).  The per-species constants the RHS reads on every evaluation are therefore moved into
// VECTOR registers once per kernel (an opaque v_mov, so the compiler can neither keep them scalar
// nor re-load them); the one-wave-per-SIMD kernels leave the AGPR half of the register file idle,
// which is where the allocator parks them.  Worth 2 % on the 64k fan.  Values are unchanged:
// bit-identical results.
// ---------------------------------------------------------------------------------------------
#ifdef RAYS_HOST_EMUL
RAYS_DEV double in_vgpr(double x) { return x; }
#else
RAYS_DEV double in_vgpr(double x) {
  double y;
  asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "s"(x));
  return y;
}
#endif
template <int EQ, int NS>
RAYS_DEV void hot_params(const DevParams& P, DevParams& H) {
  H = P;
  constexpr int kHot = NS < 2 ? NS : 2;  // electrons + first ion species (more would spill VGPRs)
#pragma unroll
  for (int is = 0; is < kHot; is++) {
    H.qs[is] = in_vgpr(P.qs[is]);
    H.ms[is] = in_vgpr(P.ms[is]);
    H.inv_ms[is] = in_vgpr(P.inv_ms[is]);
    H.qs2[is] = in_vgpr(P.qs2[is]);
    H.eps0ms[is] = in_vgpr(P.eps0ms[is]);
    H.inv_eps0ms[is] = in_vgpr(P.inv_eps0ms[is]);
    H.n0s[is] = in_vgpr(P.n0s[is]);
    H.t0s[is] = in_vgpr(P.t0s[is]);
#ifdef RAYS_TOL_FLAVOUR
    H.gam_per_b[is] = in_vgpr(P.gam_per_b[is]);
    H.alp_per_n[is] = in_vgpr(P.alp_per_n[is]);
#endif
  }
}

#include "rays_device_arith.inc"

}  // namespace rays
