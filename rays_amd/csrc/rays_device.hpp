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
};

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
  }
}

RAYS_DEV double sq(double x) { return x * x; }
RAYS_DEV double pow4(double x) { return ((x * x) * x) * x; }  // flang lowers x**4 sequentially

// Profile exponents.  The reference evaluates x**alpha with real alpha through libm pow().  pow(x,1)
// = x and pow(x,0) = 1 are exact by definition, and profiles with unit exponents
// (alphan1 = alphan2 = 1, the BASELINE fans) are the common case, so the kernels come in two
// flavours selected on the host (rays_capi.hip: unit_exponents()):
//   UE = true  : every profile exponent in use is exactly 1 -> pow_u<true, Y1> is x (Y1: the call
//                site's exponent is alpha) or 1 (the call site's exponent is alpha - 1); no pow code.
//   UE = false : general exponents, libm::pow (rays_libm.hpp: glibc's pow, bit for bit).
// Keeping pow (eight inlined call sites) out of the unit-exponent kernels is worth 15 % on the 64k
// fan although the code is never executed there: it pushes the kernel past 256 VGPRs and adds SGPR
// spills in the hot loop.
// The template argument EQ of the kernels carries the flag: EQ = model | (UE ? kEqUnitExp : 0).
constexpr int kEqUnitExp = 4;
// multi_spec_damping (damping_m.f90:35, ode_m.f90:169): one absorbed-power row per species behind the total.
// Carried in EQ as well, because nv alone is ambiguous (nv = 12 is integrate_eq_gradients without damping, or
// damping + four species' rows; nv = 13 likewise).
constexpr int kEqMultiSpec = 8;
// Tolerance flavour of a kernel (see fdiv / fsqrt below): same source, compiled in its own translation unit
// with -DRAYS_TOL_FLAVOUR -ffp-contract=fast; the bit only gives the kernel another name.
constexpr int kEqTol = 16;
// Layout of the ODE vector (ode_m.f90:160-173, initialize_ode_vector.f90:25-54):
//   v(1:6) = (r, k), v(7) = s, [v(8) = total absorbed power, [v(9:9+nspec) per species]], [5 gradient rows]
template <bool MULTI, int NS, int NV>
struct RayVec {
  static constexpr bool DAMP = MULTI || NV == 8 || NV == 13;
  static constexpr int NV0 = 7 + (DAMP ? 1 : 0) + (MULTI ? NS : 0);  // rows before the gradient block
  static constexpr bool GRAD = NV == NV0 + 5;
  static_assert(NV == NV0 || NV == NV0 + 5, "nv does not match the damping / gradient options");
};
template <bool UE, bool Y1>
RAYS_DEV double pow_u(double x, double y) {
  if (UE) return Y1 ? x : 1.0;
  if (y == 1.0) return x;
  if (y == 0.0) return 1.0;
  return libm::pow(x, y);
}

// ---------------------------------------------------------------------------------------------
// Shared-reciprocal IEEE division.
//
// The RHS performs ~80 FP64 divisions per evaluation but only ~15 distinct denominators (r, r^2,
// |B|, n_s, dD/dw, k0, omega, ...).  hipcc expands every `a/b` into the full correctly rounded
// sequence (2x v_div_scale, v_rcp, 4 FMA Newton steps, mul, FMA residual, v_div_fmas, v_div_fixup
// = 11 VALU ops, a long dependent chain).  Here the Newton-refined reciprocal is computed ONCE per
// denominator and each quotient costs mul + 2 FMA + v_div_fixup -- the same final steps as the
// compiler's expansion, so the quotient is the same correctly rounded RN(a/b) (Markstein: with
// y = RN(1/b), q = RN(a*y), r = a - b*q (exact, FMA), RN(q + r*y) = RN(a/b)) and the results stay
// bit-identical to the reference's IEEE divisions.  v_div_fixup restores the IEEE special cases
// (0/0, x/0, inf, NaN) from the original operands.  Operands here are far from the over/underflow
// range, so the v_div_scale pre-scaling of the generic expansion is the identity.
// The FMAs below are explicit fused operations (the algorithm needs the exact residual); they are
// not contractions of reference arithmetic, which stays unfused (-ffp-contract=off).
// ---------------------------------------------------------------------------------------------
struct Recip {
  double d, rc;
};
#ifdef RAYS_HOST_EMUL
RAYS_DEV Recip make_recip(double d) { return Recip{d, 0.}; }
RAYS_DEV Recip const_recip(double d, double) { return Recip{d, 0.}; }
RAYS_DEV double div(double a, const Recip& R) { return a / R.d; }
RAYS_DEV double fdiv(double a, double b) { return a / b; }
RAYS_DEV double fsqrt(double x) { return sqrt(x); }
#elif defined(RAYS_TOL_FLAVOUR)
// ---- tolerance flavour (kEqTol kernels; this translation unit is compiled with -ffp-contract=fast) ----------
// north_star's bar for floating point is 1e-10 relative per step with exact ray counts and step indices, not
// bit-identity.  The cold RK4 kernels exist a second time with that bar: quotients are a * (1/d) with the
// reciprocal refined once (v_rcp_f64 + one Newton step: ~1 ulp) instead of the correctly rounded RN(a/d)
// (mul + 2 FMA + v_div_fixup per quotient), square roots are v_rsq_f64 + one coupled Newton step + one residual
// correction (~1 ulp) instead of LLVM's correctly rounded expansion, and the compiler may fuse a*b+c.
// v_div_fixup on the reciprocal keeps 1/0 = inf, 1/inf = 0 and NaN as IEEE has them, so the NaN / inf polarity
// of the comparisons that stop a ray is unchanged.
RAYS_DEV Recip make_recip(double d) {
  double y = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  return Recip{d, __builtin_amdgcn_div_fixup(y, d, 1.0)};
}
RAYS_DEV Recip const_recip(double d, double inv) { return Recip{d, inv}; }
RAYS_DEV double div(double a, const Recip& R) { return a * R.rc; }
RAYS_DEV double fdiv(double a, double b) { return a * make_recip(b).rc; }
RAYS_DEV double fsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  // +-0 and +inf: rsq gives inf / 0 and the products above NaN; sqrt returns the argument itself
  return __builtin_amdgcn_class(x, 0x260) ? x : g;  // 0x260 = -0 | +0 | +inf
}
#else
RAYS_DEV Recip make_recip(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  return Recip{d, y};
}
RAYS_DEV Recip const_recip(double d, double inv) { return Recip{d, inv}; }
RAYS_DEV double div(double a, const Recip& R) {
  const double q = a * R.rc;
  const double r = __builtin_fma(-R.d, q, a);
  const double q1 = __builtin_fma(r, R.rc, q);
  return __builtin_amdgcn_div_fixup(q1, R.d, a);
}
// a / b and sqrt as IEEE has them (the tolerance flavour above replaces both)
RAYS_DEV double fdiv(double a, double b) { return a / b; }
RAYS_DEV double fsqrt(double x) { return sqrt(x); }
#endif

// compiler-rt __divdc3 restricted to (a + 0i)/(c + 0i) -> real part; what flang emits for
// real/complex and complex/real quotients (check_save.f90:226, suscep_m.f90:75).
RAYS_DEV double divdc3_real(double a, double c) {
#if defined(RAYS_TOL_FLAVOUR) && !defined(RAYS_HOST_EMUL)
  return fdiv(a, c);
#endif
  double fc = fabs(c);
  int k = 0;
  // logb(|c|) finite <=> c finite and non-zero
  if (fc > 0.0 && fc < __builtin_inf()) {
    k = ilogb(fc);
    c = scalbn(c, -k);
  }
  double denom = c * c;
  return scalbn((a * c) / denom, -k);
}

// ---------------------------------------------------------------------------------------------
// type eq_point (equilibrium_m.f90:39-59), NS = nspec+1 live species.
// gbt[i][j] = gradbtensor(i+1,j+1) = dB(j)/dx(i).
// ---------------------------------------------------------------------------------------------
template <int NS>
struct EqPoint {
  double bvec[3], bmag, gradbmag[3], bunit[3], gradbunit[3][3], gbt[3][3];
  double ns[NS], gradns[NS][3], ts0, gradts0[3];
  double alpha[NS], gamma[NS];
  double omgc[NS], omgp2[NS];  // cyclotron frequencies (signed) and plasma frequencies squared: deriv_num's omega differences
  double omgc0;  // electron cyclotron frequency (signed), for damp_fund_ECH
  Recip rbmag;   // shared reciprocal of |B|
  int err;
};

// parabolic_prof            slab_eq_m.f90:354-381  (fp := 0 where the reference leaves it undefined)
template <bool UE>
RAYS_DEV void parabolic_prof(double rho, double f_min, double a1, double a2, double& f, double& fp) {
  f = 0.0;
  fp = 0.0;
  if (rho < 1.0) {
    double pr = pow_u<UE, true>(rho, a2);
    f = pow_u<UE, true>(1. - pr, a1);
    fp = -a1 * a2 * pow_u<UE, false>(rho, a2 - 1.) * pow_u<UE, false>(1. - pr, a1 - 1.);
  }
  if (f < f_min) {
    f = f_min;
    fp = 0.0;
  }
}

// slab_eq                   slab_eq_m.f90:125-309
template <int NS, bool UE>
RAYS_DEV int slab_fields(const DevParams& P, const double rvec[3], double bvec[3], double gbt[3][3],
                         double ns[NS], double gradns[NS][3], double ts[NS], double gradts[NS][3],
                         bool check_box) {
  int err = 0;
  const double x = rvec[0], y = rvec[1], z = rvec[2];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) gbt[i][j] = 0.;
#pragma unroll
  for (int is = 0; is < NS; is++) {
    ns[is] = 0.;
    ts[is] = 0.;
#pragma unroll
    for (int i = 0; i < 3; i++) gradns[is][i] = gradts[is][i] = 0.;
  }
  if (x < P.xmin || x > P.xmax) err = RAYS_STOP_X_OUT_OF_BOUNDS;  // :163
  if (y < P.ymin || y > P.ymax) err = RAYS_STOP_Y_OUT_OF_BOUNDS;  // :164
  if (z < P.zmin || z > P.zmax) err = RAYS_STOP_Z_OUT_OF_BOUNDS;  // :165
  if (!check_box) err = 0;

  bvec[0] = 0.;
  bvec[1] = 0.;
  if (P.by_model == RAYS_SLAB_BY_CONSTANT) {  // :184-206
    bvec[1] = P.by0;
  } else if (P.by_model == RAYS_SLAB_BY_TOROID) {
    bvec[1] = fdiv(P.by0, 1. + fdiv(x, P.s_rmaj));
    gbt[0][1] = fdiv(-bvec[1], P.s_rmaj + x);
  } else if (P.by_model == RAYS_SLAB_BY_LINEAR_SHEAR) {
    bvec[1] = fdiv(P.by0 * x, P.LBy);
    gbt[0][1] = P.by0_over_LBy;
  }
  if (P.bz_model == RAYS_SLAB_BZ_CONSTANT) {  // :209-233
    bvec[2] = P.bz0;
  } else if (P.bz_model == RAYS_SLAB_BZ_TOROID) {
    bvec[2] = fdiv(P.bz0, 1. + fdiv(x, P.s_rmaj));
    gbt[0][2] = fdiv(-bvec[2], P.s_rmaj + x);
  } else if (P.bz_model == RAYS_SLAB_BZ_LINEAR) {
    bvec[2] = P.bz0 * (1. + fdiv(x, P.LBz));
    gbt[0][2] = P.bz0_over_LBz;
  } else {
    bvec[2] = P.bz0 + P.dBzdx * (x - P.x0);
    gbt[0][2] = P.dBzdx;
  }
  if (P.n_model == RAYS_SLAB_N_CONSTANT) {  // :237-267
#pragma unroll
    for (int is = 0; is < NS; is++) ns[is] = P.n0s[is];
  } else if (P.n_model == RAYS_SLAB_N_LINEAR) {
    const double f = 1.0 + fdiv(x, P.Ln);
#pragma unroll
    for (int is = 0; is < NS; is++) {
      ns[is] = P.n0s[is] * f;
      gradns[is][0] = P.n0s[is] * P.one_over_Ln;
    }
  } else if (P.n_model == RAYS_SLAB_N_LINEAR_2) {  // value/gradient inconsistency kept (:249-250)
#pragma unroll
    for (int is = 0; is < NS; is++) {
      ns[is] = P.n0s[is] + P.dndx * P.eta[is] * (x - P.x0);
      gradns[is][0] = P.n0s[is] * P.dndx;
    }
  } else if (P.n_model == RAYS_SLAB_N_PARABOLIC) {
    double f, fp;
    parabolic_prof<UE>(x, P.n_min, P.s_an1, P.s_an2, f, fp);
#pragma unroll
    for (int is = 0; is < NS; is++) {
      ns[is] = P.n0s[is] * f;
      gradns[is][0] = P.n0s[is] * fp;
    }
  } else {  // Gaussian
    const double g = libm::exp(-3. * P.s_an1 * sq(fdiv(x, P.s_rmin)));
    const double gp = fdiv(-6. * P.s_an1 * x, P.rmin2);
#pragma unroll
    for (int is = 0; is < NS; is++) {
      ns[is] = P.n0s[is] * g;
      gradns[is][0] = ns[is] * gp;
    }
  }
#pragma unroll
  for (int is = 0; is < NS; is++) {  // :270-301
    const int m = P.t_model[is];
    if (m == RAYS_SLAB_T_CONSTANT) {
      ts[is] = P.t0s[is];
    } else if (m == RAYS_SLAB_T_LINEAR) {
      ts[is] = P.t0s[is] * (1. + fdiv(x, P.LT));
      gradts[is][0] = P.t0s[is] * P.one_over_LT;
    } else if (m == RAYS_SLAB_T_LINEAR_2) {
      ts[is] = P.t0s[is] + P.dtdx * (x - P.x0);
      gradts[is][0] = P.t0s[is] * P.dtdx;
    } else if (m == RAYS_SLAB_T_PARABOLIC) {
      double f, fp;
      parabolic_prof<UE>(x - P.x0, P.T_min[is], P.s_at1[is], P.s_at2[is], f, fp);
      ts[is] = P.t0s[is] * f;
      gradts[is][0] = P.t0s[is] * fp;
    }
  }
  if (check_box && err == 0) {  // :305-306 (only reached in-box)
    double mn = ns[0], mt = ts[0];
#pragma unroll
    for (int is = 1; is < NS; is++) {
      if (ns[is] < mn) mn = ns[is];
      if (ts[is] < mt) mt = ts[is];
    }
    if (mn < 0.) err = RAYS_STOP_NEGATIVE_DENS;
    if (mt < 0.) err = RAYS_STOP_NEGATIVE_TEMP;
  }
  return err;
}

// Magnetics of the Solovev equilibrium: B, grad B tensor, psiN, grad psiN at (x, y, z), r = sqrt(x^2 + y^2).
//   solovev_eq_m.f90:159-204 + solovev_psi :308-318  ==  solovev_magnetics_m.f90:154-181 + :199-207, term for term
RAYS_DEV void solovev_magnetics(const DevParams& P, double x, double y, double z, double r, double bvec[3],
                                double gbt[3][3], double& psiN, double gradpsiN[3]) {
  const double bp0 = P.bp0;
  const Recip Rr = make_recip(r), Rr2 = make_recip(sq(r));
  const Recip Rrk = const_recip(P.rk, P.inv_rk), Rrk2 = const_recip(P.rk2, P.inv_rk2);
  const Recip Rrmaj = const_recip(P.rmaj, P.inv_rmaj), Rrmaj2 = const_recip(P.rmaj2, P.inv_rmaj2);
  const Recip RpsiB = const_recip(P.psiB, P.inv_psiB);
  // :170-172 (the same br, bz appear in solovev_psi :312-313)
  const double br = div(-bp0 * r * z, Rrk2);
  const double bz = bp0 * (sq(div(z, Rrk)) + .5 * (sq(div(r, Rrmaj)) - 1.));
  // solovev_psi :308-318
  const double psi = P.half_bp0 * (sq(div(r * z, Rrk)) + div(sq(r * r - P.rmaj2), Rrmaj2) * 0.25);
  const double gradpsi[3] = {x * bz, y * bz, -r * br};
  psiN = div(psi, RpsiB);
  gradpsiN[0] = div(gradpsi[0], RpsiB);
  gradpsiN[1] = div(gradpsi[1], RpsiB);
  gradpsiN[2] = div(gradpsi[2], RpsiB);

  const double bphi = div(P.bphi0_rmaj, Rr);
  const double br_r = div(br, Rr), bphi_r = div(bphi, Rr);  // br/r, bphi/r (each appears 3x)
  const double dbrdr = br_r;
  const double dbrdz = div(-bp0 * r, Rrk2);
  const double dbzdr = div(bp0 * r, Rrmaj2);
  const double dbzdz = div(P.bp0_2 * z, Rrk2);
  const double dbphidr = -bphi_r;
  bvec[0] = div(br * x, Rr) - div(bphi * y, Rr);  // :187-189
  bvec[1] = div(br * y, Rr) + div(bphi * x, Rr);
  bvec[2] = bz;
  const double x2 = sq(x), y2 = sq(y);
  gbt[0][0] = div(dbrdr * x2 + div(br * y2, Rr) + (-dbphidr + bphi_r) * x * y, Rr2);  // :192-204
  gbt[1][0] = div((dbrdr - br_r) * x * y - dbphidr * y2 - div(bphi * x2, Rr), Rr2);
  gbt[2][0] = div(dbrdz * x, Rr);
  gbt[0][1] = div((dbrdr - br_r) * x * y + dbphidr * x2 + div(bphi * y2, Rr), Rr2);
  gbt[1][1] = div(dbrdr * y2 + div(br * x2, Rr) + (dbphidr - bphi_r) * x * y, Rr2);
  gbt[2][1] = div(dbrdz * y, Rr);
  gbt[0][2] = div(dbzdr * x, Rr);
  gbt[1][2] = div(dbzdr * y, Rr);
  gbt[2][2] = dbzdz;
}

// solovev_eq + solovev_psi  solovev_eq_m.f90:122-276, 280-322
template <int NS, bool UE>
RAYS_DEV int solovev_fields(const DevParams& P, const double rvec[3], double bvec[3],
                            double gbt[3][3], double ns[NS], double gradns[NS][3], double ts[NS],
                            double gradts[NS][3], bool check_box) {
  int err = 0;
  const double x = rvec[0], y = rvec[1], z = rvec[2];
  const double r = fsqrt(x * x + y * y);
  if (r < P.box_rmin || r > P.box_rmax) err = RAYS_STOP_R_OUT_OF_BOX;  // :155
  if (z < P.box_zmin || z > P.box_zmax) err = RAYS_STOP_Z_OUT_OF_BOX;  // :156
  if (!check_box) err = 0;
  double psiN, gradpsiN[3];
  solovev_magnetics(P, x, y, z, r, bvec, gbt, psiN, gradpsiN);

#pragma unroll
  for (int is = 0; is < NS; is++) {
    ns[is] = 0.;
    ts[is] = 0.;
#pragma unroll
    for (int i = 0; i < 3; i++) gradns[is][i] = gradts[is][i] = 0.;
  }
  if (P.v_n_model == RAYS_SOLOVEV_N_CONSTANT) {  // :210-212
#pragma unroll
    for (int is = 0; is < NS; is++) ns[is] = P.n0s[is];
  } else if (psiN < 1.0) {  // :214-225
    const double a1 = P.v_an1, a2 = P.v_an2;
    const double pr = pow_u<UE, true>(psiN, a2);
    const double prof = pow_u<UE, true>(1. - pr, a1);
    const double dd_psi = -a1 * a2 * pow_u<UE, false>(psiN, a2 - 1.) * pow_u<UE, false>(1. - pr, a1 - 1.);
#pragma unroll
    for (int is = 0; is < NS; is++) {
      ns[is] = P.n0s[is] * prof;
      const double c = P.n0s[is] * dd_psi;
      gradns[is][0] = c * gradpsiN[0];
      gradns[is][1] = c * gradpsiN[1];
      gradns[is][2] = c * gradpsiN[2];
    }
  }
  // temperature :235-268 -- 'parabolic' zeroes the whole ts/gradts arrays inside the species loop
  // and uses exponent alphat1 in the gradient: kept as in the reference.
#pragma unroll
  for (int is = 0; is < NS; is++) {
    if (P.v_t_model[is] == RAYS_SOLOVEV_T_PARABOLIC) {
#pragma unroll
      for (int j = 0; j < NS; j++) {
        ts[j] = 0.;
        gradts[j][0] = gradts[j][1] = gradts[j][2] = 0.;
      }
      if (psiN < 1.) {
        const double a1 = P.v_at1[is], a2 = P.v_at2[is];
        const double pf = pow_u<UE, true>(1. - pow_u<UE, true>(psiN, a2), a1);
        ts[is] = P.t0s[is] * pf;
        const double dd_psi = -a1 * a2 * pow_u<UE, false>(psiN, a2 - 1.) * pf;
        const double c = P.t0s[is] * dd_psi;
        gradts[is][0] = c * gradpsiN[0];
        gradts[is][1] = c * gradpsiN[1];
        gradts[is][2] = c * gradpsiN[2];
      }
    }
  }
  if (check_box && err == 0) {  // :272-273
    double mn = ns[0], mt = ts[0];
#pragma unroll
    for (int is = 1; is < NS; is++) {
      if (ns[is] < mn) mn = ns[is];
      if (ts[is] < mt) mt = ts[is];
    }
    if (mn < 0.) err = RAYS_STOP_NEGATIVE_DENS;
    if (mt < 0.) err = RAYS_STOP_NEGATIVE_TEMP;
  }
  return err;
}

// ---------------------------------------------------------------------------------------------
// PPPL-pspline evaluation on uniform grids (cspevx/bcspevxy cell lookup + cspevfn/bcspevfn Horner
// forms; splines_lib/cspeval.f90, bcspeval.f90).  Targets are clamped to the grid (the reference
// clamps within 4e-7*max|x| and otherwise leaves the outputs undefined).  Tables live in global
// memory: 16 doubles per bicubic cell (128 B = one cache line), 4 per cubic cell; neighbouring rays
// hit the same or adjacent cells, so they stay L2-resident (65x65 cells = 540 KB).
// ---------------------------------------------------------------------------------------------
#ifdef RAYS_HOST_EMUL
typedef const double* eq_lds_ptr;
#else
typedef const __attribute__((address_space(3))) double* eq_lds_ptr;
#endif
template <class PTR>
RAYS_DEV int spl_cell(PTR x, int nx, double xget, double& dx) {
  RAYS_FP_AS_WRITTEN
  const double x1 = x[0], xn = x[nx - 1];
  double z = xget;
  if (z < x1) z = x1;
  if (z > xn) z = xn;
  const int nxm = nx - 1;
  int i = (int)(1 + fdiv(nxm * (z - x1), xn - x1));
  i = i < nxm ? i : nxm;
  i = i < 1 ? 1 : i;
  if (z < x[i - 1]) i = i - 1;
  else if (z > x[i]) i = i + 1;
  i = i < 1 ? 1 : (i > nxm ? nxm : i);
  dx = z - x[i - 1];
  return i;
}

template <class PTR>
RAYS_DEV void spl1_fp(PTR grid, PTR fspl, int n, double x, double& f, double& fp) {
  double dx;
  const int i = spl_cell(grid, n, x, dx);
  const PTR c = fspl + 4 * (i - 1);
  const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
  f = c0 + dx * (c1 + dx * (c2 + dx * c3));
  fp = c1 + dx * (2.0 * c2 + dx * 3.0 * c3);
}

// eval_2D_fpp: f, fx, fy, fxx, fxy, fyy (quick_cube_splines_m.f90:305-332, bcspevfn ict = 1,1,1,1,1,1)
RAYS_DEV void spl2_fpp(const DevParams& P, double x, double y, double out[6]) {
  double dx, dy;
  int i, j;
#ifndef RAYS_HOST_EMUL
  if (P.a_lds_rz) {  // wave-uniform
    const eq_lds_ptr rg = (eq_lds_ptr)(unsigned long long)P.a_lds_rz;
    i = spl_cell<eq_lds_ptr>(rg, P.a_nr, x, dx);
    j = spl_cell<eq_lds_ptr>(rg + P.a_nr, P.a_nz, y, dy);
  } else
#endif
  {
    i = spl_cell<const double*>(P.a_r_grid, P.a_nr, x, dx);
    j = spl_cell<const double*>(P.a_z_grid, P.a_nz, y, dy);
  }
  const double* c = P.a_psi_fspl + 16 * ((long long)(i - 1) + (long long)P.a_nr * (long long)(j - 1));
  double F[4][4];  // F[a-1][b-1] = f(a,b,i,j)
#pragma unroll
  for (int b = 0; b < 4; b++)
#pragma unroll
    for (int a = 0; a < 4; a++) F[a][b] = c[a + 4 * b];
  out[0] = F[0][0] + dy * (F[0][1] + dy * (F[0][2] + dy * F[0][3])) +
           dx * (F[1][0] + dy * (F[1][1] + dy * (F[1][2] + dy * F[1][3])) +
           dx * (F[2][0] + dy * (F[2][1] + dy * (F[2][2] + dy * F[2][3])) +
           dx * (F[3][0] + dy * (F[3][1] + dy * (F[3][2] + dy * F[3][3])))));
  out[1] = F[1][0] + dy * (F[1][1] + dy * (F[1][2] + dy * F[1][3])) +
           2.0 * dx * (F[2][0] + dy * (F[2][1] + dy * (F[2][2] + dy * F[2][3])) +
           1.5 * dx * (F[3][0] + dy * (F[3][1] + dy * (F[3][2] + dy * F[3][3]))));
  out[2] = F[0][1] + dy * (2.0 * F[0][2] + dy * 3.0 * F[0][3]) +
           dx * (F[1][1] + dy * (2.0 * F[1][2] + dy * 3.0 * F[1][3]) +
           dx * (F[2][1] + dy * (2.0 * F[2][2] + dy * 3.0 * F[2][3]) +
           dx * (F[3][1] + dy * (2.0 * F[3][2] + dy * 3.0 * F[3][3]))));
  out[3] = 2.0 * (F[2][0] + dy * (F[2][1] + dy * (F[2][2] + dy * F[2][3]))) +
           6.0 * dx * (F[3][0] + dy * (F[3][1] + dy * (F[3][2] + dy * F[3][3])));  // fxx
  out[5] = 2.0 * F[0][2] + 6.0 * dy * F[0][3] +
           dx * (2.0 * F[1][2] + 6.0 * dy * F[1][3] +
           dx * (2.0 * F[2][2] + 6.0 * dy * F[2][3] + dx * (2.0 * F[3][2] + 6.0 * dy * F[3][3])));  // fyy
  out[4] = F[1][1] + dy * (2.0 * F[1][2] + dy * 3.0 * F[1][3]) +
           2. * dx * (F[2][1] + dy * (2.0 * F[2][2] + dy * 3.0 * F[2][3]) +
           1.5 * dx * (F[3][1] + dy * (2.0 * F[3][2] + dy * 3.0 * F[3][3])));  // fxy
}

// 1-D table lookup through the staged LDS copy when there is one (DevParams::a_lds_tab), else through L2
RAYS_DEV void spl1_tab(const DevParams& P, const double* grid, const double* fspl, int n, double x, double& f,
                       double& fp) {
#ifndef RAYS_HOST_EMUL
  if (P.a_lds_tab) {  // wave-uniform
    const eq_lds_ptr base = (eq_lds_ptr)(unsigned long long)P.a_lds_tab;
    spl1_fp<eq_lds_ptr>(base + (grid - P.a_rb_grid), base + (fspl - P.a_rb_grid), n, x, f, fp);
    return;
  }
#endif
  spl1_fp<const double*>(grid, fspl, n, x, f, fp);
}

// ---- 'eqdsk_magnetics_lin_interp': eqdsk_utilities_m.f90:144-306 + eqdsk_magnetics_lin_interp_m.f90:146-214 ----
// GetPsi: bilinear in the cell i = 1 + int((R - R_grid(1))/(R_grid(2) - R_grid(1))), likewise j.  The reference
// does not bound i, j: its central differences reach one cell beyond the grid for points in the outermost cells
// and then read the neighbouring column through Fortran's storage order.  Same here (flat index); only an index
// outside the array altogether -- undefined in the reference -- is clamped.
RAYS_DEV double eqlin_psi_at(const DevParams& P, int i, int j) {  // Psi(i, j), 1-based
  long long k = (long long)(i - 1) + (long long)(j - 1) * P.a_nr;
  const long long n = (long long)P.a_nr * P.a_nz;
  k = k < 0 ? 0 : (k >= n ? n - 1 : k);
  return P.a_psi_fspl[k];
}
RAYS_DEV double eqlin_getpsi(const DevParams& P, double R, double Z) {  // :144-162
  const double r1 = P.a_r_grid[0], z1 = P.a_z_grid[0];
  const Recip Rhr = make_recip(P.a_r_grid[1] - r1), Rhz = make_recip(P.a_z_grid[1] - z1);
  const int i = 1 + (int)div(R - r1, Rhr);
  const int j = 1 + (int)div(Z - z1, Rhz);
  const int ic = i < 1 ? 1 : (i > P.a_nr ? P.a_nr : i), jc = j < 1 ? 1 : (j > P.a_nz ? P.a_nz : j);  // (memory safety only)
  const double x = div(R - P.a_r_grid[ic - 1], Rhr);
  const double y = div(Z - P.a_z_grid[jc - 1], Rhz);
  const double omx = 1. - x, omy = 1. - y;
  return ((eqlin_psi_at(P, i, j) * omx * omy + eqlin_psi_at(P, i + 1, j) * x * omy) + eqlin_psi_at(P, i, j + 1) * omx * y) +
         eqlin_psi_at(P, i + 1, j + 1) * x * y;
}
RAYS_DEV double eqlin_getrbphi(const DevParams& P, double R) {  // :168-184
  const double r1 = P.a_r_grid[0];
  const Recip Rhr = make_recip(P.a_r_grid[1] - r1);
  const int i = 1 + (int)div(R - r1, Rhr);
  const int ic = i < 1 ? 1 : (i > P.a_nr - 1 ? P.a_nr - 1 : i);  // (memory safety only: T(i), T(i+1))
  const double x = div(R - P.a_r_grid[ic - 1], Rhr);
  return P.a_rb_fspl[ic - 1] * (1. - x) + P.a_rb_fspl[ic] * x;
}
// psi, dPsi/dR, dPsi/dZ by the reference's differences (GetPsiR :190-204, GetPsiZ :210-223)
RAYS_DEV void eqlin_psi_grad(const DevParams& P, double R, double Z, double& psi, double& PsiR, double& PsiZ) {
  const double dR = P.a_lin_dR, dZ = P.a_lin_dZ;
  psi = eqlin_getpsi(P, R, Z);
  PsiR = (eqlin_getpsi(P, R + dR, Z) - eqlin_getpsi(P, R - dR, Z)) / 2. / dR;
  PsiZ = (eqlin_getpsi(P, R, Z + dZ) - eqlin_getpsi(P, R, Z - dZ)) / 2. / dZ;
}
// eqdsk_magnetics_lin_interp (eqdsk_magnetics_lin_interp_m.f90:146-214)
RAYS_DEV void eqlin_magnetics(const DevParams& P, double x, double y, double z, double r, double bvec[3],
                              double gbt[3][3], double& psiN, double gradpsiN[3]) {
  const double dR = P.a_lin_dR, dZ = P.a_lin_dZ;
  double psi, PsiR, PsiZ;
  eqlin_psi_grad(P, r, z, psi, PsiR, PsiZ);
  const Recip Rr = make_recip(r), Rr2 = make_recip(sq(r));
  const Recip RpsiB = const_recip(P.a_psiB, P.a_inv_psiB);
  const double br = div(-PsiZ, Rr);                            // :174
  const double bz = div(PsiR, Rr);                             // :175
  const double bphi = div(eqlin_getrbphi(P, r), Rr);           // :176
  const double gradpsi[3] = {x * bz, y * bz, -r * br};         // :178
  psiN = div(psi, RpsiB);
  gradpsiN[0] = div(gradpsi[0], RpsiB);
  gradpsiN[1] = div(gradpsi[1], RpsiB);
  gradpsiN[2] = div(gradpsi[2], RpsiB);
  // GetPsiRZ (:271-287), GetPsiZZ (:251-265), GetPsiRR (:229-245), GetRBphiR (:293-306)
  const double PsiRZ = (((eqlin_getpsi(P, r + dR, z + dZ) - eqlin_getpsi(P, r - dR, z + dZ)) - eqlin_getpsi(P, r + dR, z - dZ)) +
                        eqlin_getpsi(P, r - dR, z - dZ)) / 4. / dR / dZ;
  const double PsiZZ = ((eqlin_getpsi(P, r, z + 2. * dZ) - 2. * psi) + eqlin_getpsi(P, r, z - 2. * dZ)) / dZ / dZ;
  const double PsiRR = ((eqlin_getpsi(P, r + 2. * dR, z) - 2. * psi) + eqlin_getpsi(P, r - 2. * dR, z)) / dR / dR;
  const double RBphiR = (eqlin_getrbphi(P, r + dR) - eqlin_getrbphi(P, r - dR)) / 2. / dR;
  const double br_r = div(br, Rr), bphi_r = div(bphi, Rr);
  const double dbrdr = -br_r - div(PsiRZ, Rr);                 // :184
  const double dbrdz = div(-PsiZZ, Rr);                        // :185
  const double dbzdr = div(-bz, Rr) + div(PsiRR, Rr);          // :186
  const double dbzdz = div(PsiRZ, Rr);                         // :187
  const double dbphidr = div(RBphiR - bphi, Rr);               // :188
  bvec[0] = div(br * x, Rr) - div(bphi * y, Rr);
  bvec[1] = div(br * y, Rr) + div(bphi * x, Rr);
  bvec[2] = bz;
  const double x2 = sq(x), y2 = sq(y);
  gbt[0][0] = div(dbrdr * x2 + div(br * y2, Rr) + (-dbphidr + bphi_r) * x * y, Rr2);
  gbt[1][0] = div((dbrdr - br_r) * x * y - dbphidr * y2 - div(bphi * x2, Rr), Rr2);
  gbt[2][0] = div(dbrdz * x, Rr);
  gbt[0][1] = div((dbrdr - br_r) * x * y + dbphidr * x2 + div(bphi * y2, Rr), Rr2);
  gbt[1][1] = div(dbrdr * y2 + div(br * x2, Rr) + (dbphidr - bphi_r) * x * y, Rr2);
  gbt[2][1] = div(dbrdz * y, Rr);
  gbt[0][2] = div(dbzdr * x, Rr);
  gbt[1][2] = div(dbzdr * y, Rr);
  gbt[2][2] = dbzdz;
}

// axisym_toroid_eq + eqdsk_magnetics_spline_interp
//   axisym_toroid_eq_m.f90:215-362, eqdsk_magnetics_spline_interp_m.f90:206-282,
//   density_spline_interp_m.f90:109-130, temperature_spline_interp_m.f90
template <int NS, bool UE>
RAYS_DEV int axisym_fields(const DevParams& P, const double rvec[3], double bvec[3], double gbt[3][3],
                           double ns[NS], double gradns[NS][3], double ts[NS], double gradts[NS][3],
                           bool check_box) {
  constexpr double Tiny = 10.0e-14;
  int err = 0;
  const double x = rvec[0], y = rvec[1], z = rvec[2];
  const double r = fsqrt(x * x + y * y);
  if (r < P.a_box_rmin - Tiny || r > P.a_box_rmax + Tiny) err = RAYS_STOP_AXI_R_OUT_OF_BOX;  // :261-264
  if (z < P.a_box_zmin - Tiny || z > P.a_box_zmax + Tiny) err = RAYS_STOP_AXI_Z_OUT_OF_BOX;  // :265-268
  bool boxed = check_box && err != 0;  // reference returns here; fields below are then unused
  double psiN, gpN[3];
  if (P.a_mag_model == RAYS_AXI_MAG_SOLOVEV) {  // wave-uniform
    // solovev_magnetics (solovev_magnetics_m.f90:127-181): its own test of the same box, without the 1e-13
    // margin of the test above ('R out_of_bounds' / 'z out_of_bounds'; the reference then goes on with an
    // undefined psiN -- here the ray stops with that flag)
    int merr = 0;
    if (r < P.box_rmin || r > P.box_rmax) merr = RAYS_STOP_SOLMAG_R_OUT_OF_BOUNDS;
    if (z < P.box_zmin || z > P.box_zmax) merr = RAYS_STOP_SOLMAG_Z_OUT_OF_BOUNDS;
    if (check_box && !boxed && merr != 0) {
      boxed = true;
      err = merr;
    }
    solovev_magnetics(P, x, y, z, r, bvec, gbt, psiN, gpN);
  } else if (P.a_mag_model == RAYS_AXI_MAG_EQDSK_LIN) {  // wave-uniform
    eqlin_magnetics(P, x, y, z, r, bvec, gbt, psiN, gpN);
  } else {
    double f6[6], RBphi, RBphiR;
    spl2_fpp(P, r, z, f6);
    const double psi = f6[0], PsiR = f6[1], PsiZ = f6[2], PsiRR = f6[3], PsiRZ = f6[4], PsiZZ = f6[5];
    spl1_tab(P, P.a_rb_grid, P.a_rb_fspl, P.a_n_rb, r, RBphi, RBphiR);
    const Recip Rr = make_recip(r), Rr2 = make_recip(sq(r));
    const Recip RpsiB = const_recip(P.a_psiB, P.a_inv_psiB);
    const double br = div(PsiZ, Rr), bz = div(-PsiR, Rr), bphi = div(RBphi, Rr);
    const double gradpsi[3] = {-x * bz, -y * bz, r * br};
    psiN = div(psi, RpsiB);
    gpN[0] = div(gradpsi[0], RpsiB);
    gpN[1] = div(gradpsi[1], RpsiB);
    gpN[2] = div(gradpsi[2], RpsiB);
    const double br_r = div(br, Rr), bphi_r = div(bphi, Rr);
    const double dbrdr = -br_r + div(PsiRZ, Rr);
    const double dbrdz = div(PsiZZ, Rr);
    const double dbzdr = div(-bz, Rr) - div(PsiRR, Rr);
    const double dbzdz = div(-PsiRZ, Rr);
    const double dbphidr = div(RBphiR - bphi, Rr);
    bvec[0] = div(br * x, Rr) - div(bphi * y, Rr);
    bvec[1] = div(br * y, Rr) + div(bphi * x, Rr);
    bvec[2] = bz;
    const double x2 = sq(x), y2 = sq(y);
    gbt[0][0] = div(dbrdr * x2 + div(br * y2, Rr) + (-dbphidr + bphi_r) * x * y, Rr2);
    gbt[1][0] = div((dbrdr - br_r) * x * y - dbphidr * y2 - div(bphi * x2, Rr), Rr2);
    gbt[2][0] = div(dbrdz * x, Rr);
    gbt[0][1] = div((dbrdr - br_r) * x * y + dbphidr * x2 + div(bphi * y2, Rr), Rr2);
    gbt[1][1] = div(dbrdr * y2 + div(br * x2, Rr) + (dbphidr - bphi_r) * x * y, Rr2);
    gbt[2][1] = div(dbrdz * y, Rr);
    gbt[0][2] = div(dbzdr * x, Rr);
    gbt[1][2] = div(dbzdr * y, Rr);
    gbt[2][2] = dbzdz;
  }
  if (!boxed) err = 0;
  if (!boxed && psiN > P.a_psi_limit) err = RAYS_STOP_OUT_OF_PLASMA;  // :288

  if (P.a_n_model == RAYS_AXI_N_CONSTANT) {  // :290-312
#pragma unroll
    for (int is = 0; is < NS; is++) {
      ns[is] = P.n0s[is];
      gradns[is][0] = gradns[is][1] = gradns[is][2] = 0.;
    }
  } else {
    double dens = 0., dd_psi = 0.;
    if (P.a_n_model == RAYS_AXI_N_PARABOLIC) {
      parabolic_prof<UE>(psiN, P.a_d_scrape, P.a_an1, P.a_an2, dens, dd_psi);
    } else {
      if (psiN <= 1.0) spl1_tab(P, P.a_ne_grid, P.a_ne_fspl, P.a_n_ne, psiN, dens, dd_psi);
      if (dens < P.a_d_scrape) {
        dens = P.a_d_scrape;
        dd_psi = 0.;
      }
    }
#pragma unroll
    for (int is = 0; is < NS; is++) {
      ns[is] = P.n0s[is] * dens;
      const double c = P.n0s[is] * dd_psi;
      gradns[is][0] = c * gpN[0];
      gradns[is][1] = c * gpN[1];
      gradns[is][2] = c * gpN[2];
    }
  }
#pragma unroll
  for (int is = 0; is < NS; is++) {
    ts[is] = 0.;
    gradts[is][0] = gradts[is][1] = gradts[is][2] = 0.;
  }
#pragma unroll
  for (int is = 0; is < NS; is++) {  // :314-354
    const int m = P.a_t_model[is];
    if (m == RAYS_AXI_T_CONSTANT) {
      ts[is] = P.t0s[is];
#pragma unroll
      for (int j = 0; j < NS; j++) gradts[j][0] = gradts[j][1] = gradts[j][2] = 0.;  // `gradts = 0.` (:328)
    } else if (m == RAYS_AXI_T_PARABOLIC) {
      double t_prof, dt_dpsi;
      parabolic_prof<UE>(psiN, P.a_T_scrape, P.a_at1[is], P.a_at2[is], t_prof, dt_dpsi);
      ts[is] = P.t0s[is] * t_prof;
      const double c = P.t0s[is] * dt_dpsi;
      gradts[is][0] = c * gpN[0];
      gradts[is][1] = c * gpN[1];
      gradts[is][2] = c * gpN[2];
    } else if (m == RAYS_AXI_T_SPLINE) {
      double Te = 0., dTe = 0., Ti = 0., dTi = 0.;
      if (psiN <= 1.0) {
        spl1_tab(P, P.a_te_grid, P.a_te_fspl, P.a_n_te, psiN, Te, dTe);
        spl1_tab(P, P.a_ti_grid, P.a_ti_fspl, P.a_n_ti, psiN, Ti, dTi);
      }
      if (Te < P.a_T_scrape) { Te = P.a_T_scrape; dTe = 0.; }
      if (Ti < P.a_T_scrape) { Ti = P.a_T_scrape; dTi = 0.; }
      const double T = is == 0 ? Te : Ti, dT = is == 0 ? dTe : dTi;
      ts[is] = P.t0s[is] * T;
      const double c = P.t0s[is] * dT;
      gradts[is][0] = c * gpN[0];
      gradts[is][1] = c * gpN[1];
      gradts[is][2] = c * gpN[2];
    }
  }
  if (check_box && !boxed) {  // :358-359
    double mn = ns[0], mt = ts[0];
#pragma unroll
    for (int is = 1; is < NS; is++) {
      if (ns[is] < mn) mn = ns[is];
      if (ts[is] < mt) mt = ts[is];
    }
    if (mn < 0.) err = RAYS_STOP_NEGATIVE_DENS;
    if (mt < 0.) err = RAYS_STOP_NEGATIVE_TEMP;
  }
  return check_box ? err : 0;
}

// equilibrium               equilibrium_m.f90:135-272
// omgrf / omgrf2 are arguments because deriv_num re-evaluates the equilibrium at omgrf(1 +- delta/2)
// (deriv_num.f90:72-79); on the device these are per-call values, which also removes the
// reference's data race on the module variables.
template <int EQ, int NS>
RAYS_DEV void equilibrium(const DevParams& P, const Recip& Romgrf, const Recip& Romgrf2,
                          const double rvec[3], EqPoint<NS>& eq, bool check_box) {
  double ns[NS], gradns[NS][3], ts[NS], gradts[NS][3];
  int err;
  constexpr int MODEL = EQ & 3;
  constexpr bool UE = (EQ & kEqUnitExp) != 0;
  if (MODEL == RAYS_EQ_SLAB)
    err = slab_fields<NS, UE>(P, rvec, eq.bvec, eq.gbt, ns, gradns, ts, gradts, check_box);
  else if (MODEL == RAYS_EQ_SOLOVEV)
    err = solovev_fields<NS, UE>(P, rvec, eq.bvec, eq.gbt, ns, gradns, ts, gradts, check_box);
  else
    err = axisym_fields<NS, UE>(P, rvec, eq.bvec, eq.gbt, ns, gradns, ts, gradts, check_box);
  eq.err = err;
  // When err != 0 the reference returns with eq undefined (:198-202).  We still fill it (fields
  // evaluated at the out-of-box point): callers that stop on err never read it, and check_save,
  // which does read it, then sees defined data (DESIGN.md "defined where the reference is not").
  const double bmag = fsqrt(sq(eq.bvec[0]) + sq(eq.bvec[1]) + sq(eq.bvec[2]));  // :238
  eq.bmag = bmag;
  const Recip Rb = make_recip(bmag);
  eq.rbmag = Rb;
#pragma unroll
  for (int i = 0; i < 3; i++) eq.bunit[i] = div(eq.bvec[i], Rb);
#pragma unroll
  for (int i = 0; i < 3; i++)  // :244-246
    eq.gradbmag[i] = eq.gbt[i][0] * eq.bunit[0] + eq.gbt[i][1] * eq.bunit[1] + eq.gbt[i][2] * eq.bunit[2];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)  // :254-257
      eq.gradbunit[i][j] = div(eq.gbt[i][j] - eq.gradbmag[i] * eq.bunit[j], Rb);
#pragma unroll
  for (int is = 0; is < NS; is++) {  // :262-265
    eq.ns[is] = ns[is];
#pragma unroll
    for (int i = 0; i < 3; i++) eq.gradns[is][i] = gradns[is][i];
    const double omgc = div(P.qs[is] * bmag, const_recip(P.ms[is], P.inv_ms[is]));
    const double omgp2 = div(ns[is] * P.qs2[is], const_recip(P.eps0ms[is], P.inv_eps0ms[is]));
    eq.alpha[is] = div(omgp2, Romgrf2);
    eq.gamma[is] = div(omgc, Romgrf);
    eq.omgc[is] = omgc;
    eq.omgp2[is] = omgp2;
    if (is == 0) eq.omgc0 = omgc;
  }
  eq.ts0 = ts[0];
#pragma unroll
  for (int i = 0; i < 3; i++) eq.gradts0[i] = gradts[0][i];
}

// deriv_cold                deriv_cold.f90:1-228
template <int NS>
RAYS_DEV void deriv_cold(const DevParams& P, const EqPoint<NS>& eq, const double nvec[3],
                         double dddx[3], double dddk[3], double& dddw) {
  const Recip Rk0 = const_recip(P.k0, P.inv_k0);
  double alpha[NS], gamma[NS];
#pragma unroll
  for (int is = 0; is < NS; is++) {
    alpha[is] = eq.alpha[is];
    gamma[is] = eq.gamma[is];
  }
  const double n3 = nvec[0] * eq.bunit[0] + nvec[1] * eq.bunit[1] + nvec[2] * eq.bunit[2];  // :45
  double np[3];
#pragma unroll
  for (int i = 0; i < 3; i++) np[i] = nvec[i] - n3 * eq.bunit[i];
  const double n1 = fsqrt(sq(np[0]) + sq(np[1]) + sq(np[2]));  // :46
  double dn3dk[3], dn12dk[3], dn3dx[3], dn12dx[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    dn3dk[i] = div(eq.bunit[i], Rk0);         // :50
    dn12dk[i] = P.two_over_k0 * np[i];        // :51
    dn3dx[i] = eq.gradbunit[i][0] * nvec[0] + eq.gradbunit[i][1] * nvec[1] + eq.gradbunit[i][2] * nvec[2];
    dn12dx[i] = -2. * n3 * dn3dx[i];          // :57
  }
  double dadx[3][NS], dgdx[3][NS];
#pragma unroll
  for (int is = 0; is < NS; is++) {
    const Recip Rns = make_recip(eq.ns[is]);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      dadx[i][is] = div(eq.alpha[is] * eq.gradns[is][i], Rns);  // :64  (0*0/0 = NaN outside plasma)
      dgdx[i][is] = div(gamma[is] * eq.gradbmag[i], eq.rbmag);  // :65
    }
  }
  const double dn3dw = div(-n3, const_recip(P.omgrf, P.inv_omgrf));  // :72
  const double dn12dw = P.m2_over_omgrf * sq(n1);    // :73
  double dadw[NS], dgdw[NS];
#pragma unroll
  for (int is = 0; is < NS; is++) {
    dadw[is] = P.m2_over_omgrf * alpha[is];  // :74
    dgdw[is] = P.m1_over_omgrf * gamma[is];  // :75
  }
  double sa = 0.;
#pragma unroll
  for (int is = 0; is < NS; is++) sa += alpha[is];
  const double p = 1. - sa;  // :78
  double t = 1.;
#pragma unroll
  for (int is = 0; is < NS; is++) t *= (1. - sq(gamma[is]));  // :79
  double dq1da[NS], dq2da[NS];
#pragma unroll
  for (int is1 = 0; is1 < NS; is1++) {  // :83-91
    dq1da[is1] = 1.;
    dq2da[is1] = 1.;
#pragma unroll
    for (int is = 0; is < NS; is++)
      if (is != is1) {
        dq1da[is1] = dq1da[is1] * (1. + gamma[is]);
        dq2da[is1] = dq2da[is1] * (1. - gamma[is]);
      }
  }
  double q1 = 0., q2 = 0., su = 0.;
#pragma unroll
  for (int is = 0; is < NS; is++) q1 += alpha[is] * dq1da[is];  // :94
#pragma unroll
  for (int is = 0; is < NS; is++) q2 += alpha[is] * dq2da[is];  // :95
#pragma unroll
  for (int is = 0; is < NS; is++) su += alpha[is] * dq1da[is] * dq2da[is];
  const double u = t - su;               // :98
  const double q = 2. * u - t + q1 * q2; // :101
  const double n3_2 = sq(n3), n3_4 = pow4(n3), n1_2 = sq(n1), n1_4 = pow4(n1);
  double duda[NS], dqda[NS], ddda[NS];
#pragma unroll
  for (int is = 0; is < NS; is++) {  // :104-112
    duda[is] = -dq1da[is] * dq2da[is];
    dqda[is] = 2. * duda[is] + dq1da[is] * q2 + q1 * dq2da[is];
    ddda[is] = -t * n3_4 + (2. * (u - p * duda[is]) + (-t + duda[is]) * n1_2) * n3_2 - q +
               p * dqda[is] - (dqda[is] - u + p * duda[is]) * n1_2 + duda[is] * n1_4;
  }
  double gp[NS][NS], gm[NS][NS], gpm[NS][NS];
#pragma unroll
  for (int is1 = 0; is1 < NS; is1++)
#pragma unroll
    for (int is2 = 0; is2 < NS; is2++) {  // :116-125
      double a = 1., b = 1.;
#pragma unroll
      for (int is = 0; is < NS; is++)
        if (is != is1 && is != is2) {
          a = a * (1. + gamma[is]);
          b = b * (1. - gamma[is]);
        }
      gp[is1][is2] = a;
      gm[is1][is2] = b;
      gpm[is1][is2] = a * b;
    }
  double dtdg[NS], dudg[NS], dq1dg[NS], dq2dg[NS], dqdg[NS], dddg[NS];
#pragma unroll
  for (int is = 0; is < NS; is++) dtdg[is] = 2. * gamma[is] * duda[is];  // :128
#pragma unroll
  for (int is = 0; is < NS; is++) {
    double a = 0., b = 0., c = 0.;
#pragma unroll
    for (int j = 0; j < NS; j++) {
      a += alpha[j] * gpm[j][is];  // :132
      b += alpha[j] * gp[j][is];   // :138
      c += alpha[j] * gm[j][is];   // :144
    }
    dudg[is] = dtdg[is] + 2. * gamma[is] * (a + alpha[is] * duda[is]);  // :134
    dq1dg[is] = b - alpha[is] * dq1da[is];                              // :140
    dq2dg[is] = -c + alpha[is] * dq2da[is];                             // :146
    dqdg[is] = 2. * dudg[is] - dtdg[is] + dq1dg[is] * q2 + q1 * dq2dg[is];  // :149
    dddg[is] = dtdg[is] * p * n3_4 + (-2. * p * dudg[is] + (dtdg[is] * p + dudg[is]) * n1_2) * n3_2 +
               p * dqdg[is] - (dqdg[is] + p * dudg[is]) * n1_2 + dudg[is] * n1_4;  // :152-154
  }
  const double dddn3 = (4. * t * p * n3_2 + 2. * (-2. * p * u + (t * p + u) * n1_2)) * n3;  // :157
  const double dddn12 = (t * p + u) * n3_2 - (q + p * u) + 2. * u * n1_2;                   // :158
#pragma unroll
  for (int i = 0; i < 3; i++) {
    dddk[i] = dddn3 * dn3dk[i] + dddn12 * dn12dk[i];  // :162
    double a = 0.;
#pragma unroll
    for (int is = 0; is < NS; is++) a += ddda[is] * dadx[i][is] + dddg[is] * dgdx[i][is];  // :166
    dddx[i] = a + dddn3 * dn3dx[i] + dddn12 * dn12dx[i];                                   // :168
  }
  double a = 0.;
#pragma unroll
  for (int is = 0; is < NS; is++) a += ddda[is] * dadw[is] + dddg[is] * dgdw[is];
  dddw = a + dddn3 * dn3dw + dddn12 * dn12dw;  // :171
}

// Cold dielectric tensor in the Stix frame: the five distinct entries of eps_h
// (suscep_m.f90:53-86, 142-176) as real numbers.  eps_h(1,1)=eps_h(2,2)=e11 (real),
// eps_h(3,3)=e33 (real), eps_h(1,2) = i*x12 = -eps_h(2,1) (pure imaginary), rest zero.
// The reference forms these with complex arithmetic; every dropped term is an exact zero, and
// chi(1,2) goes through __divdc3 (see divdc3_real), so the values are bit-identical.
template <int NS>
RAYS_DEV void eps_cold(const double alpha[NS], const double gamma[NS], double& e11, double& e33,
                       double& x12) {
  double s11 = 0., s33 = 0., s12 = 0.;
#pragma unroll
  for (int is = 0; is < NS; is++) {
    const double omg2 = 1. - sq(gamma[is]);
    s11 = s11 + fdiv(-alpha[is], omg2);                         // suscep_m.f90:72
    s33 = s33 + (-alpha[is]);                                  // :74
    s12 = s12 + (-divdc3_real(alpha[is] * gamma[is], omg2));   // :75
  }
  e11 = s11 + 1.0;
  e33 = s33 + 1.0;
  x12 = s12;
}

// det(eps_h + nn - n^2 I) for n = (n1, 0, n3): deriv_num.f90:126-134 == check_save.f90:209-218
RAYS_DEV double epsn_det(double e11, double e33, double x12, double n1, double n3, double nsq) {
  const double E11 = e11 + n1 * n1 - nsq;
  const double E22 = e11 + 0. * 0. - nsq;  // n(2)*n(2) = 0
  const double E33 = e33 + n3 * n3 - nsq;
  const double E13 = 0. + n1 * n3 - 0. * nsq;
  // ctmp%re = E33*(E11*E22 - (i x12)(-i x12)... ) - 0 + E31*(0 - E22*E13)
  return E33 * (E11 * E22 - x12 * x12) - E13 * (E22 * E13);
}

// determ                    deriv_num.f90:99-153 (ray_dispersion_model == 'cold'), in two parts: what depends on the
// plasma parameters alone (the dielectric tensor and the product of :146) and what depends on the wave vector.
// deriv_num's six wave-vector differences share the first part (same alpha, gamma: same bits).
struct DetermEps {
  double e11, e33, x12, pr;
};
template <int NS>
RAYS_DEV DetermEps determ_eps(const double alpha[NS], const double gamma[NS]) {
  DetermEps E;
  eps_cold<NS>(alpha, gamma, E.e11, E.e33, E.x12);
  double pr = 1.;
#pragma unroll
  for (int is = 0; is < NS; is++) pr *= (1. - sq(gamma[is]));  // :146 (unused species contribute 1)
  E.pr = pr;
  return E;
}
RAYS_DEV double determ_k(const double bunit[3], const DetermEps& E, const double kvec[3], const Recip& Rk0) {
  const double k3 = kvec[0] * bunit[0] + kvec[1] * bunit[1] + kvec[2] * bunit[2];
  const double k1 = fsqrt(sq(kvec[0] - k3 * bunit[0]) + sq(kvec[1] - k3 * bunit[1]) + sq(kvec[2] - k3 * bunit[2]));
  const double n1 = div(k1, Rk0), n3 = div(k3, Rk0);
  const double nsq = sq(n1) + 0. + sq(n3);
  const double det = epsn_det(E.e11, E.e33, E.x12, n1, n3, nsq);
  return det * E.pr;
}
template <int NS>
RAYS_DEV double determ(const double bunit[3], const double alpha[NS], const double gamma[NS],
                       const double kvec[3], const Recip& Rk0) {
  return determ_k(bunit, determ_eps<NS>(alpha, gamma), kvec, Rk0);
}

// Light equilibrium for determ: only bunit, alpha, gamma at a (possibly perturbed) point.
template <int EQ, int NS>
RAYS_DEV void eq_for_determ(const DevParams& P, const Recip& Romgrf, const Recip& Romgrf2,
                            const double rvec[3], double bunit[3], double alpha[NS], double gamma[NS]) {
  EqPoint<NS> e;
  equilibrium<EQ, NS>(P, Romgrf, Romgrf2, rvec, e, false);
#pragma unroll
  for (int i = 0; i < 3; i++) bunit[i] = e.bunit[i];
#pragma unroll
  for (int is = 0; is < NS; is++) {
    alpha[is] = e.alpha[is];
    gamma[is] = e.gamma[is];
  }
}

// deriv_num                 deriv_num.f90:1-155
// delta = 1.e-6 (single-precision literal, :37).  Perturbed positions are evaluated without the
// box test (the reference reads an undefined eq_point there).  Loops over the three axes are kept
// rolled (selects instead of dynamic register indexing) to bound code size.
template <int EQ, int NS>
RAYS_DEV void deriv_num(const DevParams& P, const EqPoint<NS>& eq0, const double rvec0[3],
                        const double kvec0[3], double dddx[3], double dddk[3], double& dddw) {
  double bu[3], al[NS], ga[NS];
  const Recip Ro = const_recip(P.omgrf, P.inv_omgrf), Ro2 = const_recip(P.omgrf2, P.inv_omgrf2);
  const Recip Rk0 = const_recip(P.k0, P.inv_k0);
#pragma unroll 1
  for (int i = 0; i < 3; i++) {  // :40-57
    double rp[3], rm[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      rp[c] = (c == i) ? rvec0[c] + P.delta : rvec0[c];
      rm[c] = (c == i) ? rvec0[c] - P.delta : rvec0[c];
    }
    eq_for_determ<EQ, NS>(P, Ro, Ro2, rp, bu, al, ga);
    const double det_plus = determ<NS>(bu, al, ga, kvec0, Rk0);
    eq_for_determ<EQ, NS>(P, Ro, Ro2, rm, bu, al, ga);
    const double det_minus = determ<NS>(bu, al, ga, kvec0, Rk0);
    const double d = (det_plus - det_minus) / P.two_delta;
    if (i == 0) dddx[0] = d;
    if (i == 1) dddx[1] = d;
    if (i == 2) dddx[2] = d;
  }
  const DetermEps E0 = determ_eps<NS>(eq0.alpha, eq0.gamma);  // the same tensor in all six wave-vector differences
#pragma unroll 1
  for (int i = 0; i < 3; i++) {  // :60-68
    const double ki = (i == 0) ? kvec0[0] : (i == 1) ? kvec0[1] : kvec0[2];
    const double change = fmax(P.delta, fabs(P.delta * ki)) / 2.;
    double kp[3], km[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      kp[c] = (c == i) ? kvec0[c] + change : kvec0[c];
      km[c] = (c == i) ? kvec0[c] - change : kvec0[c];
    }
    const double det_plus = determ_k(eq0.bunit, E0, kp, Rk0);
    const double det_minus = determ_k(eq0.bunit, E0, km, Rk0);
    const double d = (det_plus - det_minus) / (2. * change);
    if (i == 0) dddk[0] = d;
    if (i == 1) dddk[1] = d;
    if (i == 2) dddk[2] = d;
  }
  // :71-80 omega: per-lane omgrf/k0 instead of rewriting module variables.  The equilibrium at the same point with
  // another omgrf has the same fields; only alpha = omgp2 / omgrf**2 and gamma = omgc / omgrf change
  // (equilibrium_m.f90:262-265): the same two quotients from the unperturbed point's omgp2, omgc -- same bits,
  // without evaluating the fields twice more.
  (void)bu;
#pragma unroll
  for (int is = 0; is < NS; is++) {
    al[is] = div(eq0.omgp2[is], const_recip(P.omgrf2_p, P.inv_omgrf2_p));
    ga[is] = div(eq0.omgc[is], const_recip(P.omgrf_p, P.inv_omgrf_p));
  }
  const double det_plus = determ<NS>(eq0.bunit, al, ga, kvec0, const_recip(P.k0_p, P.inv_k0_p));
#pragma unroll
  for (int is = 0; is < NS; is++) {
    al[is] = div(eq0.omgp2[is], const_recip(P.omgrf2_m, P.inv_omgrf2_m));
    ga[is] = div(eq0.omgc[is], const_recip(P.omgrf_m, P.inv_omgrf_m));
  }
  const double det_minus = determ<NS>(eq0.bunit, al, ga, kvec0, const_recip(P.k0_m, P.inv_k0_m));
  dddw = (det_plus - det_minus) / P.omgrf0_delta;
}

// ---------------------------------------------------------------------------------------------
// Damping: damp_fund_ECH (damp_fund_ECH.f90:2-128) with the splined Z function of a real argument
// (zfunctions_m.f90:351-432; cspevx/cspevfn on the uniform grid, cspeval.f90:138-146,239).
// D_WARM and DELTA are default COMPLEX (single precision) in the reference (:36): both
// assignments truncate to float, reproduced here.  Complex quotients follow __divdc3.
// ---------------------------------------------------------------------------------------------
// real**integer as flang lowers it (llvm.powi -> compiler-rt __powidf2: square and multiply)
RAYS_DEV double powi_rt(double a, int b) {
  const bool recip = b < 0;
  double r = 1.;
  while (true) {
    if (b & 1) r = r * a;
    b /= 2;
    if (b == 0) break;
    a = a * a;
  }
  return recip ? 1. / r : r;
}

struct Cplx {
  double re, im;
};
RAYS_DEV Cplx divdc3(Cplx x, Cplx y) {  // compiler-rt __divdc3, finite operands
  double c = y.re, d = y.im;
#if defined(RAYS_TOL_FLAVOUR) && !defined(RAYS_HOST_EMUL)
  {  // tolerance flavour: the textbook quotient without the logb scaling (operands are O(1) here)
    const Recip Rd = make_recip(c * c + d * d);
    return Cplx{div(x.re * c + x.im * d, Rd), div(x.im * c - x.re * d, Rd)};
  }
#endif
  const double m = fmax(fabs(c), fabs(d));
  int k = 0;
  if (m > 0.0 && m < __builtin_inf()) {
    k = ilogb(m);
    c = scalbn(c, -k);
    d = scalbn(d, -k);
  }
  const double denom = c * c + d * d;
  Cplx r;
  r.re = scalbn((x.re * c + x.im * d) / denom, -k);
  r.im = scalbn((x.im * c - x.re * d) / denom, -k);
  return r;
}

// x_grid(i), 1-based (zfunctions_m.f90:444); Rn = shared reciprocal of nx - 1 (six quotients per lookup)
RAYS_DEV double zf_x(const DevParams& P, int i, const Recip& Rn) {
  RAYS_FP_AS_WRITTEN
  return P.zf_xmin + div((double)(i - 1) * (P.zf_xmax - P.zf_xmin), Rn);
}

RAYS_DEV Cplx zfun_real_arg_spline(const DevParams& P, double z) {
  RAYS_FP_AS_WRITTEN
  double re;
  if (fabs(z) <= 10.0) {  // spline_range
    const int nxm = P.zf_nx - 1;
    const Recip Rn = make_recip((double)nxm);
    const double x1 = zf_x(P, 1, Rn), xn = zf_x(P, P.zf_nx, Rn);
    const double t = 1 + fdiv(nxm * (z - x1), xn - x1);
    int i = (int)t;  // NaN -> 0 on the device (undefined in Fortran); clamped below
    i = i < nxm ? i : nxm;
    i = i > 1 ? i : 1;
    if (z < zf_x(P, i, Rn)) i = i - 1;
    else if (z > zf_x(P, i + 1, Rn)) i = i + 1;
    i = i < 1 ? 1 : (i > nxm ? nxm : i);
    const double dx = z - zf_x(P, i, Rn);
    double f0, f1, f2, f3;
#ifndef RAYS_HOST_EMUL
    if (P.zf_lds) {  // wave-uniform
      const eq_lds_ptr f = (eq_lds_ptr)(unsigned long long)P.zf_lds + 4 * (i - 1);
      f0 = f[0]; f1 = f[1]; f2 = f[2]; f3 = f[3];
    } else
#endif
    {
      const double* f = P.zf_fspl + 4 * (long long)(i - 1);
      f0 = f[0]; f1 = f[1]; f2 = f[2]; f3 = f[3];
    }
    re = f0 + dx * (f1 + dx * (f2 + dx * f3));
  } else {  // asymptotic expansion :408-414 (unreachable from damp_fund_ECH: |xi| <= 5)
    const double A[6] = {1., 1. / 2., 3. / 4., 15. / 8., 105. / 16., 945. / 32.};
    const double z_inv = 1.0 / z;
    re = 0.;
    for (int i = 1; i <= 6; i++) re = re - powi_rt(z_inv, 2 * i - 1) * A[i - 1];
  }
  Cplx r;
  r.re = re;
  r.im = 1.7724538509055159 * libm::exp(-(z * z));  // sqrt(pi)
  return r;
}

template <int NS>
RAYS_DEV double damp_fund_ech(const DevParams& P, const EqPoint<NS>& eq, const double kvec[3],
                              const double vg[3]) {
  const double k0 = P.k0;
  const Recip Rk0 = const_recip(P.k0, P.inv_k0);
  const double nvec[3] = {div(kvec[0], Rk0), div(kvec[1], Rk0), div(kvec[2], Rk0)};
  const double k3 = kvec[0] * eq.bunit[0] + kvec[1] * eq.bunit[1] + kvec[2] * eq.bunit[2];
  const double k1 = fsqrt(sq(kvec[0] - k3 * eq.bunit[0]) + sq(kvec[1] - k3 * eq.bunit[1]) +
                         sq(kvec[2] - k3 * eq.bunit[2]));
  const double R3 = div(k3, Rk0), R1 = div(k1, Rk0);
  const double R1S = sq(R1), R3S = sq(R3), RS = R1S + R3S;
  const double B1 = eq.gamma[0], BETAE = sq(B1);
  if (R3 == 0.) return 0.;  // :59
  const double vth = fsqrt(fdiv(2. * eq.ts0, P.ms[0]));
  const double VT = fdiv(vth, P.clight);
  const double xi = fdiv(P.omgrf + eq.omgc0, k3 * vth);
  if (fabs(xi) > 5.) return 0.;  // :73
  // zfun0_real_arg (:351-372).  The reference `stop 1`s the whole program when kz is neither
  // > 0 nor < 0, i.e. when the state is already NaN (a ray that left a parabolic-density plasma);
  // here the NaN simply propagates and the ray ends at check_save ('infinite_Vg') -- DESIGN.md.
  Cplx zf;
  if (k3 > 0.) {
    zf = zfun_real_arg_spline(P, xi);
  } else {
    zf = zfun_real_arg_spline(P, -xi);
    zf.re = -zf.re;
    zf.im = -zf.im;
  }
  const double Pa = eq.alpha[0];
  const double Q = fdiv(Pa / 2., 1 - B1);
  const double L1 = (1. - Q) * RS * R1S + (1. - Pa) * RS * R3S - (1. - Q) * (1. - Pa) * (RS + R3S) -
                    (1 - 2. * Q) * R1S + (1 - 2 * Q) * (1 - Pa);
  const double L2 = -(fdiv(Pa, B1) * (RS * R1S - (1. - 2. * Q) * R1S)) +
                    fdiv(fdiv(Pa * Pa / 4., BETAE) * R1S, R3S) * (RS + R3S - 2. * (1. - 2. * Q));
  const double L5 = Pa * (RS * R3S - (1. - Q) * (RS + R3S) + (1. - 2. * Q));
  const double F = (1. - B1) * R3 * VT * (L1 + L2 + fdiv(fdiv(R1S / 2., R3), BETAE) * VT * xi * L5);
  const Cplx one = {1., 0.};
  const Cplx zinv = divdc3(one, zf);
  const double par_re = xi + zinv.re, par_im = 0. + zinv.im;
  // -(F,0)*(par): (F*a - 0*b, 0*a + F*b), negated, truncated to COMPLEX(4)
  const float dw_re = (float)(-(F * par_re - 0. * par_im));
  const float dw_im = (float)(-(0. * par_re + F * par_im));
  const double A = 1. - Pa - BETAE;
  const double B = -((1. - Pa) * A + sq(1. - Pa) - BETAE) + (A + (1. - Pa) * (1. - BETAE)) * R3S;
  const double DDNX2 = 2. * A * R1S + B;
  const double DDNZ = 2. * R3 * ((A + (1. - Pa) * (1. - BETAE)) * R1S + (1 - Pa) * (2. * (1. - BETAE) * R3S - 2. * A));
  const double nvg = fsqrt(sq(vg[0]) + sq(vg[1]) + sq(vg[2]));
  double dot = 0.;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double ddn = DDNX2 * (2 * (nvec[i] - R3 * eq.bunit[i])) + DDNZ * eq.bunit[i];
    dot += ddn * fdiv(vg[i], nvg);
  }
  const Cplx num = {-(double)dw_re, -(double)dw_im};
  const Cplx den = {dot, 0.};
  const float delta_im = (float)divdc3(num, den).im;  // DELTA is COMPLEX(4)
  return k0 * (double)delta_im;                        // ksi(0) = k0*aimag(DELTA)
}

// ---------------------------------------------------------------------------------------------
// eqn_ray tail               eqn_ray.f90:131-229: group velocity + ray equations from dD/d(x,k,w).
// Returns a stop code (0 = ok).  NV = 7 (+5 with integrate_eq_gradients).
// ---------------------------------------------------------------------------------------------
template <int NS, int NV, bool MULTI = false>
RAYS_DEV int ray_equations(const DevParams& P, const EqPoint<NS>& eq, const double kvec[3], double v7,
                           const double dddx[3], const double dddk[3], double dddw, double dvds[NV]) {
  if (!(dddw != 0.)) return RAYS_STOP_INFINITE_VG_RHS;  // :133 (`/= 0.` is true for NaN)
  const Recip Rw = make_recip(dddw);
  double vg[3];
#pragma unroll
  for (int i = 0; i < 3; i++) vg[i] = div(-dddk[i], Rw);
  const double vg0 = fsqrt(sq(vg[0]) + sq(vg[1]) + sq(vg[2]));
  double dsd;
  if (P.ray_param == RAYS_PARAM_ARCL) {  // :150-170
    if (dddk[0] != 0. || dddk[1] != 0. || dddk[2] != 0.) {
      const double sgn = copysign(1.0, dddw);
      const Recip Rnk = make_recip(fsqrt(sq(dddk[0]) + sq(dddk[1]) + sq(dddk[2])));
#pragma unroll
      for (int i = 0; i < 3; i++) {
        dvds[i] = div(-sgn * dddk[i], Rnk);
        dvds[3 + i] = div(sgn * dddx[i], Rnk);
      }
      dsd = 1.;
    } else {
      return RAYS_STOP_RAY_STALLED;
    }
  } else {  // 'time' :172-181
#pragma unroll
    for (int i = 0; i < 3; i++) {
      dvds[i] = vg[i];  // -dddk/dddw
      dvds[3 + i] = div(dddx[i], Rw);
    }
    dsd = vg0;
  }
  dvds[6] = dsd;  // :190
  typedef RayVec<MULTI, NS, NV> L;
  constexpr bool DAMP = L::DAMP;
  constexpr int NV0 = L::NV0;
  if (DAMP) {  // :196-204
    const double ki = damp_fund_ech<NS>(P, eq, kvec, vg);
    dvds[7] = dsd * 2. * ki * (1. - v7);
    if (MULTI) {  // :207-212: ksi(0) = ki, ksi(1:nspec) = 0 (damp_fund_ECH.f90:122-124)
#pragma unroll
      for (int is = 0; is < NS; is++) dvds[8 + is] = dsd * 2. * (is == 0 ? ki : 0.) * (1. - v7);
    }
  }
  if (L::GRAD) {  // :217-229 integrate_eq_gradients
    const Recip Rvg0 = make_recip(vg0);
    double vu[3];
#pragma unroll
    for (int i = 0; i < 3; i++) vu[i] = div(vg[i], Rvg0);
#pragma unroll
    for (int j = 0; j < 3; j++)
      dvds[NV0 + j] = dsd * vu[0] * eq.gbt[0][j] + dsd * vu[1] * eq.gbt[1][j] + dsd * vu[2] * eq.gbt[2][j];
    dvds[NV0 + 3] = dsd * vu[0] * eq.gradns[0][0] + dsd * vu[1] * eq.gradns[0][1] + dsd * vu[2] * eq.gradns[0][2];
    dvds[NV0 + 4] = dsd * vu[0] * eq.gradts0[0] + dsd * vu[1] * eq.gradts0[1] + dsd * vu[2] * eq.gradts0[2];
  }
  return 0;
}

// eqn_ray                   eqn_ray.f90:1-236
template <int EQ, int NS, int DERIV, int NV>
RAYS_DEV int eqn_ray(const DevParams& P, const double v[NV], double dvds[NV]) {
  const double rvec[3] = {v[0], v[1], v[2]}, kvec[3] = {v[3], v[4], v[5]};
  EqPoint<NS> eq;
  equilibrium<EQ, NS>(P, const_recip(P.omgrf, P.inv_omgrf), const_recip(P.omgrf2, P.inv_omgrf2), rvec, eq, true);
  if (eq.err) return eq.err;  // :90-102
  double dddx[3], dddk[3], dddw;
  if (DERIV == RAYS_DERIV_COLD) {
    const Recip Rk0 = const_recip(P.k0, P.inv_k0);
    const double nvec[3] = {div(kvec[0], Rk0), div(kvec[1], Rk0), div(kvec[2], Rk0)};  // :84
    deriv_cold<NS>(P, eq, nvec, dddx, dddk, dddw);
  } else {
    deriv_num<EQ, NS>(P, eq, rvec, kvec, dddx, dddk, dddw);
  }
  constexpr bool MULTI = (EQ & kEqMultiSpec) != 0;
  return ray_equations<NS, NV, MULTI>(P, eq, kvec, v[RayVec<MULTI, NS, NV>::DAMP ? 7 : 0], dddx, dddk, dddw, dvds);
}

// ---------------------------------------------------------------------------------------------
// rhs_eval: ONE evaluation of the ray-equation right-hand side at state v, optionally fused with
// check_save (check_save.f90:1-237) at the same state.
//
// check_save evaluates equilibrium(v) and deriv_cold(eq, v/k0); the next ODE step's first eqn_ray
// call evaluates the same pure functions at the same v.  Evaluating them once is bit-identical
// and removes one of the five RHS evaluations per RK4 step.  Keeping a single inlined copy of the
// RHS (instead of eqn_ray + check_save) also bounds the kernel's code size.
//   do_check : this evaluation is also a check_save call (per-lane flag; wave-uniform in practice)
//   resid    : normalised dispersion residual (check_save.f90:163-235)      [do_check only]
//   cs_flag  : flag check_save latches; cs_stop : stop_ode set by check_save [do_check only]
//   code     : stop code eqn_ray(v) returns (0 = ok), f = its dvds
// ---------------------------------------------------------------------------------------------
template <int EQ, int NS, int DERIV, int NV>
RAYS_DEV void rhs_eval(const DevParams& P, const double v[NV], bool do_check, double& resid,
                       int& cs_flag, bool& cs_stop, int& code, double f[NV]) {
  const double rvec[3] = {v[0], v[1], v[2]}, kvec[3] = {v[3], v[4], v[5]};
  EqPoint<NS> eq;
  equilibrium<EQ, NS>(P, const_recip(P.omgrf, P.inv_omgrf), const_recip(P.omgrf2, P.inv_omgrf2), rvec, eq, true);
  const Recip Rk0 = const_recip(P.k0, P.inv_k0);
  const double nvec[3] = {div(kvec[0], Rk0), div(kvec[1], Rk0), div(kvec[2], Rk0)};  // eqn_ray.f90:84
  cs_flag = 0;
  cs_stop = false;
  resid = 0.;
  if (do_check) {
    cs_flag = eq.err;  // check_save.f90:41-43: flag text only
    // :53-57
    const double k3 = kvec[0] * eq.bunit[0] + kvec[1] * eq.bunit[1] + kvec[2] * eq.bunit[2];
    const double k1 = fsqrt(sq(kvec[0] - k3 * eq.bunit[0]) + sq(kvec[1] - k3 * eq.bunit[1]) +
                           sq(kvec[2] - k3 * eq.bunit[2]));
    // residual :163-235
    const double n1 = div(k1, Rk0), n3 = div(k3, Rk0);
    const double nsq = sq(n1) + 0. + sq(n3);
    double e11, e33, x12;
    eps_cold<NS>(eq.alpha, eq.gamma, e11, e33, x12);
    const double det = epsn_det(e11, e33, x12, n1, n3, nsq);
    const double N11 = fabs(e11) + fabs(n1 * n1), N22 = fabs(e11) + fabs(0. * 0.);
    const double N33 = fabs(e33) + fabs(n3 * n3), N12 = fabs(x12) + fabs(n1 * 0.);
    const double N13 = 0. + fabs(n1 * n3), N23 = 0. + fabs(0. * n3);
    const double den = N33 * (N11 * N22) + N33 * (N12 * N12) + N23 * (N11 * N23) + N23 * (N12 * N13) +
                       N13 * (N12 * N23) + N13 * (N22 * N13);
    resid = divdc3_real(fabs(det), den);  // eps_norm is complex(rkind) -> __divdc3
    if (resid > P.resid_limit) {          // :68-71
      cs_stop = true;
      cs_flag = RAYS_STOP_DISP_RESIDUAL;
    }
  }
  double dddx[3], dddk[3], dddw;
  if (DERIV == RAYS_DERIV_COLD || do_check) {
    deriv_cold<NS>(P, eq, nvec, dddx, dddk, dddw);  // eqn_ray.f90:111 / check_save.f90:82
    if (do_check && !(fabs(dddw) > 2.2250738585072014e-308)) {  // check_save.f90:90 tiny(dddw)
      cs_stop = true;
      cs_flag = RAYS_STOP_INFINITE_VG_CHECK;  // :107-108
    }
  }
  constexpr bool MULTI = (EQ & kEqMultiSpec) != 0;
  constexpr bool DAMP = RayVec<MULTI, NS, NV>::DAMP;
  if (do_check && DAMP) {  // check_save.f90:114-125
    if (v[DAMP ? 7 : 0] > P.total_damping_limit) {
      cs_stop = true;
      cs_flag = RAYS_STOP_TOTAL_ABSORPTION;
    }
  }
  if (DERIV == RAYS_DERIV_NUM) deriv_num<EQ, NS>(P, eq, rvec, kvec, dddx, dddk, dddw);
  const int rc = ray_equations<NS, NV, MULTI>(P, eq, kvec, v[DAMP ? 7 : 0], dddx, dddk, dddw, f);
  code = eq.err ? eq.err : rc;  // eqn_ray.f90:90-102 returns before the derivatives
}

// initialize_ode_vector     initialize_ode_vector.f90:25-54
template <int EQ, int NS, int NV>
RAYS_DEV void initialize_ode_vector(const DevParams& P, const double* __restrict__ r0,
                                    const double* __restrict__ n0, double v[NV]) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    v[i] = r0[i];
    v[3 + i] = P.k0 * n0[i];
  }
  v[6] = 0.;
  typedef RayVec<(EQ & kEqMultiSpec) != 0, NS, NV> L;
  constexpr int NV0 = L::NV0;
#pragma unroll
  for (int i = 7; i < NV0; i++) v[i] = 0.;  // total absorbed power (+ one row per species)
  if (L::GRAD) {
    EqPoint<NS> eq;
    const double rvec[3] = {v[0], v[1], v[2]};
    equilibrium<EQ, NS>(P, const_recip(P.omgrf, P.inv_omgrf), const_recip(P.omgrf2, P.inv_omgrf2), rvec, eq, false);
    v[NV0] = eq.bvec[0];
    v[NV0 + 1] = eq.bvec[1];
    v[NV0 + 2] = eq.bvec[2];
    v[NV0 + 3] = eq.ns[0];
    v[NV0 + 4] = eq.ts0;
  }
}

}  // namespace rays
