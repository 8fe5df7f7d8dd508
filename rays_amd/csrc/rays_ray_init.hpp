// rays_ray_init.hpp -- ray initialisation on the device: the step right
// before the hot path.
//
// Reference path restated here:
//   ray_init_solovev_nphi_ntheta            solovev_ray_init_nphi_ntheta_m.f90:60-198
//   ray_init_axisym_toroid_R_Z_nphi_ntheta  axisym_toroid_ray_init_R_Z_nphi_ntheta_m.f90:67-244
//   simple_slab_ray_init                    simple_slab_ray_init_m.f90:59-187
//   solve_n1_vs_n2_n3                       dispersion_solvers_m.f90:49-112
//   solve_cold_n1sq_vs_n3                   disp_solve_cold_n1sq_vs_n3.f90:1-90
//   RLSDP_cold                              suscep_m.f90:180-219
//
// The reference runs a serial loop over the fan with two equilibrium calls and a quadratic solve
// per ray, and drops evanescent launches, so surviving rays are numbered in loop order.  Here one
// thread evaluates one fan member (the launch-point equilibrium is re-evaluated per thread, as the
// reference does per ray: same device functions as the trace kernels, bit-identical B, alpha,
// gamma), and the survivors are compacted in loop order with a prefix sum.  Launch positions
// (r0 + dr*i, cos/sin of the launch angle) are computed by the host caller with the host libm, the
// same functions the reference uses.
#pragma once

#include "rays_device.hpp"

namespace rays {

enum { WAVE_PLUS = 0, WAVE_MINUS = 1, WAVE_FAST = 2, WAVE_SLOW = 3 };

struct FanArgs {
  int model;              // RAYS_RAY_INIT_*
  int wave_mode, k0_sign;
  int n_launch;           // launch positions
  int n_a, n_b;           // fan: n_rindex_theta x n_rindex_phi, or n_ky x n_kz (slab)
  double a0, da, b0, db;  // rindex_theta0/delta, rindex_phi0/delta  |  rindex_y0/delta, rindex_z0/delta
  const double* launch;   // [n_launch][3]
};

// Re[(a + 0i)/(c + 0i)] as compiler-rt's __divdc3 forms it: (a*c + 0*0)/(c*c + 0*0) after logb
// scaling; the `+ 0.` terms turn a -0 product into +0, as in the reference binary.
RAYS_DEV double divdc3_re(double a, double c) {
  const double fc = fabs(c);
  int k = 0;
  if (fc > 0.0 && fc < __builtin_inf()) {
    k = ilogb(fc);
    c = scalbn(c, -k);
  }
  const double denom = c * c + 0.0;
  return scalbn((a * c + 0.0) / denom, -k);
}

// R, L, S, D, P of the cold plasma (suscep_m.f90:180-219)
template <int NS>
RAYS_DEV void rlsdp_cold(const double alpha[NS], const double gamma[NS], double& S, double& D, double& Pp,
                         double& Rr, double& L) {
  double R = 0., Ll = 0., Pq = 0.;
#pragma unroll
  for (int is = 0; is < NS; is++) {
    R = R - alpha[is] / (1.0 + gamma[is]);
    Ll = Ll - alpha[is] / (1.0 - gamma[is]);
    Pq = Pq - alpha[is];
  }
  R = 1.0 + R;
  Ll = 1.0 + Ll;
  Pq = 1.0 + Pq;
  S = (R + Ll) / 2.0;
  D = (R - Ll) / 2.0;
  Pp = Pq;
  Rr = R;
  L = Ll;
}

// Real n1 of the requested mode, or evanescent (returns false).  The reference works in complex
// arithmetic; for a real discriminant >= 0 every imaginary part is an exact zero and the complex
// quotients are compiler-rt __divdc3 with zero imaginary parts (divdc3_re).
template <int NS>
RAYS_DEV bool solve_n1_vs_n2_n3(const double alpha[NS], const double gamma[NS], int wave_mode, int k_sign,
                                double n2, double n3, double& n1) {
  double S, D, Pp, R, L;
  rlsdp_cold<NS>(alpha, gamma, S, D, Pp, R, L);
  const double n3sq = n3 * n3;
  const double a = S;
  const double b = -R * L - Pp * S + n3sq * (Pp + S);
  const double c = Pp * (n3sq - R) * (n3sq - L);
  const double discr = b * b - 4.0 * a * c;
  if (!(discr >= 0.0)) return false;
  const double sd = sqrt(discr);
  const bool neg = copysign(1.0, b) < 0.0;
  // sgn(b) < 0: plus = (-b + sd)/(2a), minus = 2c/(-b + sd); else minus = (-b - sd)/(2a), plus = 2c/(-b - sd)
  const double t = neg ? -b + sd : -b - sd;
  const double big = divdc3_re(t, 2.0 * a);
  const double small = divdc3_re(2.0 * c, t);
  const double plus = neg ? big : small, minus = neg ? small : big;
  const bool pf = fabs(plus) <= fabs(minus);
  const double fast = pf ? plus : minus, slow = pf ? minus : plus;
  const double nperp_sq = wave_mode == WAVE_PLUS ? plus : wave_mode == WAVE_MINUS ? minus
                          : wave_mode == WAVE_FAST ? fast : slow;
  const double arg = nperp_sq - n2 * n2;
  if (!(arg >= 0.0)) return false;
  n1 = (double)k_sign * sqrt(fabs(arg));
  return true;
}

// grad(psi) at the launch point: solovev_psi (solovev_eq_m.f90:308-318) / axisym_toroid_psi
template <int EQ>
RAYS_DEV void launch_gradpsi(const DevParams& P, const double rvec[3], double g[3]) {
  const double x = rvec[0], y = rvec[1], z = rvec[2];
  const double r = sqrt(x * x + y * y);
  if ((EQ & 3) == RAYS_EQ_SOLOVEV || ((EQ & 3) == RAYS_EQ_AXISYM && P.a_mag_model == RAYS_AXI_MAG_SOLOVEV)) {
    // (axisym_toroid_psi -> solovev_magnetics_psi: the same statements, solovev_magnetics_m.f90:199-207)
    const Recip Rrk = const_recip(P.rk, P.inv_rk), Rrk2 = const_recip(P.rk2, P.inv_rk2);
    const Recip Rrmaj = const_recip(P.rmaj, P.inv_rmaj);
    const double br = div(-P.bp0 * r * z, Rrk2);
    const double bz = P.bp0 * (sq(div(z, Rrk)) + .5 * (sq(div(r, Rrmaj)) - 1.));
    g[0] = x * bz;
    g[1] = y * bz;
    g[2] = -r * br;
  } else if (P.a_mag_model == RAYS_AXI_MAG_EQDSK_LIN) {  // eqdsk_magnetics_lin_interp_psi (:216-249)
    double psi, PsiR, PsiZ;
    eqlin_psi_grad(P, r, z, psi, PsiR, PsiZ);
    const Recip Rr = make_recip(r);
    const double br = div(-PsiZ, Rr), bz = div(PsiR, Rr);
    g[0] = x * bz;
    g[1] = y * bz;
    g[2] = -r * br;
  } else {
    double f6[6];
    spl2_fpp(P, r, z, f6);
    const Recip Rr = make_recip(r);
    const double br = div(f6[2], Rr), bz = div(-f6[1], Rr);
    g[0] = -x * bz;
    g[1] = -y * bz;
    g[2] = r * br;
  }
}

// One fan member: launch position `rvec`, fan indices (ia, ib).  Returns false when the launch is
// dropped (equilibrium error or evanescent); else rindex0 = the initial refractive index vector.
template <int EQ, int NS>
RAYS_DEV bool fan_member(const DevParams& P, const FanArgs& F, const double rvec[3], int ia, int ib,
                         double rindex0[3]) {
  EqPoint<NS> eq;
  equilibrium<EQ, NS>(P, const_recip(P.omgrf, P.inv_omgrf), const_recip(P.omgrf2, P.inv_omgrf2), rvec, eq, true);
  if (eq.err) return false;
  if ((EQ & 3) == RAYS_EQ_SLAB) {  // simple_slab_ray_init_m.f90:131-165
    const double ny = F.a0 + (double)ia * F.da, nz = F.b0 + (double)ib * F.db;
    const double n2 = ny * eq.bunit[2] - nz * eq.bunit[1];
    const double n3 = ny * eq.bunit[1] + nz * eq.bunit[2];
    double nx;
    if (!solve_n1_vs_n2_n3<NS>(eq.alpha, eq.gamma, F.wave_mode, F.k0_sign, n2, n3, nx)) return false;
    rindex0[0] = nx;
    rindex0[1] = ny;
    rindex0[2] = nz;
    return true;
  }
  double g[3];
  launch_gradpsi<EQ>(P, rvec, g);
  const double gn = sqrt((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
  const double psi_unit[3] = {g[0] / gn, g[1] / gn, g[2] / gn};
  double th[3] = {-g[2], 0., g[0]};  // psi_unit X phi_unit
  const double tn = sqrt((th[0] * th[0] + 0.) + th[2] * th[2]);
  th[0] = th[0] / tn;
  th[1] = th[1] / tn;
  th[2] = th[2] / tn;
  const double* bu = eq.bunit;
  const double trans[3] = {bu[1] * psi_unit[2] - bu[2] * psi_unit[1], bu[2] * psi_unit[0] - bu[0] * psi_unit[2],
                           bu[0] * psi_unit[1] - bu[1] * psi_unit[0]};
  const double r_th = F.a0 + (double)ia * F.da, r_ph = F.b0 + (double)ib * F.db;
  // rindex_vec = rindex_phi*phi_unit + rindex_theta*theta_unit, phi_unit = (0,1,0)
  const double rv[3] = {r_ph * 0. + r_th * th[0], r_ph * 1. + r_th * th[1], r_ph * 0. + r_th * th[2]};
  const double n3 = (bu[0] * rv[0] + bu[1] * rv[1]) + bu[2] * rv[2];
  const double n2 = (trans[0] * rv[0] + trans[1] * rv[1]) + trans[2] * rv[2];
  double npsi;
  if (!solve_n1_vs_n2_n3<NS>(eq.alpha, eq.gamma, F.wave_mode, F.k0_sign, n2, n3, npsi)) return false;
#pragma unroll
  for (int i = 0; i < 3; i++) rindex0[i] = rv[i] - npsi * psi_unit[i];
  return true;
}

}  // namespace rays
