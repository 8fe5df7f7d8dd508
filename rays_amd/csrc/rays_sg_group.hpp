// rays_sg_group.hpp -- Shampine-Gordon ray-trace kernel with ONE RAY PER GROUP OF G LANES, for the finite-difference
// dispersion derivatives (ode_solver_name = 'SG_ODE', ray_deriv_name = 'numerical', nv = 7: BASELINE config 3).
//
// Reference path restated here: the same as rays_sg.hpp (trace_rays ray_tracing.f90:67-264, SG_ode SG_ode_m.f90:89-159,
// ode/de/step/intrp ode_RAYS.f90) + deriv_num (deriv_num.f90:1-155).
//
// Why another mapping.  With one ray per lane (rays_sg.hpp) an evaluation of the right-hand side is TWENTY
// sub-evaluations run one after the other in the lane -- deriv_num's fourteen determinants, six of them at perturbed
// equilibria, two at perturbed frequencies (deriv_num.f90:40-84) -- and the Adams integrator around it keeps
// phi(7,16) and six coefficient vectors per lane: 256 VGPRs + ~200 AGPRs + all of the LDS + a global workspace, one
// wave per SIMD, ~13 clocks per bookkeeping instruction (nothing hides an LDS round trip), and a 65536-ray fan is
// exactly one ray per resident lane: nothing to refill, the pass lasts as long as its longest ray while half the
// SIMDs idle.  Here a ray owns a group of G lanes (G = 4, a DPP quad, in the product; 8 and 16 build and are tested):
//   * the group's lanes share deriv_num's seven central differences.  G = 4: lanes 0..2 take the position
//     differences (two equilibria + two determinants each), lane 3 the unperturbed point (box test); then lanes 0..2
//     take the wave-vector differences and lane 3 the frequency difference, all four at the unperturbed fields lane 3
//     has just evaluated and broadcast.  A right-hand side is 2 equilibria + 4 determinants deep instead of 9 + 14;
//     the differences are exchanged with DPP quad_perm moves (one full-rate VALU instruction per dword);
//   * lane j owns components j and j + G of the ODE vector: its rows of the divided differences phi(l, 1:16) are
//     registers for the orders in use (rows 1..6 + the round-off rows; the rest, reached in 0.03 % of the steps, in
//     the launch's workspace); predictor, corrector and difference updates (ode_RAYS.f90:985-1011, 1128-1164) are
//     two operations per lane instead of seven, the weighted norms of `step` are gathered in the reference's
//     summation order;
//   * the scalar part of `step` (order and step-size selection, the coefficient recurrences psi, alpha, beta, sig,
//     g, v) is replicated in the group's lanes; the coefficient vectors and the rarely touched per-ray scalars live
//     in LDS, one column per RAY;
//   * a wave holds 16 rays in 168 VGPRs, three waves share a SIMD (their LDS round trips overlap), and a 64k fan is
//     4096 waves: finished groups pull the next ray, most of the idle tail is gone.
// What it costs: the scalar part of `step` is paid per 64 / G rays instead of per 64 (it does not shrink with G),
// which is why small groups win: measured on the 64k-ray Solovev fan (cfg 3, ms per pass; one ray per lane: 296-311)
// G = 8: 396 -> 327 (phi out of scratch, parameter loads per trip), G = 4: 266 -> 250 -> 242 (DPP, gstr table) ->
// 234 (shared unperturbed equilibrium) -> 212 (three waves per SIMD).  DESIGN.md 4.4.
// Every floating-point operation is the reference's, in its order: results are bit-identical to rays_sg.hpp's
// and to the reference (tests/test_cpu_group_emul.py on the host wave emulator, tests/test_gpu_baseline_kernels.py).
//
// Cross-lane operations (__any, __ballot, __shfl) are only issued from wave-uniform control flow.
#pragma once

#include "rays_sg.hpp"

namespace rays {

template <int G>
struct GrpGeom {
  static_assert(G == 4 || G == 8 || G == 16, "lanes per ray");
  static constexpr int NV = 7;
  static constexpr int CPL = (NV + 1 + G - 1) / G;  // per lane: ODE components = difference pairs (7 + the base point)
  static constexpr int kBaseLane = NV % G, kBaseSlot = NV / G;  // who holds "pair 7", the unperturbed equilibrium
  static constexpr int kThreads = 256;  // the launch block (rays_launch.hpp: kBlock)
  static constexpr int kRaysPerBlock = kThreads / G;
  static constexpr int kCoefLen = 13, kCoefArrays = 6, kScalars = 20, kInts = 8;
  static constexpr int kDoublesPerRay = kCoefArrays * kCoefLen + kScalars;
  static constexpr size_t kLdsBytes = 0;  // (dynamic LDS: none -- the coefficient columns are static arrays of the kernel)
};

// value held by lane j of this lane's group (all lanes of the wave take part)
#ifndef RAYS_HOST_EMUL
// G = 4: a group is a quad, and DPP's quad_perm broadcasts lane j of every quad in ONE full-rate VALU move per dword
// (immediate pattern: no address register, no trip through the LDS crossbar, nothing to wait for).
template <int J>
RAYS_DEV int quad_bcast(int x) {
  return __builtin_amdgcn_update_dpp(x, x, J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false);
}
RAYS_DEV int quad_bcast_j(int x, int j) {  // j is a constant after unrolling: the switch folds away
  switch (j & 3) {
    case 0: return quad_bcast<0>(x);
    case 1: return quad_bcast<1>(x);
    case 2: return quad_bcast<2>(x);
    default: return quad_bcast<3>(x);
  }
}
#endif
template <int G>
RAYS_DEV int grp_bcast(int x, int j) {
#ifndef RAYS_HOST_EMUL
  if (G == 4) return quad_bcast_j(x, j);
#endif
  return __shfl(x, j, G);
}
template <int G>
RAYS_DEV double grp_bcast(double x, int j) {
#ifndef RAYS_HOST_EMUL
  if (G == 4) return __hiloint2double(quad_bcast_j(__double2hiint(x), j), quad_bcast_j(__double2loint(x), j));
#endif
  return __shfl(x, j, G);
}
// sum over the ODE components in the reference's order l = 1..7 (component l lives in slot l / G of lane l % G)
template <int G>
RAYS_DEV double grp_sum(const double t[GrpGeom<G>::CPL]) {
  double s = 0.;
#pragma unroll
  for (int l = 0; l < 7; l++) s = s + grp_bcast<G>(t[l / G], l % G);
  return s;
}

// The coefficient vectors of `step` (psi, alpha, beta, sig, g, v; ode_RAYS.f90:662-672), one LDS column per ray.
// Every lane of the group computes the same values and writes them (same address, same data); a lane only ever
// reads back what it wrote itself.
struct GCoef {
  enum { PSI = 0, ALPHA, BETA, SIG, GG, V, SC, kArrays };
  // One LDS array per coefficient vector (distinct objects: the compiler then knows that a write of beta(i) cannot
  // change psi(i - 1) and does not wait for it), each [14][rays per block]; `a[x]` points at this ray's column.
  sg_lds_ptr a[kArrays];
#ifdef RAYS_HOST_EMUL
  int* ia;
#else
  __attribute__((address_space(3))) int* ia;  // the ray's column of integer scalars
#endif
  int stride;
  struct Ref {
    sg_lds_ptr p;
    RAYS_DEV operator double() const { return *p; }
    RAYS_DEV const Ref& operator=(double x) const { *p = x; return *this; }
    RAYS_DEV const Ref& operator=(const Ref& o) const { return *this = (double)o; }
  };
  RAYS_DEV Ref at(int arr, int i) const { return Ref{a[arr] + (i - 1) * stride}; }
  RAYS_DEV Ref psi(int i) const { return at(PSI, i); }      // 1..12
  RAYS_DEV Ref alpha(int i) const { return at(ALPHA, i); }  // 1..12
  RAYS_DEV Ref beta(int i) const { return at(BETA, i); }    // 1..12
  RAYS_DEV Ref sig(int i) const { return at(SIG, i); }      // 1..13
  RAYS_DEV Ref g(int i) const { return at(GG, i); }         // 1..13
  RAYS_DEV Ref v(int i) const { return at(V, i); }          // 1..12
  RAYS_DEV Ref wi(int i) const { return at(ALPHA, i); }     // intrp's w(1:14) reuses alpha (the integrator restarts after it)
  RAYS_DEV Ref gi(int i) const { return at(SIG, i); }       // intrp's g(1:13) reuses sig
  // Per-ray scalars of trace_rays / SG_ode / de that are touched once or twice per trip (or per interval): kept in
  // LDS, one column per ray, instead of in registers of every lane of the group.
  enum { T_ = 0, TOUT, ABSDEL, TEND, RELEPS, ABSEPS, SOUT, DS_RAY, LAST_RESID, PREV_RESID, MAXR, REL_ERR, ABS_ERR, HOLD,
         XOLD, ROUND, P5EPS, ABSH, ERK, ERKM1 };
  RAYS_DEV Ref sc(int i) const { return Ref{a[SC] + i * stride}; }
  // ... and the integer ones of `step` / de that a trip touches once or twice
  enum { NSTEP = 0, KOLD, NS_, KNEW, IFAIL, NOSTEP, KLE4 };
  struct IRef {
#ifdef RAYS_HOST_EMUL
    int* p;
#else
    __attribute__((address_space(3))) int* p;
#endif
    RAYS_DEV operator int() const { return *p; }
    RAYS_DEV const IRef& operator=(int x) const { *p = x; return *this; }
    RAYS_DEV const IRef& operator=(const IRef& o) const { return *this = (int)o; }
  };
  RAYS_DEV IRef isc(int i) const { return IRef{ia + i * stride}; }
};

// This lane's rows of the divided differences phi(l, 1:16) for its C components.  Rows are indexed by the ray's
// order k, which stays low because SG_ode restarts the integrator on every output interval (k <= 4 in 99.7 % of the
// step attempts of the Solovev fan, k <= 6 in 99.97 %):
//   rows 1..KR (= 6) and the two round-off rows 15, 16   registers; every loop over them is unrolled and BRANCH-FREE
//                                                        (selects on the ray's bounds; `ncap` = a wave-uniform bound
//                                                        on the rows in play ends the unrolled loop with a scalar branch)
//   rows KR+1..14                                        the launch's global workspace (TraceArgs::sg_far, [slot][lane]);
//                                                        reached behind a wave-uniform test `any_hi`
// (a fully unrolled loop must not `break`: with an early exit LLVM leaves a rolled remainder, the rows are then
// indexed dynamically and the whole array moves to scratch memory)
#ifndef RAYS_SG_GROUP_KR
#define RAYS_SG_GROUP_KR 6  // (tests/hip_emul builds a second library with 2, so that ordinary rays cross the tier boundary)
#endif
template <int C>
struct GPhi {
  static constexpr int KR = RAYS_SG_GROUP_KR, kFarRows = 14 - KR, kFarDoubles = kFarRows * C;
  static_assert(KR >= 2 && KR <= 12, "rows 1 and 2 are addressed directly");
  double lo[KR + 2][C];  // [1..KR]; [KR+1] is a zero pad for the unrolled restore loop
  double r15[C], r16[C];
  const TraceArgs* args;  // the workspace column is re-derived where a row above the register tier is touched (rare):
                          // two pointers less to carry through the wave loop
  RAYS_DEV static long long fstride() { return (long long)gridDim.x * blockDim.x; }
  RAYS_DEV double* far() const { return cold_args(*args).sg_far + ((long long)blockIdx.x * blockDim.x + threadIdx.x); }
  RAYS_DEV double& hi(int q, int c) const { return far()[((q - KR - 1) * C + c) * fstride()]; }
  RAYS_DEV void clear() {
#pragma unroll
    for (int q = 0; q < KR + 2; q++)
#pragma unroll
      for (int c = 0; c < C; c++) lo[q][c] = 0.;
#pragma unroll
    for (int c = 0; c < C; c++) r15[c] = r16[c] = 0.;
  }
  RAYS_DEV void get(int i, double out[C], bool any_hi) const {  // out = phi(:, i); i outside 1..14: zeros
#pragma unroll
    for (int c = 0; c < C; c++) out[c] = 0.;
#pragma unroll
    for (int q = 1; q <= KR; q++)
#pragma unroll
      for (int c = 0; c < C; c++) out[c] = i == q ? lo[q][c] : out[c];
    if (RAYS_RARE(any_hi)) {
      if (i > KR && i <= 14) {
#pragma unroll
        for (int c = 0; c < C; c++) out[c] = hi(i, c);
      }
    }
  }
  RAYS_DEV void set(int i, const double in[C], bool any_hi) {
#pragma unroll
    for (int q = 1; q <= KR; q++)
#pragma unroll
      for (int c = 0; c < C; c++) lo[q][c] = i == q ? in[c] : lo[q][c];
    if (RAYS_RARE(any_hi)) {
      if (i > KR && i <= 14) {
#pragma unroll
        for (int c = 0; c < C; c++) hi(i, c) = in[c];
      }
    }
  }
  // phi(:, i) = beta(i) * phi(:, i), i = a..b                       (ode_RAYS.f90:992-996)
  RAYS_DEV void scale(int a, int b, const GCoef& S, int ncap, bool any_hi) {
#pragma unroll
    for (int q = 1; q <= KR; q++) {
      if (q > ncap) continue;
      const bool on = q >= a && q <= b;
      const double bq = S.beta(q);
#pragma unroll
      for (int c = 0; c < C; c++) lo[q][c] = on ? bq * lo[q][c] : lo[q][c];
    }
    if (RAYS_RARE(any_hi)) {
      for (int q = a > KR + 1 ? a : KR + 1; q <= b; q++) {
        const double bq = S.beta(q);
#pragma unroll
        for (int c = 0; c < C; c++) hi(q, c) = bq * hi(q, c);
      }
    }
  }
  // predictor (:1003-1011), i = k..1:  p += phi(:,i)*g(i); phi(:,i) += phi(:,i+1)   (phi(:,k+1) has just been zeroed)
  RAYS_DEV void predict(int k, const GCoef& S, double pp[C], int ncap, bool any_hi) {
    double up[C];
#pragma unroll
    for (int c = 0; c < C; c++) up[c] = 0.;
    if (RAYS_RARE(any_hi)) {
      for (int q = k; q > KR; q--) {
        const double gg = S.g(q);
#pragma unroll
        for (int c = 0; c < C; c++) {
          double x = hi(q, c);
          pp[c] = pp[c] + x * gg;
          x = x + up[c];
          hi(q, c) = x;
          up[c] = x;
        }
      }
    }
#pragma unroll
    for (int q = KR; q >= 1; q--) {
      if (q > ncap) continue;
      const bool on = q <= k;
      const double gg = S.g(q);
#pragma unroll
      for (int c = 0; c < C; c++) {
        const double x = lo[q][c];
        const double pn = pp[c] + x * gg, xn = x + up[c];
        pp[c] = on ? pn : pp[c];
        lo[q][c] = on ? xn : x;
        up[c] = on ? xn : up[c];
      }
    }
  }
  // failed step (:1090-1094), i = 1..k ascending:  phi(:,i) = (phi(:,i) - phi(:,i+1)) / beta(i)
  RAYS_DEV void restore(int k, const GCoef& S, int ncap, bool any_hi) {
    double nxt[C];  // phi(:, KR+1), not yet modified when row KR is restored
#pragma unroll
    for (int c = 0; c < C; c++) nxt[c] = 0.;
    if (RAYS_RARE(any_hi)) {
      if (k >= KR) {
#pragma unroll
        for (int c = 0; c < C; c++) nxt[c] = hi(KR + 1, c);
      }
    }
#pragma unroll
    for (int q = 1; q <= KR; q++) {
      if (q > ncap) continue;
      const bool on = q <= k;
      const Recip b = make_recip(S.beta(q));
#pragma unroll
      for (int c = 0; c < C; c++) {
        const double above = q < KR ? lo[q + 1][c] : nxt[c];  // old value: rows are restored in ascending order
        const double x = div(lo[q][c] - above, b);
        lo[q][c] = on ? x : lo[q][c];
      }
    }
    if (RAYS_RARE(any_hi)) {
      for (int q = KR + 1; q <= k; q++) {
        const Recip b = make_recip(S.beta(q));
#pragma unroll
        for (int c = 0; c < C; c++) hi(q, c) = div(hi(q, c) - hi(q + 1, c), b);
      }
    }
  }
  // phi(:, i) += d, i = 1..k                                         (:1160-1164)
  RAYS_DEV void add(int k, const double d[C], int ncap, bool any_hi) {
#pragma unroll
    for (int q = 1; q <= KR; q++) {
      if (q > ncap) continue;
#pragma unroll
      for (int c = 0; c < C; c++) lo[q][c] = q <= k ? lo[q][c] + d[c] : lo[q][c];
    }
    if (RAYS_RARE(any_hi)) {
      for (int q = KR + 1; q <= k; q++)
#pragma unroll
        for (int c = 0; c < C; c++) hi(q, c) = hi(q, c) + d[c];
    }
  }
  // intrp (:1343-1349), i = ki..1:  yout += g(i) * phi(:, i)
  RAYS_DEV void interp(int ki, const GCoef& S, double yout[C], bool any_hi) const {
    if (RAYS_RARE(any_hi)) {
      for (int q = ki; q > KR; q--) {
        const double gg = S.gi(q);
#pragma unroll
        for (int c = 0; c < C; c++) yout[c] = yout[c] + gg * hi(q, c);
      }
    }
#pragma unroll
    for (int q = KR; q >= 1; q--) {
      const bool on = q <= ki;
      const double gg = S.gi(q);
#pragma unroll
      for (int c = 0; c < C; c++) yout[c] = on ? yout[c] + gg * lo[q][c] : yout[c];
    }
  }
};
// doubles per lane of the launch's workspace (TraceArgs::sg_far) the lane-group kernel needs
template <int G>
constexpr int sg_group_far_doubles_per_lane() { return GPhi<GrpGeom<G>::CPL>::kFarDoubles + GrpGeom<G>::CPL; }  // + y of SG_ode

// Equilibrium for determ at a (possibly perturbed) point; returns the equilibrium's stop code when check_box is set
// (the group's base lane: equilibrium_m.f90:198-202, eqn_ray.f90:90-102).
template <int EQ, int NS>
RAYS_DEV int eq_for_determ_err(const DevParams& P, const Recip& Romgrf, const Recip& Romgrf2, const double rvec[3],
                               double bunit[3], double alpha[NS], double gamma[NS], bool check_box,
                               double* omgc = nullptr, double* omgp2 = nullptr) {
  EqPoint<NS> e;
  equilibrium<EQ, NS>(P, Romgrf, Romgrf2, rvec, e, check_box);
#pragma unroll
  for (int i = 0; i < 3; i++) bunit[i] = e.bunit[i];
#pragma unroll
  for (int is = 0; is < NS; is++) {
    alpha[is] = e.alpha[is];
    gamma[is] = e.gamma[is];
    if (omgc) omgc[is] = e.omgc[is];
    if (omgp2) omgp2[is] = e.omgp2[is];
  }
  return e.err;
}

// ONE evaluation of eqn_ray (ray_deriv_name = 'numerical') for the group's ray, optionally with check_save at the
// same state.  win: this lane's components of the state.  f: this lane's components of dv/ds.  The other outputs are
// the same in every lane of the group.  Call from wave-uniform control flow.
//   deriv_num.f90:40-84: dddx(i) = (D(r + delta e_i) - D(r - delta e_i)) / (2 delta), dddk(i) likewise with
//   change = max(delta, |delta k_i|) / 2, dddw from omgrf (1 +- delta/2) with k0 rescaled; D = determ (:99-153).
template <int EQ, int NS, int G>
RAYS_DEV void group_rhs(const DevParams& P_in, const double win[GrpGeom<G>::CPL], int gl, bool do_check, bool any_check,
                        double f[GrpGeom<G>::CPL], int& code, double& resid, int& cs_flag, bool& cs_stop) {
  const DevParams& P = cold_params(P_in);  // the run constants are loaded here, per trip (rays_trace.hpp)
  typedef GrpGeom<G> GEO;
  constexpr int C = GEO::CPL;
  double v[7];
#pragma unroll
  for (int l = 0; l < 7; l++) v[l] = grp_bcast<G>(win[l / G], l % G);
  const double rvec[3] = {v[0], v[1], v[2]}, kvec[3] = {v[3], v[4], v[5]};
  double d[C];
  int err_base = 0;
  double dd[7];
  int err;
  if constexpr (G == 4) {
    // Four lanes, two slots.  Slot 0: lanes 0..2 take the position differences (two equilibria + two determinants
    // each), lane 3 the unperturbed point (box test).  Slot 1: lanes 0..2 the wave-vector differences, lane 3 the
    // frequency difference -- all four at the UNPERTURBED fields (deriv_num.f90:60-80), which lane 3 has just
    // evaluated: its bunit, alpha, gamma (and, for the frequency difference, omgc and omgp2, of which alpha and gamma
    // are the quotients by omgrf**2 and omgrf: equilibrium_m.f90:262-265) are broadcast instead of evaluating the
    // equilibrium eight more times.  Same operands, same operations: same bits.
    const Recip Ro0 = const_recip(P.omgrf, P.inv_omgrf), Ro20 = const_recip(P.omgrf2, P.inv_omgrf2);
    const Recip Rk00 = const_recip(P.k0, P.inv_k0);
    double b0[3], a0[NS], g0[NS], oc0[NS], op0[NS];
#pragma unroll
    for (int i = 0; i < 3; i++) b0[i] = 0.;
#pragma unroll
    for (int is = 0; is < NS; is++) a0[is] = g0[is] = oc0[is] = op0[is] = 0.;
    {
      double det_plus = 0., det_minus = 0.;
#pragma unroll 1
      for (int sgn = 0; sgn < 2; sgn++) {
        const bool plus = sgn == 0;
        const double dr = plus ? P.delta : -P.delta;
        double rr[3], bu[3], al[NS], ga[NS], oc[NS], op[NS];
#pragma unroll
        for (int i = 0; i < 3; i++) rr[i] = (gl == i) ? rvec[i] + dr : rvec[i];  // :42-43, :48
        const int e1 = eq_for_determ_err<EQ, NS>(P, Ro0, Ro20, rr, bu, al, ga, plus && gl == 3, oc, op);
        const double det = determ<NS>(bu, al, ga, kvec, Rk00);
        if (plus) {
          det_plus = det;
          err_base = e1;  // (lane 3's is the one that is read)
#pragma unroll
          for (int i = 0; i < 3; i++) b0[i] = bu[i];
#pragma unroll
          for (int is = 0; is < NS; is++) {
            a0[is] = al[is];
            g0[is] = ga[is];
            oc0[is] = oc[is];
            op0[is] = op[is];
          }
        } else {
          det_minus = det;
        }
      }
      d[0] = (det_plus - det_minus) / P.two_delta;  // :50
    }
    // the unperturbed point's fields, from lane 3
#pragma unroll
    for (int i = 0; i < 3; i++) b0[i] = grp_bcast<G>(b0[i], 3);
#pragma unroll
    for (int is = 0; is < NS; is++) {
      a0[is] = grp_bcast<G>(a0[is], 3);
      g0[is] = grp_bcast<G>(g0[is], 3);
      oc0[is] = grp_bcast<G>(oc0[is], 3);
      op0[is] = grp_bcast<G>(op0[is], 3);
    }
    {
      const bool w = gl == 3;
      const double kc = gl == 0 ? kvec[0] : (gl == 1 ? kvec[1] : kvec[2]);
      const double change = fmax(P.delta, fabs(P.delta * kc)) / 2.;  // :61
      double det_plus = 0., det_minus = 0.;
#pragma unroll 1
      for (int sgn = 0; sgn < 2; sgn++) {
        const bool plus = sgn == 0;
        const double dk = plus ? change : -change;
        double kk[3], al[NS], ga[NS];
#pragma unroll
        for (int i = 0; i < 3; i++) kk[i] = (gl == i) ? kvec[i] + dk : kvec[i];  // :62, :65
        const Recip Ro = const_recip(plus ? P.omgrf_p : P.omgrf_m, plus ? P.inv_omgrf_p : P.inv_omgrf_m);
        const Recip Ro2 = const_recip(plus ? P.omgrf2_p : P.omgrf2_m, plus ? P.inv_omgrf2_p : P.inv_omgrf2_m);
        const Recip Rk = const_recip(w ? (plus ? P.k0_p : P.k0_m) : P.k0, w ? (plus ? P.inv_k0_p : P.inv_k0_m) : P.inv_k0);
#pragma unroll
        for (int is = 0; is < NS; is++) {  // :71-80 on the frequency lane
          const double aw = div(op0[is], Ro2), gw = div(oc0[is], Ro);
          al[is] = w ? aw : a0[is];
          ga[is] = w ? gw : g0[is];
        }
        const double det = determ<NS>(b0, al, ga, kk, Rk);
        if (plus) det_plus = det;
        else det_minus = det;
      }
      d[1] = (det_plus - det_minus) / (w ? P.omgrf0_delta : 2. * change);  // :63, :80
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      dd[i] = grp_bcast<G>(d[0], i);
      dd[3 + i] = grp_bcast<G>(d[1], i);
    }
    dd[6] = grp_bcast<G>(d[1], 3);
    err = grp_bcast<G>(err_base, 3);
  } else {
#pragma unroll
  for (int c = 0; c < C; c++) {
    const int p = gl + G * c;  // 0..2: d/dx_p   3..5: d/dk_(p-3)   6: d/domega   7: the unperturbed point (box test)   > 7: none
    // :61 change = max(delta, |delta k_i|) / 2 of this lane's wave-vector component (p = 3..5)
    const double kc = p == 3 ? kvec[0] : (p == 4 ? kvec[1] : kvec[2]);
    const double change = fmax(P.delta, fabs(P.delta * kc)) / 2.;
    const bool w = p == 6;  // :71-80: per-lane omgrf / k0 (the reference rewrites its module variables)
    // the two determinants of the difference, one after the other through ONE copy of the code (rolled loop: half
    // the instructions and half the live registers of two inlined evaluations); a - b is a + (-b) bit for bit
    double det_plus = 0., det_minus = 0.;
#pragma unroll 1
    for (int sgn = 0; sgn < 2; sgn++) {
      const bool plus = sgn == 0;
      const double dr = plus ? P.delta : -P.delta, dk = plus ? change : -change;
      double rr[3], kk[3], bu[3], al[NS], ga[NS];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        rr[i] = (p == i) ? rvec[i] + dr : rvec[i];      // :42-43, :48
        kk[i] = (p == 3 + i) ? kvec[i] + dk : kvec[i];  // :62, :65
      }
      const Recip Ro = const_recip(w ? (plus ? P.omgrf_p : P.omgrf_m) : P.omgrf, w ? (plus ? P.inv_omgrf_p : P.inv_omgrf_m) : P.inv_omgrf);
      const Recip Ro2 = const_recip(w ? (plus ? P.omgrf2_p : P.omgrf2_m) : P.omgrf2, w ? (plus ? P.inv_omgrf2_p : P.inv_omgrf2_m) : P.inv_omgrf2);
      const Recip Rk = const_recip(w ? (plus ? P.k0_p : P.k0_m) : P.k0, w ? (plus ? P.inv_k0_p : P.inv_k0_m) : P.inv_k0);
      const int e1 = eq_for_determ_err<EQ, NS>(P, Ro, Ro2, rr, bu, al, ga, plus && p == 7);
      const double det = determ<NS>(bu, al, ga, kk, Rk);
      if (plus) {
        det_plus = det;
        if (p == 7) err_base = e1;
      } else {
        det_minus = det;
      }
    }
    const double den = p < 3 ? P.two_delta : (p < 6 ? 2. * change : P.omgrf0_delta);
    d[c] = (det_plus - det_minus) / den;  // :50, :63, :80
  }
#pragma unroll
  for (int p = 0; p < 7; p++) dd[p] = grp_bcast<G>(d[p / G], p % G);
  err = grp_bcast<G>(err_base, GEO::kBaseLane);
  }
  const double dddx[3] = {dd[0], dd[1], dd[2]}, dddk[3] = {dd[3], dd[4], dd[5]};
  double dvds[7];
#pragma unroll
  for (int l = 0; l < 7; l++) dvds[l] = 0.;
  EqPoint<NS> eq_unused;  // (ray_equations reads it for the damping / gradient rows only; nv = 7 has neither)
  eq_unused.err = 0;
  const int rc = ray_equations<NS, 7, false>(P, eq_unused, kvec, 0., dddx, dddk, dd[6], dvds);
  code = err ? err : rc;  // eqn_ray.f90:90-102 returns before the derivatives
#pragma unroll
  for (int c = 0; c < C; c++) {
    f[c] = 0.;
#pragma unroll
    for (int l = 0; l < 7; l++)
      if (l == gl + G * c) f[c] = dvds[l];
  }
  // check_save at this state (once per output interval): equilibrium with the box test, residual, deriv_cold's
  // dD/dw -- the base lane evaluates it with the cold kernels' fused routine, whose check outputs do not depend on
  // the derivative model (rays_device.hpp: rhs_eval)
  resid = 0.;
  cs_flag = 0;
  cs_stop = false;
  if (any_check) {  // wave-uniform
    double r_ = 0.;
    int fl_ = 0, st_ = 0;
    if (do_check && gl == GEO::kBaseLane) {
      int code_u = 0;
      bool stop_u = false;
      double f_u[7];
      rhs_eval<EQ, NS, RAYS_DERIV_COLD, 7>(P, v, true, r_, fl_, stop_u, code_u, f_u);
      st_ = stop_u ? 1 : 0;
    }
    resid = grp_bcast<G>(r_, GEO::kBaseLane);
    cs_flag = grp_bcast<G>(fl_, GEO::kBaseLane);
    cs_stop = grp_bcast<G>(st_, GEO::kBaseLane) != 0;
  }
}

#ifndef RAYS_SG_GROUP_WAVES
// waves per SIMD the kernel is compiled for (register budget 512 / this).  Measured on the 64k-ray Solovev fan, G = 4:
// 2 waves (256 VGPRs, 1 spill) 235 ms, 3 waves (168 VGPRs, 141 spills -- most of them in check_save's share, which
// runs once per output interval) 212 ms, 4 waves (128 VGPRs) 331 ms.
#define RAYS_SG_GROUP_WAVES 3
#endif
#ifndef RAYS_HOST_EMUL
template <int EQ, int NS, int G>
__global__ void __launch_bounds__(256, RAYS_SG_GROUP_WAVES)
sg_group_kernel(const DevParams P_kernarg, const TraceArgs A_hot)
#else
template <int EQ, int NS, int G>
void sg_group_kernel(const DevParams P_kernarg, const TraceArgs A_hot)
#endif
{
  typedef GrpGeom<G> GEO;
  constexpr int C = GEO::CPL, NV = 7;
  // (no hot_params copy into vector registers: this kernel is built for several waves per SIMD, where registers are
  // what is scarce and a scalar load of a run constant is hidden by the other waves)
  const DevParams& P = P_kernarg;
  extern __shared__ double lds[];
  const int gl = threadIdx.x & (G - 1);  // lane within the ray's group
  const int gib = threadIdx.x / G;       // group within the block
  GCoef S;
#ifdef RAYS_HOST_EMUL
  // (the host wave emulator runs the lanes of a wave one after the other between cross-lane operations, not in
  // lock step: a column shared by the group's lanes would be read by one lane after another has updated it, so
  // there every lane keeps a copy of its own; the values are the same)
  double coef_private[GCoef::kArrays][20];
  int icoef_private[8];
  for (int x = 0; x < GCoef::kArrays; x++) S.a[x] = coef_private[x];
  S.ia = icoef_private;
  S.stride = 1;
  (void)lds;
#else
  {
    constexpr int R = GEO::kRaysPerBlock;
    // (13 entries each: psi, beta, v use 12, sig, g 13; intrp's w(1:13) in alpha -- its w(14) is never read, its
    // loops end at ki + 1 - j <= 12)
    __shared__ double s_psi[13 * R], s_alpha[13 * R], s_beta[13 * R], s_sig[13 * R], s_g[13 * R], s_v[13 * R], s_sc[20 * R];
    __shared__ int s_isc[8 * R];
    S.ia = (__attribute__((address_space(3))) int*)(s_isc + gib);
    S.a[GCoef::PSI] = (sg_lds_ptr)(s_psi + gib);
    S.a[GCoef::ALPHA] = (sg_lds_ptr)(s_alpha + gib);
    S.a[GCoef::BETA] = (sg_lds_ptr)(s_beta + gib);
    S.a[GCoef::SIG] = (sg_lds_ptr)(s_sig + gib);
    S.a[GCoef::GG] = (sg_lds_ptr)(s_g + gib);
    S.a[GCoef::V] = (sg_lds_ptr)(s_v + gib);
    S.a[GCoef::SC] = (sg_lds_ptr)(s_sc + gib);
    S.stride = R;
  }
  // gstr(1:13) (ode_RAYS.f90:776-779) as a table in LDS: indexed by the ray's order, a lookup instead of rays_sg.hpp's
  // select chain over thirteen constants (which costs ~40 instructions a call, four calls per step)
  __shared__ double s_gstr[16];
  if (threadIdx.x < 14) s_gstr[threadIdx.x] = gstr((int)threadIdx.x);
  __syncthreads();
  const sg_lds_ptr gstr_tab = (sg_lds_ptr)s_gstr;
  (void)lds;
#define gstr(i) (gstr_tab[(i)])
#endif
  GPhi<C> F;
  F.clear();
  F.args = &A_hot;
  bool valid[C];  // this lane's slot holds an ODE component
#pragma unroll
  for (int c = 0; c < C; c++) valid[c] = gl + G * c < NV;

  const unsigned total_groups = gridDim.x * GEO::kRaysPerBlock;
  const long long npt = (long long)P.nstep_max + 1;
  constexpr double kEps = 2.220446049250313e-16;  // epsilon(1._rkind)
  constexpr double twou = 2.0 * kEps, fouru = 2.0 * twou;
  constexpr int maxnum = 500;  // ode_RAYS.f90:395

  // ---- per-ray state, the same in every lane of the group unless marked (lane) --------------------------------------
  int ray = blockIdx.x * GEO::kRaysPerBlock + gib;
  bool alive = ray < A_hot.nray;
  bool need_init = alive;
  int pc = PC_CHECK;
  double yy[C], pp[C];  // (lane) y of `step`, the predicted p
  // (y of SG_ode -- the state `ode` last returned, read when the ray stops -- lives behind the workspace rows)
#define YSAVE(c) F.far()[(long long)(GPhi<C>::kFarDoubles + (c)) * GPhi<C>::fstride()]
  Recip wt[C];                    // (lane)
  double x = 0., h = 0., eps = 0.;
#define p5eps S.sc(GCoef::P5EPS)
#define absh S.sc(GCoef::ABSH)
#define erk S.sc(GCoef::ERK)
#define erkm1 S.sc(GCoef::ERKM1)
  // (t, tout, absdel, tend, releps, abseps, sout, ds_ray, the residual statistics, rel_err, abs_err, hold, xold, round:
  // in the ray's LDS column, GCoef::sc)
#define t S.sc(GCoef::T_)
#define tout S.sc(GCoef::TOUT)
#define absdel S.sc(GCoef::ABSDEL)
#define tend S.sc(GCoef::TEND)
#define releps S.sc(GCoef::RELEPS)
#define abseps S.sc(GCoef::ABSEPS)
#define sout S.sc(GCoef::SOUT)
#define ds_ray S.sc(GCoef::DS_RAY)
#define last_resid S.sc(GCoef::LAST_RESID)
#define prev_resid S.sc(GCoef::PREV_RESID)
#define maxr S.sc(GCoef::MAXR)
#define rel_err S.sc(GCoef::REL_ERR)
#define abs_err S.sc(GCoef::ABS_ERR)
#define hold S.sc(GCoef::HOLD)
#define xold S.sc(GCoef::XOLD)
#define round_ S.sc(GCoef::ROUND)
  int k = 1;
#define nstep S.isc(GCoef::NSTEP)
#define kold S.isc(GCoef::KOLD)
#define ns S.isc(GCoef::NS_)
#define knew S.isc(GCoef::KNEW)
#define ifail S.isc(GCoef::IFAIL)
#define nostep S.isc(GCoef::NOSTEP)
#define kle4 S.isc(GCoef::KLE4)
  kold = 0;
  ns = 0;
  knew = 1;
  ifail = 0;
  nostep = 0;
  kle4 = 0;
  nstep = 0;
  int resume = SEG_WAIT;
  int waited = 0;
  int patience = kSgPatienceMax, probe = 0;
  unsigned fl = FL_START | FL_PHASE1 | FL_NORND | FL_FIRST;
#pragma unroll
  for (int c = 0; c < C; c++) {
    yy[c] = pp[c] = 0.;
    wt[c] = make_recip(1.0);
  }

  SG_PROF_DECL
  while (__any(alive)) {
    SG_PROF(0);  // loop, refill
    if (RAYS_RARE(need_init)) {  // ray_tracing.f90:77-93, SG_ode_m.f90:73-85
      const TraceArgs& A = cold_args(A_hot);
      double v0[NV], s_start = 0., ds_start = 0.;
      start_ray<EQ, NS, NV>(P, A, ray, v0, s_start, ds_start);
      sout = s_start;
      ds_ray = ds_start;
#pragma unroll
      for (int c = 0; c < C; c++) {
        yy[c] = 0.;
#pragma unroll
        for (int l = 0; l < NV; l++)
          if (l == gl + G * c) yy[c] = v0[l];
        YSAVE(c) = yy[c];
      }
      pc = PC_CHECK;
      resume = SEG_WAIT;
      fl |= FL_FIRST;
      nstep = 0;
      t = sout;
      last_resid = 0.;
      prev_resid = 0.;
      maxr = -1.7976931348623157e308;
      rel_err = P.rel_err0;
      abs_err = P.abs_err0;
      need_init = false;
    }

    // wave-uniform bound on the orders in play this trip (an order grows by at most one per trip)
    int kcap = 2;
#pragma unroll
    for (int q = 2; q <= 12; q++) {
      if (!__any(alive && k >= q)) break;
      kcap = q + 1;
    }
    if (kcap > 12) kcap = 12;
    const int ncap = kcap < GPhi<C>::KR ? kcap : GPhi<C>::KR;  // register rows in play
    const bool any_hi = kcap + 2 > GPhi<C>::KR;                 // a ray of this wave may touch the workspace rows

    // ---- which lanes does this trip's right-hand side serve (phase and interval alignment: rays_sg.hpp) -------------
    const bool in_f2 = pc == PC_F2;
    const bool wants_rhs = alive && resume == SEG_WAIT;
    const bool at_check = pc == PC_CHECK;
    bool act;
    if (patience > 0) {
      const int n_f2 = __popcll(__ballot(wants_rhs && in_f2));
      const int n_f3 = __popcll(__ballot(wants_rhs && !in_f2 && !at_check));
      const bool all_arrived = n_f2 + n_f3 == 0;
      const bool timed_out = !all_arrived && __any(wants_rhs && at_check && waited >= patience);
      const bool serve_check = all_arrived || timed_out;
      const bool serve_f2 = n_f2 > n_f3;
      act = wants_rhs && (serve_check ? at_check : (!at_check && in_f2 == serve_f2));
      waited = (wants_rhs && at_check && !serve_check) ? waited + 1 : 0;
      if (timed_out) patience = patience / 2;
      else if (all_arrived) patience = patience + 8 < kSgPatienceMax ? patience + 8 : kSgPatienceMax;
      probe = 0;
    } else {
      const int n_f2 = __popcll(__ballot(wants_rhs && in_f2)), n_other = __popcll(__ballot(wants_rhs && !in_f2));
      const bool serve_f2 = n_f2 > n_other;
      act = wants_rhs && (in_f2 == serve_f2);
      waited = 0;
      probe = probe + 1;
      if (probe >= kSgProbeTrips) patience = kSgPatienceProbe;
    }

    // ---- the one right-hand side of this trip, spread over the group's lanes ---------------------------------------
    double win[C], f[C], resid = 0.;
    int code = 0, cs_flag = 0;
    bool cs_stop = false;
#pragma unroll
    for (int c = 0; c < C; c++) win[c] = pc == PC_F2 ? pp[c] : yy[c];
    const bool do_check = act && pc == PC_CHECK;
    const bool any_check = __any(do_check);
    SG_PROF(1);  // init, order cap, phase vote
#ifdef RAYS_SG_PROFILE
    prof_acc[25] += __popcll(__ballot(act));
    prof_acc[26] += 1;
    prof_acc[27] += __popcll(__ballot(alive));
    prof_acc[28] += __popcll(__ballot(alive && at_check && !act));
#endif
    group_rhs<EQ, NS, G>(P, win, gl, do_check, any_check, f, code, resid, cs_flag, cs_stop);
    SG_PROF(any_check ? 15 : 2);  // RHS (slot 15: trips that also run check_save)

    // ---- per-ray continuation (flat control flow, segments in pipeline order: rays_sg.hpp) -------------------------
    int stop = 0;
    int done = 0;
    int seg = resume;
    resume = SEG_WAIT;
    int have_f = 0;
    if (act) {
      if (pc == PC_CHECK) {
        seg = SEG_DE_BEGIN;
        if (RAYS_RARE(fl & FL_FIRST)) {  // ray_tracing.f90:92-112
          const TraceArgs& A = cold_args(A_hot);
#pragma unroll
          for (int c = 0; c < C; c++)
            if (valid[c]) A.ray_vec[(long long)ray * npt * NV + gl + G * c] = yy[c];
          if (gl == 0) A.residual[(long long)ray * npt] = 0.;
          fl &= ~FL_FIRST;
          if (RAYS_RARE(cs_stop)) {
#pragma unroll
            for (int c = 0; c < C; c++)
              if (valid[c] && A.end_ray_vec) A.end_ray_vec[(long long)ray * NV + gl + G * c] = 0.;
            if (gl == 0) {
              A.npoints[ray] = 1;
              A.stop_code[ray] = cs_flag;
              if (A.end_residuals) A.end_residuals[ray] = 0.;
              if (A.max_residuals) A.max_residuals[ray] = 0.;
            }
            done = 1;
            stop = -1;
            seg = SEG_WAIT;
          }
        } else {
          if (RAYS_RARE(cs_stop)) {  // ray_tracing.f90:214-234
            stop = cs_flag;
            seg = SEG_STOP;
          } else {  // :237-243
            nstep = nstep + 1;
            const TraceArgs& A = cold_args(A_hot);
            const long long pt = (long long)ray * npt + nstep;
#pragma unroll
            for (int c = 0; c < C; c++)
              if (valid[c]) A.ray_vec[pt * NV + gl + G * c] = yy[c];
            if (gl == 0) A.residual[pt] = resid;
            if (fabs(last_resid) > maxr) maxr = fabs(last_resid);
            prev_resid = last_resid;
            last_resid = resid;
          }
        }
        if (seg == SEG_DE_BEGIN) {  // ray_tracing.f90:118-172
          t = sout;
          sout = sout + ds_ray;
          tout = sout;
          if (sout > P.s_max) {
            stop = RAYS_STOP_SOUT_GT_SMAX;
            seg = SEG_STOP;
          } else if (nstep + 1 > P.nstep_max) {
            stop = RAYS_STOP_NSTEP_MAX;
            seg = SEG_STOP;
          }
          have_f = 1;
        }
      } else if (pc == PC_F1) {
        have_f = 1;
        seg = SEG_START_DONE;
      } else if (pc == PC_F2) {
        seg = SEG_AFTER_F2;
      } else {
        seg = SEG_AFTER_F3;
      }
    }

    // ---- AFTER_F2: error estimates, accept or reject (ode_RAYS.f90:1020-1120) -------------------------------------
    SG_PROF(3);  // CHECK bookkeeping
    {
      const bool here = seg == SEG_AFTER_F2 && !code;
      if (RAYS_RARE(seg == SEG_AFTER_F2 && code)) {  // :1020
        stop = code;
        seg = SEG_STOP;
      }
      if (__any(here)) {  // wave-uniform
        const int kp1 = k + 1, km1 = k - 1, km2 = k - 2;
        double rk[C], rkm1[C], tm2[C], tm1[C], t0[C];
        F.get(k, rk, any_hi);
        F.get(km1, rkm1, any_hi);
#pragma unroll
        for (int c = 0; c < C; c++) {
          const double ph1 = F.lo[1][c];
          const double q2 = div(rkm1[c] + f[c] - ph1, wt[c]);
          const double q1 = div(rk[c] + f[c] - ph1, wt[c]);
          const double q0 = div(f[c] - ph1, wt[c]);
          tm2[c] = q2 * q2;
          tm1[c] = q1 * q1;
          t0[c] = q0 * q0;
        }
        const double s2 = grp_sum<G>(tm2), s1 = grp_sum<G>(tm1), s0 = grp_sum<G>(t0);
        if (here) {
          double erkm2 = 0.0;
          erkm1 = 0.0;
          if (0 < km2) erkm2 = absh * S.sig(km1) * gstr(km2) * sqrt(s2);
          if (0 <= km2) erkm1 = absh * S.sig(k) * gstr(km1) * sqrt(s1);
          const double err = absh * sqrt(s0) * (S.g(k) - S.g(kp1));
          erk = absh * sqrt(s0) * S.sig(kp1) * gstr(k);
          knew = k;
          if (0 < km2) {
            if (fmax(erkm1, erkm2) <= erk) knew = km1;
          } else if (0 == km2) {
            if (erkm1 <= 0.5 * erk) knew = km1;
          }
          if (err <= eps) {
            // ---- successful: correct (:1128-1142) ----
            kold = k;
            hold = h;
            const double hg = h * S.g(kp1);
            if (!(fl & FL_NORND)) {
#pragma unroll
              for (int c = 0; c < C; c++) {
                const double rho = hg * (f[c] - F.lo[1][c]) - F.r16[c];
                yy[c] = pp[c] + rho;
                F.r15[c] = (yy[c] - pp[c]) - rho;
              }
            } else {
#pragma unroll
              for (int c = 0; c < C; c++) yy[c] = pp[c] + hg * (f[c] - F.lo[1][c]);
            }
            pc = PC_F3;
            seg = SEG_WAIT;
          } else {
            // ---- failed step: restore, shrink (:1086-1120) ----
            fl &= ~FL_PHASE1;
            x = xold;
            F.restore(k, S, ncap, any_hi);
            for (int i = 2; i <= k; i++) S.psi(i - 1) = S.psi(i) - h;
            ifail = ifail + 1;
            double temp2 = 0.5;
            if (3 < ifail) {
              if (p5eps < 0.25 * erk) temp2 = sqrt(p5eps / erk);
            }
            if (3 <= ifail) knew = 1;
            h = temp2 * h;
            k = knew;
            if (fabs(h) < fouru * fabs(x)) {
              h = copysign(fouru * fabs(x), h);
              eps = eps + eps;
              seg = SEG_CRASH;
            } else {
              seg = SEG_COEF;
            }
          }
        }
      }
    }

    // ---- AFTER_F3: update differences, choose order and step size (:1145-1231) ---------------------------------------
    SG_PROF(4);
    {
      const bool here = seg == SEG_AFTER_F3 && !code;
      if (RAYS_RARE(seg == SEG_AFTER_F3 && code)) {  // :1145
        stop = code;
        seg = SEG_STOP;
      }
      if (__any(here)) {  // wave-uniform
        const int kp1 = k + 1, kp2 = k + 2, km1 = k - 1;
        double d1[C], d2[C], tp[C];
        F.get(kp2, d2, any_hi);
#pragma unroll
        for (int c = 0; c < C; c++) {
          d1[c] = f[c] - F.lo[1][c];
          d2[c] = d1[c] - d2[c];
          const double q = div(d2[c], wt[c]);
          tp[c] = q * q;
        }
        const double sp = grp_sum<G>(tp);
        if (here) {
          F.set(kp1, d1, any_hi);
          F.set(kp2, d2, any_hi);
          F.add(k, d1, ncap, any_hi);
          double erkp1 = 0.0;
          if (knew == km1 || k == 12) fl &= ~FL_PHASE1;
          if (fl & FL_PHASE1) {
            k = kp1;
            erk = erkp1;
          } else if (knew == km1) {
            k = km1;
            erk = erkm1;
          } else if (kp1 <= ns) {
            erkp1 = absh * gstr(kp1) * sqrt(sp);
            if (k == 1) {
              if (erkp1 < 0.5 * erk) {
                k = kp1;
                erk = erkp1;
              }
            } else if (erkm1 <= fmin(erk, erkp1)) {
              k = km1;
              erk = erkm1;
            } else if (erkp1 < erk && k < 12) {
              k = kp1;
              erk = erkp1;
            }
          }
          double hnew = h + h;
          if (!(fl & FL_PHASE1)) {
            const double two_k1 = (double)(2 << k);  // two(k+1)
            if (p5eps < erk * two_k1) {
              hnew = h;
              if (p5eps < erk) {
                const double temp2 = (double)(k + 1);
                const double r = libm::pow(p5eps / erk, 1.0 / temp2);
                hnew = absh * fmax(0.5, fmin((double)0.9f, r));
                hnew = copysign(fmax(hnew, fouru * fabs(x)), h);
              }
            }
          }
          h = hnew;
          // ---- back in de (:579-588) ----
          nostep = nostep + 1;
          kle4 = kle4 + 1;
          if (4 < kold) kle4 = 0;
          if (50 <= kle4) fl |= FL_STIFF;
          seg = SEG_DE_TOP;
        }
      }
    }

    if (RAYS_RARE(seg == SEG_CRASH)) {  // de returns iflag = 3 (:566-575), SG_ode_m.f90:139-149
    SG_PROF(5);
      rel_err = eps * releps;
      abs_err = eps * abseps;
#pragma unroll
      for (int c = 0; c < C; c++) YSAVE(c) = yy[c];  // y = yy
      t = x;
      const double total_error = fabs(rel_err) + fabs(abs_err);
      if (total_error > P.sg_error_limit) {
        stop = RAYS_STOP_ODE_TOTAL_ERROR;
        seg = SEG_STOP;
      } else {
        have_f = 0;
        seg = SEG_DE_BEGIN;
      }
    }

    if (seg == SEG_DE_BEGIN) {  // de parameter tests + restart (:423-505)
      if (t == tout) {
        stop = RAYS_STOP_SG_T_EQ_TOUT;
        seg = SEG_STOP;
      } else if (rel_err < 0.0 || abs_err < 0.0) {
        stop = RAYS_STOP_SG_NEG_ERR;
        seg = SEG_STOP;
      } else {
        eps = fmax(rel_err, abs_err);
        if (eps <= 0.0) {
          stop = RAYS_STOP_SG_EPS_LE_0;
          seg = SEG_STOP;
        } else {
          const double del = tout - t;
          absdel = fabs(del);
          tend = t + 10.0 * del;  // :485
          nostep = 0;
          kle4 = 0;
          fl &= ~FL_STIFF;
          releps = rel_err / eps;
          abseps = abs_err / eps;
          fl |= FL_START;  // :497-505
          x = t;
          h = copysign(fmax(fabs(tout - x), fouru * fabs(x)), tout - x);
          seg = SEG_DE_TOP;
        }
      }
    }

    // ---- DE_TOP: interval complete (intrp), or the entry of `step` (:511-556, 833-885) ------------------------------
    SG_PROF(7);  // CRASH + DE_BEGIN
    {
      const bool top = seg == SEG_DE_TOP;
      const bool finish = top && absdel <= fabs(x - t);
      const bool enter = top && !finish && !(maxnum <= nostep);
      if (RAYS_RARE(top && !finish && maxnum <= nostep)) {  // :536-548
        stop = (fl & FL_STIFF) ? RAYS_STOP_SG_STIFF : RAYS_STOP_SG_MAXNUM;
#pragma unroll
        for (int c = 0; c < C; c++) YSAVE(c) = yy[c];
        t = x;
        seg = SEG_STOP;
      }
      if (finish) {
        // ---- intrp (:1235-1362) -> y(tout) ----
        const double hi = tout - x;
        const int ki = kold + 1;
        for (int i = 1; i <= ki; i++) S.wi(i) = 1.0 / (double)i;
        S.gi(1) = 1.0;
        double term = 0.0;
        for (int j = 2; j <= ki; j++) {
          const double psijm1 = S.psi(j - 1);
          const Recip rpsi = make_recip(psijm1);
          const double gamma = div(hi + term, rpsi);
          const double eta = div(hi, rpsi);
          for (int i = 1; i <= ki + 1 - j; i++) S.wi(i) = gamma * S.wi(i) - eta * S.wi(i + 1);
          S.gi(j) = S.wi(1);
          term = psijm1;
        }
        double yout[C];
#pragma unroll
        for (int c = 0; c < C; c++) yout[c] = 0.0;
        F.interp(ki, S, yout, any_hi);
#pragma unroll
        for (int c = 0; c < C; c++) {
          yy[c] = yy[c] + hi * yout[c];
          YSAVE(c) = yy[c];
        }
        t = tout;
        pc = PC_CHECK;
        seg = SEG_WAIT;
      }
      if (__any(enter)) {  // wave-uniform
        double tq[C];
        if (enter) {
          h = copysign(fmin(fabs(h), fabs(tend - x)), h);  // :552-553
#pragma unroll
          for (int c = 0; c < C; c++) wt[c] = make_recip(releps * fabs(yy[c]) + abseps);
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
          const double q = div(yy[c], wt[c]);
          tq[c] = q * q;
        }
        const double sm = grp_sum<G>(tq);
        if (enter) {
          if (fabs(h) < fouru * fabs(x)) {
            h = copysign(fouru * fabs(x), h);
            seg = SEG_CRASH;
          } else {
            p5eps = 0.5 * eps;
            round_ = twou * sqrt(sm);  // :844
            if (p5eps < round_) {
              eps = 2.0 * round_ * (1.0 + fouru);
              seg = SEG_CRASH;
            } else {
              S.g(1) = 1.0;
              S.g(2) = 0.5;
              S.sig(1) = 1.0;
              if (fl & FL_START) {
                if (have_f) {
                  seg = SEG_START_DONE;
                } else {  // f(x, yy) needed (:860)
                  pc = PC_F1;
                  seg = SEG_WAIT;
                }
              } else {
                ifail = 0;
                seg = SEG_COEF;
              }
            }
          }
        }
      }
    }

    // ---- START_DONE: first step of an interval (:863-885) ------------------------------------------------------------
    SG_PROF(8);
    {
      const bool here = seg == SEG_START_DONE && !code;
      if (RAYS_RARE(seg == SEG_START_DONE && code)) {  // :863
        stop = code;
        seg = SEG_STOP;
      }
      if (__any(here)) {  // wave-uniform
        double tq[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
          const double q = div(f[c], wt[c]);
          tq[c] = q * q;
        }
        const double sm = grp_sum<G>(tq);
        if (here) {
          have_f = 0;
#pragma unroll
          for (int c = 0; c < C; c++) {
            F.lo[1][c] = f[c];
            F.lo[2][c] = 0.0;
          }
          const double total = sqrt(sm);
          absh = fabs(h);
          if (eps < 16.0 * total * h * h) absh = 0.25 * sqrt(eps / total);
          h = copysign(fmax(absh, fouru * fabs(x)), h);
          hold = 0.0;
          k = 1;
          kold = 0;
          fl &= ~FL_START;
          fl |= FL_PHASE1;
          fl |= FL_NORND;
          if (p5eps <= 100.0 * round_) {
            fl &= ~FL_NORND;
#pragma unroll
            for (int c = 0; c < C; c++) F.r15[c] = 0.0;
          }
          ifail = 0;
          seg = SEG_COEF;
        }
      }
    }

    // ---- COEF: coefficients + predictor (:892-1015) ------------------------------------------------------------------
    SG_PROF(9);
    if (seg == SEG_COEF) {
      const int kp1 = k + 1, kp2 = k + 2;
      if (h != hold) ns = 0;
      if (ns <= kold) ns = ns + 1;
      const int nsp1 = ns + 1;
      if (ns <= k) {
        // (order and register carries as in rays_sg.hpp: the v / w block first, then the recurrence psi -> beta -> alpha
        //  -> sig and g(i+1) from alpha(i) as ONE loop; every operation and its order are the reference's)
        const double alpha_ns = 1.0 / (double)ns;
        S.beta(ns) = 1.0;
        S.alpha(ns) = alpha_ns;
        double temp1 = h * (double)ns;
        S.sig(nsp1) = 1.0;
        double w[14];
#pragma unroll
        for (int iq = 0; iq < 14; iq++) w[iq] = 0.;
        if (ns <= 1) {
#pragma unroll
          for (int iq = 1; iq <= 12; iq++) {
            if (iq > kcap) continue;
            if (iq <= k) {
              const double cq = 1.0 / (double)(iq * (iq + 1));
              S.v(iq) = cq;
              w[iq] = cq;
            }
          }
        } else {
          if (kold < k) {
            S.v(k) = 1.0 / (double)(k * kp1);
            for (int j = 1; j <= ns - 2; j++) {
              const int i = k - j;
              S.v(i) = S.v(i) - S.alpha(j + 1) * S.v(i + 1);
            }
          }
          const int lim = kp1 - ns;
          double v_cur = S.v(1);
#pragma unroll
          for (int iq = 1; iq <= 12; iq++) {  // ascending: v(iq+1) is still the old value
            if (iq > kcap) continue;
            if (iq <= lim) {
              const double v_nxt = S.v(iq + 1);
              const double cq = v_cur - alpha_ns * v_nxt;
              S.v(iq) = cq;
              w[iq] = cq;
              v_cur = v_nxt;
            }
          }
          S.g(nsp1) = w[1];
        }
        double beta_c = 1.0, sig_c = 1.0;
        for (int i = nsp1; i <= k; i++) {
          const double temp2 = S.psi(i - 1);
          S.psi(i - 1) = temp1;
          beta_c = beta_c * temp1 / temp2;  // beta(i) = beta(i-1)*psi(i-1)/temp2
          S.beta(i) = beta_c;
          temp1 = temp2 + h;
          const double alpha_i = h / temp1;
          S.alpha(i) = alpha_i;
          sig_c = (double)i * alpha_i * sig_c;  // sig(i+1) = i*alpha(i)*sig(i)
          S.sig(i + 1) = sig_c;
          const int lim = kp1 - i;  // g(i+1): w(iq) = w(iq) - alpha(i)*w(iq+1), iq = 1..kp2-(i+1)
#pragma unroll
          for (int iq = 1; iq <= 12; iq++) {
            if (iq > kcap) continue;
            if (iq <= lim) w[iq] = w[iq] - alpha_i * w[iq + 1];
          }
          S.g(i + 1) = w[1];
        }
        S.psi(k) = temp1;
      }
      SG_PROF(12);  // coefficient block
      F.scale(nsp1, k, S, ncap, any_hi);
      {
        double row[C];
        F.get(kp1, row, any_hi);
        F.set(kp2, row, any_hi);  // phi(:,kp2) = phi(:,kp1)
#pragma unroll
        for (int c = 0; c < C; c++) {
          row[c] = 0.0;
          pp[c] = 0.0;
        }
        F.set(kp1, row, any_hi);  // phi(:,kp1) = 0
      }
      SG_PROF(13);  // scale + shift
      F.predict(k, S, pp, ncap, any_hi);
      SG_PROF(14);  // predictor
      if (!(fl & FL_NORND)) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          const double tau = h * pp[c] - F.r15[c];
          pp[c] = yy[c] + tau;
          F.r16[c] = (pp[c] - yy[c]) - tau;
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; c++) pp[c] = yy[c] + h * pp[c];
      }
      xold = x;
      x = x + h;
      absh = fabs(h);
      pc = PC_F2;
      seg = SEG_WAIT;
    }

    SG_PROF(10);  // COEF tail
    if (RAYS_RARE(seg == SEG_STOP)) {
      done = 1;
      seg = SEG_WAIT;
    }
    if (RAYS_RARE(seg != SEG_WAIT)) resume = seg;  // SEG_CRASH entered from DE_TOP: next trip

    if (RAYS_RARE(done && stop >= 0)) {  // ray_tracing.f90:252-260
      const TraceArgs& A = cold_args(A_hot);
      if (A.end_ray_vec) {
#pragma unroll
        for (int c = 0; c < C; c++)
          if (valid[c]) A.end_ray_vec[(long long)ray * NV + gl + G * c] = YSAVE(c);
      }
      if (gl == 0) {
        A.npoints[ray] = nstep + 1;
        A.stop_code[ray] = stop;
        if (A.end_residuals) A.end_residuals[ray] = nstep >= 1 ? prev_resid : 0.;
        if (A.max_residuals) A.max_residuals[ray] = maxr;
      }
    }

    SG_PROF(11);
    // ---- refill finished groups --------------------------------------------------------------------------------------
    if (__any(done)) {  // wave-uniform
      int nxt = 0;
      if (done && gl == 0) nxt = (int)(atomicAdd(cold_args(A_hot).next_ray, 1u) + total_groups);
      nxt = grp_bcast<G>(nxt, 0);
      if (done) {
        if ((unsigned)nxt < (unsigned)A_hot.nray) {
          ray = nxt;
          need_init = true;
        } else {
          alive = false;
        }
      }
    }
  }
  SG_PROF_FLUSH
#ifndef RAYS_HOST_EMUL
#undef gstr
#endif
#undef t
#undef tout
#undef absdel
#undef tend
#undef releps
#undef abseps
#undef sout
#undef ds_ray
#undef last_resid
#undef prev_resid
#undef maxr
#undef rel_err
#undef abs_err
#undef hold
#undef xold
#undef round_
#undef nstep
#undef kold
#undef ns
#undef knew
#undef ifail
#undef nostep
#undef kle4
#undef p5eps
#undef absh
#undef erk
#undef erkm1
#undef YSAVE
}

}  // namespace rays
