// rays_sg.hpp -- Shampine-Gordon ray-trace kernel (ode_solver_name = 'SG_ODE').
//
// Reference path restated here:
//   trace_rays        ray_tracing.f90:67-264
//   SG_ode            SG_ode_m.f90:89-159      (iflag = 1 every call, tolerance-inflation retry loop)
//   ode/de/step/intrp ode_RAYS.f90:1-230 / 232-593 / 595-1234 / 1235-1362
//                     (Burkardt's F90 of Shampine & Gordon's variable-order Adams PECE code)
//
// MI355X design.  `step` calls the RHS at three places (start, predictor, corrector).  Run as
// written, lanes that sit at different call sites would serialise three inlined copies of the
// (expensive) RHS.  Here the integrator is a per-lane coroutine: every trip of the wave loop does
// ONE convergent rhs_eval for all live lanes, then each lane runs the cheap continuation of its
// own integrator (segments SEG_*) up to its next RHS request.  Divided differences phi(nv,16) and
// the coefficient vectors are per-lane private arrays with data-dependent indices (order k is per
// lane), i.e. scratch memory; the ODE vectors y, p, yp, wt are register-resident.
//
// Because SG_ode restarts the integrator on every output interval (fresh work arrays, iflag = 1),
// the first RHS of `step` (start = true) is evaluated at the state check_save just saw, so it is
// fused with check_save exactly as in the RK4 kernel (rhs_eval with do_check).
#pragma once

#include "rays_trace.hpp"

namespace rays {

enum : int {
  PC_CHECK = 0,  // rhs at the recorded state: check_save + start-of-interval f
  PC_F1 = 1,     // start f after a crash/restart inside an interval (ode_RAYS.f90:860)
  PC_F2 = 2,     // predictor evaluation (:1017)
  PC_F3 = 3      // corrector evaluation (:1142)
};

enum : unsigned { FL_START = 1u, FL_PHASE1 = 2u, FL_NORND = 4u, FL_STIFF = 8u, FL_FIRST = 16u };

enum : int {
  SEG_WAIT = 0,
  SEG_DE_BEGIN,    // ode/de entry for interval [t, tout]
  SEG_DE_TOP,      // top of de's step loop
  SEG_START_DONE,  // have f(x, yy) for start = true
  SEG_COEF,        // coefficients + predictor
  SEG_AFTER_F2,
  SEG_AFTER_F3,
  SEG_CRASH,       // iflag = 3 return to SG_ode
  SEG_STOP         // ray finished with code `stop`
};

template <int NV>
struct SgLane {
  double phi[NV][17];
  double psi[13], alpha[13], beta[13], sig[14], v[13], w[13], g[14];
};

// gstr(1:13) -- single-precision literals widened to double (ode_RAYS.f90:776-779)
__device__ static const double kGstr[14] = {
    0., (double)0.50e+00f, (double)0.0833e+00f, (double)0.0417e+00f, (double)0.0264e+00f,
    (double)0.0188e+00f, (double)0.0143e+00f, (double)0.0114e+00f, (double)0.00936e+00f,
    (double)0.00789e+00f, (double)0.00679e+00f, (double)0.00592e+00f, (double)0.00524e+00f,
    (double)0.00468e+00f};

template <int EQ, int NS, int DERIV, int NV, int K>
__global__ void __launch_bounds__(256)
sg_trace_kernel(const DevParams P, const TraceArgs A_hot) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  PointStage<NV, K> stage;
  stage.base = lds + wave * PointStage<NV, K>::kDoublesPerWave;
  stage.lane = lane;

  const unsigned total_lanes = gridDim.x * blockDim.x;
  const long long npt = (long long)P.nstep_max + 1;
  constexpr double kEps = 2.220446049250313e-16;  // epsilon(1._rkind)
  constexpr double twou = 2.0 * kEps, fouru = 2.0 * twou;
  constexpr int maxnum = 500;  // ode_RAYS.f90:395

  // ---- per-lane ray state -------------------------------------------------------------------
  int ray = blockIdx.x * blockDim.x + threadIdx.x;
  bool alive = ray < A_hot.nray;
  bool need_init = alive;
  int pc = PC_CHECK;
  int nstep = 0;
  double sout = 0.;
  double vst[NV];   // v: the ray state at the last completed output point (y of SG_ode)
  double win[NV];   // RHS input of the current trip
  double last_resid = 0., prev_resid = 0., maxr = -1.7976931348623157e308;
  int nbuf = 0;
  long long first_pt = 0;

  // ---- per-lane integrator state (de / step locals that live across RHS evaluations) ----------
  SgLane<NV> S;
  double yy[NV], wt[NV], pp[NV], yp[NV];
  double t = 0., tout = 0., x = 0., h = 0., hold = 0., eps = 0.;
  double rel_err = 0., abs_err = 0., releps = 0., abseps = 0., absdel = 0., tend = 0.;
  double p5eps = 0., round_ = 0., xold = 0., absh = 0., erk = 0., erkm1 = 0.;
  int k = 1, kold = 0, ns = 0, knew = 1, ifail = 0, nostep = 0, kle4 = 0;
  // Per-lane logicals are kept as bits of ONE integer VGPR rather than as `bool`s: a bool that is
  // live across the divergent continuation loop is a 64-bit lane mask in SGPRs, and with ~10 of
  // them the register allocator spills lane masks inside divergent control flow.
  unsigned fl = FL_START | FL_PHASE1 | FL_NORND | FL_FIRST;
#pragma unroll
  for (int i = 0; i < NV; i++) vst[i] = win[i] = yy[i] = wt[i] = pp[i] = yp[i] = 0.;

  while (__any(alive)) {
    if (need_init) {  // ray_tracing.f90:77-93, SG_ode_m.f90:73-85
      const TraceArgs& A = cold_args(A_hot);  // rays_trace.hpp
      initialize_ode_vector<EQ, NS, NV>(P, A.rvec0 + 3ll * ray, A.rindex_vec0 + 3ll * ray, vst);
#pragma unroll
      for (int i = 0; i < NV; i++) win[i] = vst[i];
      pc = PC_CHECK;
      fl |= FL_FIRST;
      nstep = 0;
      sout = 0.;
      t = 0.;
      last_resid = 0.;
      prev_resid = 0.;
      maxr = -1.7976931348623157e308;
      rel_err = P.rel_err0;
      abs_err = P.abs_err0;
      need_init = false;
    }

    // ---- the one RHS evaluation of this trip -------------------------------------------------
    double f[NV], resid = 0.;
    int code = 0, cs_flag = 0;
    bool cs_stop = false;
    if (alive) rhs_eval<EQ, NS, DERIV, NV>(P, win, pc == PC_CHECK, resid, cs_flag, cs_stop, code, f);

    // ---- per-lane continuation -------------------------------------------------------------------
    int stop = 0;
    int done = 0;
#ifdef RAYS_SG_DEBUG
    double dbg[7] = {-1, -1, -1, -1, -1, -1, -1};
#endif
    if (alive) {
      int seg;
      int have_f = 0;  // f(x, yy) for start = true is already in f[]
      if (pc == PC_CHECK) {
        seg = SEG_DE_BEGIN;
        if (fl & FL_FIRST) {  // ray_tracing.f90:92-112
          if (nbuf == 0) first_pt = (long long)ray * npt;
          stage.put(nbuf, vst, 0.);
          nbuf++;
          fl &= ~FL_FIRST;
          if (cs_stop) {
            const TraceArgs& A = cold_args(A_hot);
            A.npoints[ray] = 1;
            A.stop_code[ray] = cs_flag;
            if (A.end_ray_vec)
#pragma unroll
              for (int i = 0; i < NV; i++) A.end_ray_vec[(long long)ray * NV + i] = 0.;
            if (A.end_residuals) A.end_residuals[ray] = 0.;
            if (A.max_residuals) A.max_residuals[ray] = 0.;
            done = 1;
            stop = -1;
            seg = SEG_WAIT;
          }
        } else {
          // the interval completed: win is y(tout)  (ray_tracing.f90:212-243)
#pragma unroll
          for (int i = 0; i < NV; i++) vst[i] = win[i];
          if (cs_stop) {
            stop = cs_flag;
            seg = SEG_STOP;
          } else {
            nstep = nstep + 1;
            if (nbuf == 0) first_pt = (long long)ray * npt + nstep;
            stage.put(nbuf, vst, resid);
            nbuf++;
            if (fabs(last_resid) > maxr) maxr = fabs(last_resid);
            prev_resid = last_resid;
            last_resid = resid;
          }
        }
        if (seg == SEG_DE_BEGIN) {  // ray_tracing.f90:118-172
          t = sout;  // s = sout
          sout = sout + P.ds;
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

      while (seg != SEG_WAIT) {
        if (seg == SEG_STOP) {
          done = 1;
#ifdef RAYS_SG_DEBUG
          dbg[0] = (double)stop; dbg[1] = (double)pc; dbg[2] = (double)nostep; dbg[3] = (double)k;
          dbg[4] = (double)code; dbg[5] = t; dbg[6] = tout;
#endif
          seg = SEG_WAIT;
        } else if (seg == SEG_DE_BEGIN) {
          // ---- de parameter tests + restart (ode_RAYS.f90:423-505); y == vst, t, tout set ----
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
#pragma unroll
              for (int i = 0; i < NV; i++) yy[i] = vst[i];
              h = copysign(fmax(fabs(tout - x), fouru * fabs(x)), tout - x);
              seg = SEG_DE_TOP;
            }
          }
        } else if (seg == SEG_DE_TOP) {
          if (absdel <= fabs(x - t)) {
            // ---- intrp (ode_RAYS.f90:1235-1362) -> y(tout); interval done (:511-518) ----
            double gi[14], rho[14], wi[15];
            const double hi = tout - x;
            const int ki = kold + 1;
            for (int i = 1; i <= ki; i++) wi[i] = 1.0 / (double)i;
            gi[1] = 1.0;
            rho[1] = 1.0;
            double term = 0.0;
            for (int j = 2; j <= ki; j++) {
              const double psijm1 = S.psi[j - 1];
              const double gamma = (hi + term) / psijm1;
              const double eta = hi / psijm1;
              for (int i = 1; i <= ki + 1 - j; i++) wi[i] = gamma * wi[i] - eta * wi[i + 1];
              gi[j] = wi[1];
              rho[j] = gamma * rho[j - 1];
              term = psijm1;
            }
            double yout[NV];
#pragma unroll
            for (int l = 0; l < NV; l++) yout[l] = 0.0;
            for (int j = 1; j <= ki; j++) {
              const int i = ki + 1 - j;
              const double gg = gi[i];
#pragma unroll
              for (int l = 0; l < NV; l++) yout[l] = yout[l] + gg * S.phi[l][i];
            }
#pragma unroll
            for (int l = 0; l < NV; l++) win[l] = yy[l] + hi * yout[l];
            t = tout;
            pc = PC_CHECK;
            seg = SEG_WAIT;
          } else if (maxnum <= nostep) {  // :536-548
            stop = (fl & FL_STIFF) ? RAYS_STOP_SG_STIFF : RAYS_STOP_SG_MAXNUM;
#pragma unroll
            for (int i = 0; i < NV; i++) vst[i] = yy[i];  // y = yy; t = x
            t = x;
            seg = SEG_STOP;
          } else {
            h = copysign(fmin(fabs(h), fabs(tend - x)), h);  // :552-553
#pragma unroll
            for (int l = 0; l < NV; l++) wt[l] = releps * fabs(yy[l]) + abseps;
            // ---- step entry (ode_RAYS.f90:833-885) ----
            if (fabs(h) < fouru * fabs(x)) {
              h = copysign(fouru * fabs(x), h);
              seg = SEG_CRASH;
            } else {
              p5eps = 0.5 * eps;
              double sm = 0.;
#pragma unroll
              for (int l = 0; l < NV; l++) {
                const double q = yy[l] / wt[l];
                sm += q * q;
              }
              round_ = twou * sqrt(sm);  // :844
              if (p5eps < round_) {
                eps = 2.0 * round_ * (1.0 + fouru);
                seg = SEG_CRASH;
              } else {
                S.g[1] = 1.0;
                S.g[2] = 0.5;
                S.sig[1] = 1.0;
                if (fl & FL_START) {
                  if (have_f) {
                    seg = SEG_START_DONE;
                  } else {  // f(x, yy) needed (:860)
#pragma unroll
                    for (int l = 0; l < NV; l++) win[l] = yy[l];
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
        } else if (seg == SEG_START_DONE) {
          have_f = 0;
          if (code) {  // :863 stop inside f: y, t untouched
            stop = code;
            seg = SEG_STOP;
          } else {  // :865-885
            double sm = 0.;
#pragma unroll
            for (int l = 0; l < NV; l++) {
              yp[l] = f[l];
              S.phi[l][1] = f[l];
              S.phi[l][2] = 0.0;
              const double q = f[l] / wt[l];
              sm += q * q;
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
              for (int l = 0; l < NV; l++) S.phi[l][15] = 0.0;
            }
            ifail = 0;
            seg = SEG_COEF;
          }
        } else if (seg == SEG_COEF) {
          // ---- coefficients + predictor (ode_RAYS.f90:892-1015) ----
          const int kp1 = k + 1, kp2 = k + 2;
          if (h != hold) ns = 0;
          if (ns <= kold) ns = ns + 1;
          const int nsp1 = ns + 1;
          if (ns <= k) {
            S.beta[ns] = 1.0;
            S.alpha[ns] = 1.0 / (double)ns;
            double temp1 = h * (double)ns;
            S.sig[nsp1] = 1.0;
            for (int i = nsp1; i <= k; i++) {
              const double temp2 = S.psi[i - 1];
              S.psi[i - 1] = temp1;
              S.beta[i] = S.beta[i - 1] * S.psi[i - 1] / temp2;
              temp1 = temp2 + h;
              S.alpha[i] = h / temp1;
              S.sig[i + 1] = (double)i * S.alpha[i] * S.sig[i];
            }
            S.psi[k] = temp1;
            if (ns <= 1) {
              for (int iq = 1; iq <= k; iq++) {
                S.v[iq] = 1.0 / (double)(iq * (iq + 1));
                S.w[iq] = S.v[iq];
              }
            } else {
              if (kold < k) {
                S.v[k] = 1.0 / (double)(k * kp1);
                for (int j = 1; j <= ns - 2; j++) {
                  const int i = k - j;
                  S.v[i] = S.v[i] - S.alpha[j + 1] * S.v[i + 1];
                }
              }
              for (int iq = 1; iq <= kp1 - ns; iq++) {
                S.v[iq] = S.v[iq] - S.alpha[ns] * S.v[iq + 1];
                S.w[iq] = S.v[iq];
              }
              S.g[nsp1] = S.w[1];
            }
            for (int i = ns + 2; i <= kp1; i++) {
              for (int iq = 1; iq <= kp2 - i; iq++) S.w[iq] = S.w[iq] - S.alpha[i - 1] * S.w[iq + 1];
              S.g[i] = S.w[1];
            }
          }
          for (int i = nsp1; i <= k; i++) {
            const double b = S.beta[i];
#pragma unroll
            for (int l = 0; l < NV; l++) S.phi[l][i] = b * S.phi[l][i];
          }
#pragma unroll
          for (int l = 0; l < NV; l++) {
            S.phi[l][kp2] = S.phi[l][kp1];
            S.phi[l][kp1] = 0.0;
            pp[l] = 0.0;
          }
          for (int j = 1; j <= k; j++) {
            const int i = kp1 - j;
            const double gg = S.g[i];
#pragma unroll
            for (int l = 0; l < NV; l++) {
              pp[l] = pp[l] + S.phi[l][i] * gg;
              S.phi[l][i] = S.phi[l][i] + S.phi[l][i + 1];
            }
          }
          if (!(fl & FL_NORND)) {
#pragma unroll
            for (int l = 0; l < NV; l++) {
              const double tau = h * pp[l] - S.phi[l][15];
              pp[l] = yy[l] + tau;
              S.phi[l][16] = (pp[l] - yy[l]) - tau;
            }
          } else {
#pragma unroll
            for (int l = 0; l < NV; l++) pp[l] = yy[l] + h * pp[l];
          }
          xold = x;
          x = x + h;
          absh = fabs(h);
#pragma unroll
          for (int l = 0; l < NV; l++) win[l] = pp[l];
          pc = PC_F2;
          seg = SEG_WAIT;
        } else if (seg == SEG_AFTER_F2) {
          if (code) {  // :1020
            stop = code;
            seg = SEG_STOP;
          } else {
            // ---- error estimates (ode_RAYS.f90:1026-1070) ----
            const int kp1 = k + 1, km1 = k - 1, km2 = k - 2;
            double erkm2 = 0.0;
            erkm1 = 0.0;
            erk = 0.0;
#pragma unroll
            for (int l = 0; l < NV; l++) {
              yp[l] = f[l];
              const double ph1 = S.phi[l][1];
              if (0 < km2) {
                const double q = (S.phi[l][km1] + yp[l] - ph1) / wt[l];
                erkm2 = erkm2 + q * q;
              }
              if (0 <= km2) {
                const double q = (S.phi[l][k] + yp[l] - ph1) / wt[l];
                erkm1 = erkm1 + q * q;
              }
              const double q = (yp[l] - ph1) / wt[l];
              erk = erk + q * q;
            }
            if (0 < km2) erkm2 = absh * S.sig[km1] * kGstr[km2] * sqrt(erkm2);
            if (0 <= km2) erkm1 = absh * S.sig[k] * kGstr[km1] * sqrt(erkm1);
            const double err = absh * sqrt(erk) * (S.g[k] - S.g[kp1]);
            erk = absh * sqrt(erk) * S.sig[kp1] * kGstr[k];
            knew = k;
            if (0 < km2) {
              if (fmax(erkm1, erkm2) <= erk) knew = km1;
            } else if (0 == km2) {
              if (erkm1 <= 0.5 * erk) knew = km1;
            }
            if (err <= eps) {
              // ---- successful: correct (ode_RAYS.f90:1128-1142) ----
              kold = k;
              hold = h;
              const double hg = h * S.g[kp1];
              if (!(fl & FL_NORND)) {
#pragma unroll
                for (int l = 0; l < NV; l++) {
                  const double rho = hg * (yp[l] - S.phi[l][1]) - S.phi[l][16];
                  yy[l] = pp[l] + rho;
                  S.phi[l][15] = (yy[l] - pp[l]) - rho;
                }
              } else {
#pragma unroll
                for (int l = 0; l < NV; l++) yy[l] = pp[l] + hg * (yp[l] - S.phi[l][1]);
              }
#pragma unroll
              for (int l = 0; l < NV; l++) win[l] = yy[l];
              pc = PC_F3;
              seg = SEG_WAIT;
            } else {
              // ---- failed step: restore, shrink (ode_RAYS.f90:1086-1120) ----
              fl &= ~FL_PHASE1;
              x = xold;
              for (int i = 1; i <= k; i++) {
                const double b = S.beta[i];
#pragma unroll
                for (int l = 0; l < NV; l++) S.phi[l][i] = (S.phi[l][i] - S.phi[l][i + 1]) / b;
              }
              for (int i = 2; i <= k; i++) S.psi[i - 1] = S.psi[i] - h;
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
        } else if (seg == SEG_AFTER_F3) {
          if (code) {  // :1145
            stop = code;
            seg = SEG_STOP;
          } else {
            // ---- update differences, choose order and step (ode_RAYS.f90:1151-1231) ----
            const int kp1 = k + 1, kp2 = k + 2, km1 = k - 1;
#pragma unroll
            for (int l = 0; l < NV; l++) {
              yp[l] = f[l];
              S.phi[l][kp1] = yp[l] - S.phi[l][1];
              S.phi[l][kp2] = S.phi[l][kp1] - S.phi[l][kp2];
            }
            for (int i = 1; i <= k; i++) {
#pragma unroll
              for (int l = 0; l < NV; l++) S.phi[l][i] = S.phi[l][i] + S.phi[l][kp1];
            }
            double erkp1 = 0.0;
            if (knew == km1 || k == 12) fl &= ~FL_PHASE1;
            if (fl & FL_PHASE1) {
              k = kp1;
              erk = erkp1;
            } else if (knew == km1) {
              k = km1;
              erk = erkm1;
            } else if (kp1 <= ns) {
#pragma unroll
              for (int l = 0; l < NV; l++) {
                const double q = S.phi[l][kp2] / wt[l];
                erkp1 = erkp1 + q * q;
              }
              erkp1 = absh * kGstr[kp1] * sqrt(erkp1);
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
              const double two_k1 = (double)(2 << k);  // two(k+1) = 2**(k+1)
              if (p5eps < erk * two_k1) {
                hnew = h;
                if (p5eps < erk) {
                  const double temp2 = (double)(k + 1);
                  const double r = pow(p5eps / erk, 1.0 / temp2);
                  hnew = absh * fmax(0.5, fmin((double)0.9f, r));
                  hnew = copysign(fmax(hnew, fouru * fabs(x)), h);
                }
              }
            }
            h = hnew;
            // ---- back in de (ode_RAYS.f90:579-588) ----
            nostep = nostep + 1;
            kle4 = kle4 + 1;
            if (4 < kold) kle4 = 0;
            if (50 <= kle4) fl |= FL_STIFF;
            seg = SEG_DE_TOP;
          }
        } else {  // SEG_CRASH: de returns iflag = 3 (ode_RAYS.f90:566-575), SG_ode_m.f90:139-149
          rel_err = eps * releps;
          abs_err = eps * abseps;
#pragma unroll
          for (int i = 0; i < NV; i++) vst[i] = yy[i];  // y = yy
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
      }

      if (done && stop >= 0) {  // ray_tracing.f90:252-260
        const TraceArgs& A = cold_args(A_hot);
        A.npoints[ray] = nstep + 1;
        A.stop_code[ray] = stop;
        if (A.end_ray_vec)
#pragma unroll
          for (int i = 0; i < NV; i++) A.end_ray_vec[(long long)ray * NV + i] = vst[i];
        if (A.end_residuals) A.end_residuals[ray] = nstep >= 1 ? prev_resid : 0.;
        if (A.max_residuals) A.max_residuals[ray] = maxr;
#ifdef RAYS_SG_DEBUG
        for (int i = 0; i < 7; i++) A.end_ray_vec[(long long)ray * NV + i] = dbg[i];
#endif
      }
    }

    // ---- wave-level: flush staged points, refill finished lanes --------------------------------
    if (done) {
      stage.drain_own(A_hot, nbuf, first_pt);
      nbuf = 0;
    }
    if (__any(nbuf == K)) {
      stage.flush(A_hot, nbuf, first_pt);
      nbuf = 0;
    }
    if (done) {
      const TraceArgs& A = cold_args(A_hot);
      const unsigned nxt = atomicAdd(A.next_ray, 1u) + total_lanes;
      if (nxt < (unsigned)A.nray) {
        ray = (int)nxt;
        need_init = true;
      } else {
        alive = false;
      }
    }
  }
}

}  // namespace rays
