/*
 * rays_oracle_sg.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the Shampine-Gordon variable-order Adams PECE integrator as RAYS drives it:
 *     SG_ode          RAYS_project/RAYS_lib/SG_ode_m.f90:89-159
 *     ode/de/step/intrp   RAYS_project/RAYS_lib/ode_RAYS.f90:1-230 / 232-593 / 595-1234 / 1235-1362
 * (ode_RAYS.f90 is Burkardt's F90 version of Shampine & Gordon's ODE, LGPL, with ORNL's
 * ray_stop early returns.)  Algorithm reference: L. Shampine, M. Gordon, "Computer Solution of
 * Ordinary Differential Equations: The Initial Value Problem", Freeman 1975.
 *
 * SG_ode forces iflag = 1 on every call (SG_ode_m.f90:120) with fresh automatic work arrays
 * (:106-107), so `de` always takes the restart branch (ode_RAYS.f90:497-505) and isn = +1; the
 * iflag = -1 / continuation branches of `de` are unreachable from RAYS and are not restated.
 *
 * Arrays are 1-based like the Fortran (index 0 unused) so loop bounds read the same.
 * Single-precision literals that the reference widens to double are written as (double)<x>f:
 * gstr(1:13) (ode_RAYS.f90:776-779) and real(0.9) (:1223).
 */
#include <math.h>
#include <string.h>

#include "rays_oracle.h"

#define NV RAYS_ORACLE_NV_MAX
#define DBL_EPS 2.220446049250313e-16 /* epsilon(1.0_rkind) */

typedef struct {
  double phi[NV][17];
  double psi[13], alpha[13], beta[13], sig[14], v[13], w[13], g[14];
  double yy[NV], wt[NV], p[NV], yp[NV], ypout[NV];
  double x, h, hold, eps;
  int k, kold, ns;
  int start, phase1, nornd, crash;
} sg_work;

static const double gstr[14] = {0.,
  (double)0.50e+00f,    (double)0.0833e+00f,  (double)0.0417e+00f,  (double)0.0264e+00f,
  (double)0.0188e+00f,  (double)0.0143e+00f,  (double)0.0114e+00f,  (double)0.00936e+00f,
  (double)0.00789e+00f, (double)0.00679e+00f, (double)0.00592e+00f, (double)0.00524e+00f,
  (double)0.00468e+00f};
static const double two[14] = {0., 2., 4., 8., 16., 32., 64., 128., 256., 512., 1024., 2048.,
                               4096., 8192.};

static double wnorm(int neqn, const double* a, const double* wt) { /* sqrt(sum((a/wt)**2)) */
  double s = 0.;
  for (int l = 0; l < neqn; l++) {
    double t = a[l] / wt[l];
    s += t * t;
  }
  return sqrt(s);
}

/* step                     ode_RAYS.f90:595-1234.   Returns stop code from f (0 = none). */
static int sg_step(sg_work* W, int neqn, rays_oracle_rhs_fn f, void* ctx, int* nrhs) {
  double* y = W->yy;
  double* wt = W->wt;
  double* p = W->p;
  double* yp = W->yp;
  const double twou = 2.0 * DBL_EPS, fouru = 2.0 * twou;
  double absh, erk, erkm1, erkm2, erkp1, err, hnew, p5eps, r, rho, round, total, tau, temp1, temp2,
      xold;
  int i, ifail, iq, j, km1, km2, knew, kp1, kp2, l, nsp1, st;

  W->crash = 1; /* :833 */
  if (fabs(W->h) < fouru * fabs(W->x)) {
    W->h = copysign(fouru * fabs(W->x), W->h);
    return 0;
  }
  p5eps = 0.5 * W->eps;
  round = twou * wnorm(neqn, y, wt); /* :844 */
  if (p5eps < round) {
    W->eps = 2.0 * round * (1.0 + fouru);
    return 0;
  }
  W->crash = 0;
  W->g[1] = 1.0;
  W->g[2] = 0.5;
  W->sig[1] = 1.0;

  if (W->start) { /* :858-885 */
    st = f(ctx, y, yp);
    (*nrhs)++;
    if (st) return st;
    for (l = 0; l < neqn; l++) {
      W->phi[l][1] = yp[l];
      W->phi[l][2] = 0.0;
    }
    total = wnorm(neqn, yp, wt);
    absh = fabs(W->h);
    if (W->eps < 16.0 * total * W->h * W->h) absh = 0.25 * sqrt(W->eps / total);
    W->h = copysign(fmax(absh, fouru * fabs(W->x)), W->h);
    W->hold = 0.0;
    W->k = 1;
    W->kold = 0;
    W->start = 0;
    W->phase1 = 1;
    W->nornd = 1;
    if (p5eps <= 100.0 * round) {
      W->nornd = 0;
      for (l = 0; l < neqn; l++) W->phi[l][15] = 0.0;
    }
  }
  ifail = 0;

  for (;;) { /* :892 */
    const int k = W->k;
    const double h = W->h;
    kp1 = k + 1;
    kp2 = k + 2;
    km1 = k - 1;
    km2 = k - 2;
    if (h != W->hold) W->ns = 0;
    if (W->ns <= W->kold) W->ns = W->ns + 1;
    const int ns = W->ns;
    nsp1 = ns + 1;

    if (ns <= k) { /* :914-977 */
      W->beta[ns] = 1.0;
      W->alpha[ns] = 1.0 / (double)ns;
      temp1 = h * (double)ns;
      W->sig[nsp1] = 1.0;
      for (i = nsp1; i <= k; i++) {
        temp2 = W->psi[i - 1];
        W->psi[i - 1] = temp1;
        W->beta[i] = W->beta[i - 1] * W->psi[i - 1] / temp2;
        temp1 = temp2 + h;
        W->alpha[i] = h / temp1;
        W->sig[i + 1] = (double)i * W->alpha[i] * W->sig[i];
      }
      W->psi[k] = temp1;
      if (ns <= 1) {
        for (iq = 1; iq <= k; iq++) {
          W->v[iq] = 1.0 / (double)(iq * (iq + 1));
          W->w[iq] = W->v[iq];
        }
      } else {
        if (W->kold < k) {
          W->v[k] = 1.0 / (double)(k * kp1);
          for (j = 1; j <= ns - 2; j++) {
            i = k - j;
            W->v[i] = W->v[i] - W->alpha[j + 1] * W->v[i + 1];
          }
        }
        for (iq = 1; iq <= kp1 - ns; iq++) {
          W->v[iq] = W->v[iq] - W->alpha[ns] * W->v[iq + 1];
          W->w[iq] = W->v[iq];
        }
        W->g[nsp1] = W->w[1];
      }
      for (i = ns + 2; i <= kp1; i++) {
        for (iq = 1; iq <= kp2 - i; iq++) W->w[iq] = W->w[iq] - W->alpha[i - 1] * W->w[iq + 1];
        W->g[i] = W->w[1];
      }
    }
    /* :985-1001 predict */
    for (i = nsp1; i <= k; i++)
      for (l = 0; l < neqn; l++) W->phi[l][i] = W->beta[i] * W->phi[l][i];
    for (l = 0; l < neqn; l++) {
      W->phi[l][kp2] = W->phi[l][kp1];
      W->phi[l][kp1] = 0.0;
      p[l] = 0.0;
    }
    for (j = 1; j <= k; j++) {
      i = kp1 - j;
      for (l = 0; l < neqn; l++) {
        p[l] = p[l] + W->phi[l][i] * W->g[i];
        W->phi[l][i] = W->phi[l][i] + W->phi[l][i + 1];
      }
    }
    if (!W->nornd) { /* :1003-1011 */
      for (l = 0; l < neqn; l++) {
        tau = h * p[l] - W->phi[l][15];
        p[l] = y[l] + tau;
        W->phi[l][16] = (p[l] - y[l]) - tau;
      }
    } else {
      for (l = 0; l < neqn; l++) p[l] = y[l] + h * p[l];
    }
    xold = W->x;
    W->x = W->x + h;
    absh = fabs(h);
    st = f(ctx, p, yp); /* :1017 */
    (*nrhs)++;
    if (st) return st;

    erkm2 = 0.0; erkm1 = 0.0; erk = 0.0; /* :1026-1053 */
    for (l = 0; l < neqn; l++) {
      if (0 < km2) {
        double t = (W->phi[l][km1] + yp[l] - W->phi[l][1]) / wt[l];
        erkm2 = erkm2 + t * t;
      }
      if (0 <= km2) {
        double t = (W->phi[l][k] + yp[l] - W->phi[l][1]) / wt[l];
        erkm1 = erkm1 + t * t;
      }
      double t = (yp[l] - W->phi[l][1]) / wt[l];
      erk = erk + t * t;
    }
    if (0 < km2) erkm2 = absh * W->sig[km1] * gstr[km2] * sqrt(erkm2);
    if (0 <= km2) erkm1 = absh * W->sig[k] * gstr[km1] * sqrt(erkm1);
    err = absh * sqrt(erk) * (W->g[k] - W->g[kp1]);
    erk = absh * sqrt(erk) * W->sig[kp1] * gstr[k];
    knew = k;
    if (0 < km2) { /* :1058-1070 */
      if (fmax(erkm1, erkm2) <= erk) knew = km1;
    } else if (0 == km2) {
      if (erkm1 <= 0.5 * erk) knew = km1;
    }
    if (err <= W->eps) break; /* :1074 */

    /* :1086-1120 unsuccessful step */
    W->phase1 = 0;
    W->x = xold;
    for (i = 1; i <= k; i++)
      for (l = 0; l < neqn; l++) W->phi[l][i] = (W->phi[l][i] - W->phi[l][i + 1]) / W->beta[i];
    for (i = 2; i <= k; i++) W->psi[i - 1] = W->psi[i] - h;
    ifail = ifail + 1;
    temp2 = 0.5;
    if (3 < ifail) {
      if (p5eps < 0.25 * erk) temp2 = sqrt(p5eps / erk);
    }
    if (3 <= ifail) knew = 1;
    W->h = temp2 * h;
    W->k = knew;
    if (fabs(W->h) < fouru * fabs(W->x)) {
      W->crash = 1;
      W->h = copysign(fouru * fabs(W->x), W->h);
      W->eps = W->eps + W->eps;
      return 0;
    }
  }

  /* :1128-1231 successful step */
  {
    const int k = W->k;
    const double h = W->h;
    W->kold = k;
    W->hold = h;
    if (!W->nornd) {
      for (l = 0; l < neqn; l++) {
        rho = h * W->g[kp1] * (yp[l] - W->phi[l][1]) - W->phi[l][16];
        y[l] = p[l] + rho;
        W->phi[l][15] = (y[l] - p[l]) - rho;
      }
    } else {
      for (l = 0; l < neqn; l++) y[l] = p[l] + h * W->g[kp1] * (yp[l] - W->phi[l][1]);
    }
    st = f(ctx, y, yp); /* :1142 */
    (*nrhs)++;
    if (st) return st;
    for (l = 0; l < neqn; l++) {
      W->phi[l][kp1] = yp[l] - W->phi[l][1];
      W->phi[l][kp2] = W->phi[l][kp1] - W->phi[l][kp2];
    }
    for (i = 1; i <= k; i++)
      for (l = 0; l < neqn; l++) W->phi[l][i] = W->phi[l][i] + W->phi[l][kp1];

    erkp1 = 0.0;
    if (knew == km1 || k == 12) W->phase1 = 0;
    if (W->phase1) {
      W->k = kp1;
      erk = erkp1;
    } else if (knew == km1) {
      W->k = km1;
      erk = erkm1;
    } else if (kp1 <= W->ns) {
      for (l = 0; l < neqn; l++) {
        double t = W->phi[l][kp2] / wt[l];
        erkp1 = erkp1 + t * t;
      }
      erkp1 = absh * gstr[kp1] * sqrt(erkp1);
      if (k == 1) {
        if (erkp1 < 0.5 * erk) {
          W->k = kp1;
          erk = erkp1;
        }
      } else if (erkm1 <= fmin(erk, erkp1)) {
        W->k = km1;
        erk = erkm1;
      } else if (erkp1 < erk && k < 12) {
        W->k = kp1;
        erk = erkp1;
      }
    }
    hnew = h + h; /* :1212 */
    if (!W->phase1) {
      if (p5eps < erk * two[W->k + 1]) {
        hnew = h;
        if (p5eps < erk) {
          temp2 = (double)(W->k + 1);
          r = pow(p5eps / erk, 1.0 / temp2);
          hnew = absh * fmax(0.5, fmin((double)0.9f, r));
          hnew = copysign(fmax(hnew, fouru * fabs(W->x)), h);
        }
      }
    }
    W->h = hnew;
  }
  return 0;
}

/* intrp                    ode_RAYS.f90:1235-1362 */
static void sg_intrp(const sg_work* W, int neqn, double xout, double* yout, double* ypout) {
  double g[14], rho[14], w[15];
  const double hi = xout - W->x;
  const int ki = W->kold + 1;
  int i, j, l;
  for (i = 1; i <= ki; i++) w[i] = 1.0 / (double)i;
  g[1] = 1.0;
  rho[1] = 1.0;
  double term = 0.0;
  for (j = 2; j <= ki; j++) {
    double psijm1 = W->psi[j - 1];
    double gamma = (hi + term) / psijm1;
    double eta = hi / psijm1;
    for (i = 1; i <= ki + 1 - j; i++) w[i] = gamma * w[i] - eta * w[i + 1];
    g[j] = w[1];
    rho[j] = gamma * rho[j - 1];
    term = psijm1;
  }
  for (l = 0; l < neqn; l++) {
    ypout[l] = 0.0;
    yout[l] = 0.0;
  }
  for (j = 1; j <= ki; j++) {
    i = ki + 1 - j;
    for (l = 0; l < neqn; l++) {
      yout[l] = yout[l] + g[i] * W->phi[l][i];
      ypout[l] = ypout[l] + rho[i] * W->phi[l][i];
    }
  }
  for (l = 0; l < neqn; l++) yout[l] = W->yy[l] + hi * yout[l];
}

/* ode + de with iflag = 1   ode_RAYS.f90:1-230, 232-593.
 * Returns iflag (2,3,4,5,6); *stop = stop code set by f or by the parameter checks. */
static int sg_de(int neqn, rays_oracle_rhs_fn f, void* ctx, double* y, double* t, double tout,
                 double* relerr, double* abserr, int* stop, int* nrhs) {
  sg_work W;
  const double fouru = 4.0 * DBL_EPS;
  const int maxnum = 500;
  *stop = 0;
  if (*t == tout) { *stop = RAYS_STOP_SG_T_EQ_TOUT; return 6; }                      /* :431 */
  if (*relerr < 0.0 || *abserr < 0.0) { *stop = RAYS_STOP_SG_NEG_ERR; return 6; }    /* :437 */
  W.eps = fmax(*relerr, *abserr);
  if (W.eps <= 0.0) { *stop = RAYS_STOP_SG_EPS_LE_0; return 6; }                     /* :445 */
  const double del = tout - *t, absdel = fabs(del);
  const double tend = *t + 10.0 * del; /* :485 */
  int nostep = 0, kle4 = 0, stiff = 0;
  const double releps = *relerr / W.eps, abseps = *abserr / W.eps;
  /* :497-505 restart */
  W.start = 1;
  W.x = *t;
  for (int l = 0; l < neqn; l++) W.yy[l] = y[l];
  W.h = copysign(fmax(fabs(tout - W.x), fouru * fabs(W.x)), tout - W.x);
  W.k = 0; W.kold = 0; W.ns = 0; W.hold = 0.; W.phase1 = 0; W.nornd = 0; W.crash = 0;
  memset(W.psi, 0, sizeof W.psi);
  for (;;) {
    if (absdel <= fabs(W.x - *t)) { /* :511-518 */
      sg_intrp(&W, neqn, tout, y, W.ypout);
      *t = tout;
      return 2;
    }
    if (maxnum <= nostep) { /* :536-548 */
      *stop = stiff ? RAYS_STOP_SG_STIFF : RAYS_STOP_SG_MAXNUM;
      for (int l = 0; l < neqn; l++) y[l] = W.yy[l];
      *t = W.x;
      return stiff ? 5 : 4;
    }
    W.h = copysign(fmin(fabs(W.h), fabs(tend - W.x)), W.h); /* :552-553 */
    for (int l = 0; l < neqn; l++) W.wt[l] = releps * fabs(W.yy[l]) + abseps;
    int st = sg_step(&W, neqn, f, ctx, nrhs);
    if (st) { *stop = st; return 0; } /* :560 */
    if (W.crash) {                    /* :566-575 */
      *relerr = W.eps * releps;
      *abserr = W.eps * abseps;
      for (int l = 0; l < neqn; l++) y[l] = W.yy[l];
      *t = W.x;
      return 3;
    }
    nostep = nostep + 1;
    kle4 = kle4 + 1;
    if (4 < W.kold) kle4 = 0;
    if (50 <= kle4) stiff = 1;
  }
}

/* SG_ode                   SG_ode_m.f90:89-159.  Returns stop code (0 = reached sout). */
int rays_oracle_sg_ode(const rays_params_t* P, rays_oracle_rhs_fn f, void* ctx, double* v, double* s,
                       double* sout, double* rel_err, double* abs_err, int* nrhs) {
  for (;;) {
    int stop = 0;
    int iflag = sg_de(P->nv, f, ctx, v, s, *sout, rel_err, abs_err, &stop, nrhs);
    if (stop && iflag == 0) { /* stop_ode set inside eqn_ray (:132-135) */
      *sout = *s;
      return stop;
    }
    if (iflag == 2) return 0;
    if (iflag == 3) {
      double total_error = fabs(*rel_err) + fabs(*abs_err);
      if (total_error > P->SG_error_limit) return RAYS_STOP_ODE_TOTAL_ERROR; /* :141-147 */
      continue;
    }
    return stop; /* :150-155 error return: flag text was set in de */
  }
}
