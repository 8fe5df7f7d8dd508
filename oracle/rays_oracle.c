/*
 * rays_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * Plain-C CPU restatement of the ORNL-Fusion/RAYS per-ray ODE hot path, written to follow the
 * reference's *operation order* statement by statement so that it reproduces the reference
 * binary (oracle/_ref/rays_ref_dump, amdflang -O2 -ffp-contract=off) to the last bit wherever
 * libm is not involved.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product (rays_amd/, librays_hip.so) never does.
 *
 * Pinned by: the tests/golden fixtures, cut from dumps of the reference binary on configs/
 * (tests/golden/make_golden.py).  The reference's own tests hold no golden vectors
 * (SURVEY.md 4), so this is the only pin; see DESIGN.md "Oracle".
 *
 * All citations are RAYS_project/RAYS_lib/<file>:<line> under /root/reference.
 *
 * Compile:  gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 *           (NEVER -ffast-math / -Ofast: NaN polarity and rounding are part of the contract).
 *
 * Notes on how flang lowers the reference (checked in the disassembly of rays_ref_dump):
 *   x**2 -> x*x ; x**4 -> ((x*x)*x)*x (sequential, not squaring) ; x**y (real y) -> libm pow
 *   complex*complex -> (ac-bd, bc+ad) inline ; complex/real -> compiler-rt __divdc3 (logb
 *   scaling + textbook formula) ; abs(complex) -> libm cabs ; sum/product/dot_product ->
 *   sequential accumulation from the first element.
 */
#include "rays_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NS0 RAYS_NS0

/* --------------------------------------------------------------------------------------------
 * type eq_point            equilibrium_m.f90:39-59
 * gradbtensor[i][j] = d B(j) / d x(i)  (Fortran gradbtensor(i+1,j+1)); same for gradbunit.
 * err replaces character(len=60) equib_err ('' == 0).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  double bvec[3], bmag, gradbmag[3], bunit[3], gradbunit[3][3], gradbtensor[3][3];
  double ns[NS0], gradns[NS0][3];
  double ts[NS0], gradts[NS0][3];
  double omgc[NS0], omgp2[NS0], alpha[NS0], gamma[NS0];
  int err;
} eq_point;

/* rf_m module variables omgrf,k0 are rewritten by deriv_num (deriv_num.f90:72-84); here they are
 * an explicit per-call context so the restatement is thread-safe. */
typedef struct { double omgrf, k0; } rf_ctx;

static inline double sq(double x) { return x * x; }
static inline double pow4(double x) { return ((x * x) * x) * x; } /* flang: x**4 */

/* --------------------------------------------------------------------------------------------
 * parabolic_prof           slab_eq_m.f90:354-381
 * NB reference leaves fp undefined when rho >= 1 and f >= f_min; we return 0 (the caller then
 * forms alpha*gradns/ns = 0*fp/0 = NaN regardless of fp, deriv_cold.f90:64).
 * ------------------------------------------------------------------------------------------ */
static void parabolic_prof(double rho, double f_min, double alpha1, double alpha2, double* f,
                           double* fp) {
  *f = 0.0;
  *fp = 0.0;
  if (rho < 1.0) {
    *f = pow(1. - pow(rho, alpha2), alpha1);
    *fp = -alpha1 * alpha2 * pow(rho, alpha2 - 1.) * pow(1. - pow(rho, alpha2), alpha1 - 1.);
  }
  if (*f < f_min) {
    *f = f_min;
    *fp = 0.0;
  }
}

/* --------------------------------------------------------------------------------------------
 * slab_eq                  slab_eq_m.f90:125-309
 * check_box = 0 is used only where the reference reads an eq_point it left undefined
 * (check_save.f90:40-43 after an out-of-box return): see DESIGN.md "defined where the reference
 * is undefined".
 * ------------------------------------------------------------------------------------------ */
static int slab_eq(const rays_params_t* P, const double rvec[3], double bvec[3],
                   double gbt[3][3], double* ns, double (*gradns)[3], double* ts,
                   double (*gradts)[3], int check_box) {
  const rays_slab_params_t* S = &P->slab;
  const int nspec = P->nspec;
  int err = 0;
  double x = rvec[0], y = rvec[1], z = rvec[2];
  memset(gbt, 0, 9 * sizeof(double));
  for (int is = 0; is <= nspec; is++) {
    ns[is] = 0.;
    ts[is] = 0.;
    for (int i = 0; i < 3; i++) gradns[is][i] = gradts[is][i] = 0.;
  }
  /* :163-169 */
  if (x < S->xmin || x > S->xmax) err = RAYS_STOP_X_OUT_OF_BOUNDS;
  if (y < S->ymin || y > S->ymax) err = RAYS_STOP_Y_OUT_OF_BOUNDS;
  if (z < S->zmin || z > S->zmax) err = RAYS_STOP_Z_OUT_OF_BOUNDS;
  if (err && check_box) return err;
  err = 0;

  bvec[0] = 0.; /* :174 'zero' */
  switch (S->by_prof_model) { /* :184-206 */
    case RAYS_SLAB_BY_ZERO: bvec[1] = 0.; break;
    case RAYS_SLAB_BY_CONSTANT: bvec[1] = S->by0; break;
    case RAYS_SLAB_BY_TOROID:
      bvec[1] = S->by0 / (1. + x / S->rmaj);
      gbt[0][1] = -bvec[1] / (S->rmaj + x);
      break;
    case RAYS_SLAB_BY_LINEAR_SHEAR:
      bvec[1] = S->by0 * x / S->LBy_shear_scale;
      gbt[0][1] = S->by0 / S->LBy_shear_scale;
      break;
  }
  switch (S->bz_prof_model) { /* :209-233 */
    case RAYS_SLAB_BZ_CONSTANT: bvec[2] = S->bz0; break;
    case RAYS_SLAB_BZ_TOROID:
      bvec[2] = S->bz0 / (1. + x / S->rmaj);
      gbt[0][2] = -bvec[2] / (S->rmaj + x);
      break;
    case RAYS_SLAB_BZ_LINEAR:
      bvec[2] = S->bz0 * (1. + x / S->LBz_scale);
      gbt[0][2] = S->bz0 / S->LBz_scale;
      break;
    case RAYS_SLAB_BZ_LINEAR_2:
      bvec[2] = S->bz0 + S->dBzdx * (x - S->x0);
      gbt[0][2] = S->dBzdx;
      break;
  }
  switch (S->dens_prof_model) { /* :237-267 */
    case RAYS_SLAB_N_CONSTANT:
      for (int is = 0; is <= nspec; is++) ns[is] = P->n0s[is];
      break;
    case RAYS_SLAB_N_LINEAR:
      for (int is = 0; is <= nspec; is++) {
        ns[is] = P->n0s[is] * (1.0 + x / S->Ln_scale);
        gradns[is][0] = P->n0s[is] * (1.0 / S->Ln_scale);
      }
      break;
    case RAYS_SLAB_N_LINEAR_2: /* value uses dndx*eta, gradient n0s*dndx (:249-250) -- as is */
      for (int is = 0; is <= nspec; is++) {
        ns[is] = P->n0s[is] + S->dndx * P->eta[is] * (x - S->x0);
        gradns[is][0] = P->n0s[is] * S->dndx;
      }
      break;
    case RAYS_SLAB_N_PARABOLIC: {
      double f, fp;
      parabolic_prof(x, S->n_min, S->alphan1, S->alphan2, &f, &fp);
      for (int is = 0; is <= nspec; is++) {
        ns[is] = P->n0s[is] * f;
        gradns[is][0] = P->n0s[is] * fp;
      }
    } break;
    case RAYS_SLAB_N_GAUSSIAN:
      for (int is = 0; is <= nspec; is++) {
        ns[is] = P->n0s[is] * exp(-3. * S->alphan1 * sq(x / S->rmin));
        gradns[is][0] = ns[is] * (-6. * S->alphan1 * x / sq(S->rmin));
      }
      break;
  }
  for (int is = 0; is <= nspec; is++) { /* :270-301 */
    switch (S->t_prof_model[is]) {
      case RAYS_SLAB_T_ZERO: ts[is] = 0.; break;
      case RAYS_SLAB_T_CONSTANT: ts[is] = P->t0s[is]; break;
      case RAYS_SLAB_T_LINEAR:
        ts[is] = P->t0s[is] * (1. + x / S->LT_scale);
        gradts[is][0] = P->t0s[is] * (1. / S->LT_scale);
        break;
      case RAYS_SLAB_T_LINEAR_2:
        ts[is] = P->t0s[is] + S->dtdx * (x - S->x0);
        gradts[is][0] = P->t0s[is] * S->dtdx;
        break;
      case RAYS_SLAB_T_PARABOLIC: {
        double f, fp;
        parabolic_prof(x - S->x0, S->T_min[is], S->alphat1[is], S->alphat2[is], &f, &fp);
        ts[is] = P->t0s[is] * f;
        gradts[is][0] = P->t0s[is] * fp;
      } break;
    }
  }
  /* :305-306  minval(ns) < 0 */
  double mn = ns[0], mt = ts[0];
  for (int is = 1; is <= nspec; is++) {
    if (ns[is] < mn) mn = ns[is];
    if (ts[is] < mt) mt = ts[is];
  }
  if (mn < 0.) err = RAYS_STOP_NEGATIVE_DENS;
  if (mt < 0.) err = RAYS_STOP_NEGATIVE_TEMP;
  return check_box ? err : 0;
}

/* --------------------------------------------------------------------------------------------
 * solovev_psi              solovev_eq_m.f90:280-322
 * ------------------------------------------------------------------------------------------ */
static void solovev_psi(const rays_params_t* P, const double rvec[3], double* psi,
                        double gradpsi[3], double* psiN, double gradpsiN[3]) {
  const rays_solovev_params_t* S = &P->solovev;
  double x = rvec[0], y = rvec[1], z = rvec[2];
  double R = sqrt(x * x + y * y);
  double bp0 = S->bphi0 * S->iota0;
  *psi = .5 * bp0 * (sq(R * z / (S->rmaj * S->kappa)) + (sq(R * R - S->rmaj * S->rmaj)) / sq(S->rmaj) / 4.);
  double br = -bp0 * R * z / sq(S->rmaj * S->kappa);
  double bz = bp0 * (sq(z / (S->rmaj * S->kappa)) + .5 * (sq(R / S->rmaj) - 1.));
  gradpsi[0] = x * bz;
  gradpsi[1] = y * bz;
  gradpsi[2] = -R * br;
  *psiN = *psi / S->psiB;
  for (int i = 0; i < 3; i++) gradpsiN[i] = gradpsi[i] / S->psiB;
}

/* --------------------------------------------------------------------------------------------
 * solovev_eq               solovev_eq_m.f90:122-276
 * ------------------------------------------------------------------------------------------ */
static int solovev_eq(const rays_params_t* P, const double rvec[3], double bvec[3],
                      double gbt[3][3], double* ns, double (*gradns)[3], double* ts,
                      double (*gradts)[3], int check_box) {
  const rays_solovev_params_t* S = &P->solovev;
  const int nspec = P->nspec;
  int err = 0;
  double x = rvec[0], y = rvec[1], z = rvec[2];
  double r = sqrt(x * x + y * y);
  if (r < S->box_rmin || r > S->box_rmax) err = RAYS_STOP_R_OUT_OF_BOX; /* :155 */
  if (z < S->box_zmin || z > S->box_zmax) err = RAYS_STOP_Z_OUT_OF_BOX; /* :156 */
  double bp0 = S->bphi0 * S->iota0;
  double psi, gradpsi[3], psiN, gradpsiN[3];
  solovev_psi(P, rvec, &psi, gradpsi, &psiN, gradpsiN);
  if (err && check_box) return err; /* :164-167 */
  err = 0;

  /* :170-184 */
  double br = -bp0 * r * z / sq(S->rmaj * S->kappa);
  double bz = bp0 * (sq(z / (S->rmaj * S->kappa)) + .5 * (sq(r / S->rmaj) - 1.));
  double bphi = S->bphi0 * S->rmaj / r;
  double dbrdr = br / r;
  double dbrdz = -bp0 * r / sq(S->rmaj * S->kappa);
  double dbzdr = bp0 * r / sq(S->rmaj);
  double dbzdz = bp0 * 2. * z / sq(S->rmaj * S->kappa);
  double dbphidr = -bphi / r;
  /* :187-189 */
  bvec[0] = br * x / r - bphi * y / r;
  bvec[1] = br * y / r + bphi * x / r;
  bvec[2] = bz;
  /* :192-204  gradbtensor(i,j) -> gbt[i-1][j-1] */
  gbt[0][0] = (dbrdr * sq(x) + br * sq(y) / r + (-dbphidr + bphi / r) * x * y) / sq(r);
  gbt[1][0] = ((dbrdr - br / r) * x * y - dbphidr * sq(y) - bphi * sq(x) / r) / sq(r);
  gbt[2][0] = dbrdz * x / r;
  gbt[0][1] = ((dbrdr - br / r) * x * y + dbphidr * sq(x) + bphi * sq(y) / r) / sq(r);
  gbt[1][1] = (dbrdr * sq(y) + br * sq(x) / r + (dbphidr - bphi / r) * x * y) / sq(r);
  gbt[2][1] = dbrdz * y / r;
  gbt[0][2] = dbzdr * x / r;
  gbt[1][2] = dbzdr * y / r;
  gbt[2][2] = dbzdz;

  /* density :208-231 */
  if (S->dens_prof_model == RAYS_SOLOVEV_N_CONSTANT) {
    for (int is = 0; is <= nspec; is++) {
      ns[is] = P->n0s[is];
      gradns[is][0] = gradns[is][1] = gradns[is][2] = 0.;
    }
  } else { /* parabolic */
    for (int is = 0; is <= nspec; is++) {
      ns[is] = 0.;
      gradns[is][0] = gradns[is][1] = gradns[is][2] = 0.;
    }
    if (psiN < 1.0) {
      double a1 = S->alphan1, a2 = S->alphan2;
      double prof = pow(1. - pow(psiN, a2), a1);
      double dd_psi = -a1 * a2 * pow(psiN, a2 - 1.) * pow(1. - pow(psiN, a2), a1 - 1.);
      for (int is = 0; is <= nspec; is++) {
        ns[is] = P->n0s[is] * prof;
        gradns[is][0] = P->n0s[is] * dd_psi * gradpsiN[0];
        gradns[is][1] = P->n0s[is] * dd_psi * gradpsiN[1];
        gradns[is][2] = P->n0s[is] * dd_psi * gradpsiN[2];
      }
    }
  }
  /* temperature :235-268.  'parabolic' zeroes the WHOLE ts/gradts arrays inside the species
   * loop (:251-252) and uses exponent alphat1 (not alphat1-1) in the gradient (:256-257):
   * replicated, not fixed.  The 'constant' case (:243-245) leaves ts undefined in the
   * reference and is rejected by rays_oracle_check_params. */
  for (int is = 0; is <= nspec; is++) {
    if (S->t_prof_model[is] == RAYS_SOLOVEV_T_ZERO) {
      ts[is] = 0.;
      gradts[is][0] = gradts[is][1] = gradts[is][2] = 0.;
    } else {
      for (int j = 0; j <= nspec; j++) {
        ts[j] = 0.;
        gradts[j][0] = gradts[j][1] = gradts[j][2] = 0.;
      }
      if (psiN < 1.) {
        double a1 = S->alphat1[is], a2 = S->alphat2[is];
        ts[is] = P->t0s[is] * pow(1. - pow(psiN, a2), a1);
        double dd_psi = -a1 * a2 * pow(psiN, a2 - 1.) * pow(1. - pow(psiN, a2), a1);
        gradts[is][0] = P->t0s[is] * dd_psi * gradpsiN[0];
        gradts[is][1] = P->t0s[is] * dd_psi * gradpsiN[1];
        gradts[is][2] = P->t0s[is] * dd_psi * gradpsiN[2];
      }
    }
  }
  double mn = ns[0], mt = ts[0];
  for (int is = 1; is <= nspec; is++) {
    if (ns[is] < mn) mn = ns[is];
    if (ts[is] < mt) mt = ts[is];
  }
  if (mn < 0.) err = RAYS_STOP_NEGATIVE_DENS; /* :272 */
  if (mt < 0.) err = RAYS_STOP_NEGATIVE_TEMP; /* :273 */
  return check_box ? err : 0;
}

/* --------------------------------------------------------------------------------------------
 * PPPL-pspline evaluation on uniform grids (ilinx = iliny = 1):
 *   cspevx / bcspevxy   splines_lib/cspeval.f90:93-160, bcspeval.f90:128-255  (cell lookup)
 *   cspevfn / bcspevfn  cspeval.f90:205-250, bcspeval.f90:259-458              (Horner forms)
 * Out-of-range targets: the reference clamps within 4e-7*max|x| and otherwise returns ier = 1
 * leaving the outputs undefined; here the target is always clamped (defined behaviour, only
 * reachable for deriv_num's perturbed points, which skip the box test).
 * ------------------------------------------------------------------------------------------ */
static int spl_cell(const double* x, int nx, double xget, double* dx) {
  double z = xget;
  if (z < x[0]) z = x[0];
  if (z > x[nx - 1]) z = x[nx - 1];
  const int nxm = nx - 1;
  int ii = (int)(1 + nxm * (z - x[0]) / (x[nx - 1] - x[0]));
  int i = ii < nxm ? ii : nxm;
  if (i < 1) i = 1;
  if (z < x[i - 1]) i = i - 1;
  else if (z > x[i]) i = i + 1;
  if (i < 1) i = 1;
  if (i > nxm) i = nxm;
  *dx = z - x[i - 1];
  return i; /* 1-based cell index */
}

typedef struct {
  int nr, nz, n_rb, n_ne, n_te, n_ti;
  double *r_grid, *z_grid, *psi_fspl, *rb_grid, *rb_fspl, *ne_grid, *ne_fspl, *te_grid, *te_fspl,
      *ti_grid, *ti_fspl;
  int lin;       /* tables of 'eqdsk_magnetics_lin_interp': psi_fspl = Psi(nr, nz), rb_fspl = T(nr) */
  double dR, dZ; /* eqdsk_utilities_m: half the grid spacing */
} axisym_tables;
static axisym_tables AX;

static double* dup_d(const double* p, size_t n) {
  if (!p || !n) return NULL;
  double* q = (double*)malloc(n * sizeof(double));
  memcpy(q, p, n * sizeof(double));
  return q;
}

static int set_tables(const rays_axisym_tables_t* t, int lin, double dR, double dZ);
int rays_oracle_set_axisym_tables(const rays_axisym_tables_t* t) { return set_tables(t, 0, 0., 0.); }
/* same meaning as rays_hip_set_eqdsk_lin_tables */
int rays_oracle_set_eqdsk_lin_tables(const rays_axisym_tables_t* t, double dR, double dZ) { return set_tables(t, 1, dR, dZ); }
static int set_tables(const rays_axisym_tables_t* t, int lin, double dR, double dZ) {
  free(AX.r_grid); free(AX.z_grid); free(AX.psi_fspl); free(AX.rb_grid); free(AX.rb_fspl);
  free(AX.ne_grid); free(AX.ne_fspl); free(AX.te_grid); free(AX.te_fspl); free(AX.ti_grid); free(AX.ti_fspl);
  memset(&AX, 0, sizeof AX);
  AX.nr = t->nr; AX.nz = t->nz; AX.n_rb = t->n_rb; AX.n_ne = t->n_ne; AX.n_te = t->n_te; AX.n_ti = t->n_ti;
  AX.r_grid = dup_d(t->r_grid, t->nr); AX.z_grid = dup_d(t->z_grid, t->nz);
  AX.lin = lin; AX.dR = dR; AX.dZ = dZ;
  AX.psi_fspl = dup_d(t->psi_fspl, (size_t)(lin ? 1 : 16) * t->nr * t->nz);
  AX.rb_grid = lin ? NULL : dup_d(t->rb_grid, t->n_rb);
  AX.rb_fspl = dup_d(t->rb_fspl, (size_t)(lin ? 1 : 4) * t->n_rb);
  AX.ne_grid = dup_d(t->ne_grid, t->n_ne); AX.ne_fspl = dup_d(t->ne_fspl, (size_t)4 * t->n_ne);
  AX.te_grid = dup_d(t->te_grid, t->n_te); AX.te_fspl = dup_d(t->te_fspl, (size_t)4 * t->n_te);
  AX.ti_grid = dup_d(t->ti_grid, t->n_ti); AX.ti_fspl = dup_d(t->ti_fspl, (size_t)4 * t->n_ti);
  return 0;
}

/* eval_1D_fp: value and first derivative (quick_cube_splines_m.f90:105-121, cspevfn) */
static void spl1_fp(const double* grid, const double* fspl, int n, double x, double* f, double* fp) {
  double dx;
  const int i = spl_cell(grid, n, x, &dx);
  const double* c = fspl + 4 * (size_t)(i - 1);
  *f = c[0] + dx * (c[1] + dx * (c[2] + dx * c[3]));
  *fp = c[1] + dx * (2.0 * c[2] + dx * 3.0 * c[3]);
}

/* eval_2D_fpp (quick_cube_splines_m.f90:305-332): f, fx, fy, fxx, fxy, fyy from fspl(4,4,nx,ny) */
static void spl2_fpp(double x, double y, double out[6]) {
  double dx, dy;
  const int i = spl_cell(AX.r_grid, AX.nr, x, &dx);
  const int j = spl_cell(AX.z_grid, AX.nz, y, &dy);
  const double* c = AX.psi_fspl + 16 * ((size_t)(i - 1) + (size_t)AX.nr * (size_t)(j - 1));
#define F(a, b) c[((a) - 1) + 4 * ((b) - 1)]
  out[0] = F(1,1) + dy * (F(1,2) + dy * (F(1,3) + dy * F(1,4))) +
           dx * (F(2,1) + dy * (F(2,2) + dy * (F(2,3) + dy * F(2,4))) +
           dx * (F(3,1) + dy * (F(3,2) + dy * (F(3,3) + dy * F(3,4))) +
           dx * (F(4,1) + dy * (F(4,2) + dy * (F(4,3) + dy * F(4,4))))));
  out[1] = F(2,1) + dy * (F(2,2) + dy * (F(2,3) + dy * F(2,4))) +
           2.0 * dx * (F(3,1) + dy * (F(3,2) + dy * (F(3,3) + dy * F(3,4))) +
           1.5 * dx * (F(4,1) + dy * (F(4,2) + dy * (F(4,3) + dy * F(4,4)))));
  out[2] = F(1,2) + dy * (2.0 * F(1,3) + dy * 3.0 * F(1,4)) +
           dx * (F(2,2) + dy * (2.0 * F(2,3) + dy * 3.0 * F(2,4)) +
           dx * (F(3,2) + dy * (2.0 * F(3,3) + dy * 3.0 * F(3,4)) +
           dx * (F(4,2) + dy * (2.0 * F(4,3) + dy * 3.0 * F(4,4)))));
  out[3] = 2.0 * (F(3,1) + dy * (F(3,2) + dy * (F(3,3) + dy * F(3,4)))) +
           6.0 * dx * (F(4,1) + dy * (F(4,2) + dy * (F(4,3) + dy * F(4,4))));          /* fxx */
  out[5] = 2.0 * F(1,3) + 6.0 * dy * F(1,4) +
           dx * (2.0 * F(2,3) + 6.0 * dy * F(2,4) +
           dx * (2.0 * F(3,3) + 6.0 * dy * F(3,4) + dx * (2.0 * F(4,3) + 6.0 * dy * F(4,4)))); /* fyy */
  out[4] = F(2,2) + dy * (2.0 * F(2,3) + dy * 3.0 * F(2,4)) +
           2. * dx * (F(3,2) + dy * (2.0 * F(3,3) + dy * 3.0 * F(3,4)) +
           1.5 * dx * (F(4,2) + dy * (2.0 * F(4,3) + dy * 3.0 * F(4,4))));              /* fxy */
#undef F
}

/* --------------------------------------------------------------------------------------------
 * axisym_toroid_eq + eqdsk_magnetics_spline_interp
 *   axisym_toroid_eq_m.f90:215-362, eqdsk_magnetics_spline_interp_m.f90:206-282,
 *   density_spline_interp_m.f90:109-130, temperature_spline_interp_m.f90
 * ------------------------------------------------------------------------------------------ */
/* solovev_magnetics        solovev_magnetics_m.f90:127-181 (+ solovev_magnetics_psi :183-213): the magnetics of
 * solovev_eq above, statement for statement, as a magnetics model of axisym_toroid_eq.  Returns its own box flag
 * ('R out_of_bounds' / 'z out_of_bounds': the same box as the caller's, without the caller's 1e-13 margin). */
static int solovev_magnetics(const rays_params_t* P, const double rvec[3], double bvec[3], double gbt[3][3],
                             double* psiN_out, double gpN[3]) {
  const rays_solovev_params_t* S = &P->solovev;
  int err = 0;
  double x = rvec[0], y = rvec[1], z = rvec[2];
  double r = sqrt(x * x + y * y);
  if (r < S->box_rmin || r > S->box_rmax) err = RAYS_STOP_SOLMAG_R_OUT_OF_BOUNDS; /* :147 */
  if (z < S->box_zmin || z > S->box_zmax) err = RAYS_STOP_SOLMAG_Z_OUT_OF_BOUNDS; /* :148 */
  double bp0 = S->bphi0 * S->iota0;
  double psi, gradpsi[3];
  solovev_psi(P, rvec, &psi, gradpsi, psiN_out, gpN); /* solovev_magnetics_psi == solovev_psi, term for term */
  double br = -bp0 * r * z / sq(S->rmaj * S->kappa);
  double bz = bp0 * (sq(z / (S->rmaj * S->kappa)) + .5 * (sq(r / S->rmaj) - 1.));
  double bphi = S->bphi0 * S->rmaj / r;
  double dbrdr = br / r;
  double dbrdz = -bp0 * r / sq(S->rmaj * S->kappa);
  double dbzdr = bp0 * r / sq(S->rmaj);
  double dbzdz = bp0 * 2. * z / sq(S->rmaj * S->kappa);
  double dbphidr = -bphi / r;
  bvec[0] = br * x / r - bphi * y / r;
  bvec[1] = br * y / r + bphi * x / r;
  bvec[2] = bz;
  gbt[0][0] = (dbrdr * sq(x) + br * sq(y) / r + (-dbphidr + bphi / r) * x * y) / sq(r);
  gbt[1][0] = ((dbrdr - br / r) * x * y - dbphidr * sq(y) - bphi * sq(x) / r) / sq(r);
  gbt[2][0] = dbrdz * x / r;
  gbt[0][1] = ((dbrdr - br / r) * x * y + dbphidr * sq(x) + bphi * sq(y) / r) / sq(r);
  gbt[1][1] = (dbrdr * sq(y) + br * sq(x) / r + (dbphidr - bphi / r) * x * y) / sq(r);
  gbt[2][1] = dbrdz * y / r;
  gbt[0][2] = dbzdr * x / r;
  gbt[1][2] = dbzdr * y / r;
  gbt[2][2] = dbzdz;
  return err;
}

/* ---- 'eqdsk_magnetics_lin_interp': eqdsk_utilities_m.f90:144-306, eqdsk_magnetics_lin_interp_m.f90:146-249 ----
 * GetPsi does not bound its cell indices; where the central differences reach one cell beyond the grid it reads the
 * neighbouring column of Psi(NRBOX, NZBOX) through the storage order.  Same here; an index outside the array
 * altogether (undefined in the reference) is clamped. */
static double lin_psi_at(int i, int j) {
  long long k = (long long)(i - 1) + (long long)(j - 1) * AX.nr;
  const long long n = (long long)AX.nr * AX.nz;
  if (k < 0) k = 0;
  if (k >= n) k = n - 1;
  return AX.psi_fspl[k];
}
static double lin_getpsi(double R, double Z) { /* :144-162 */
  const double hr = AX.r_grid[1] - AX.r_grid[0], hz = AX.z_grid[1] - AX.z_grid[0];
  const int i = 1 + (int)((R - AX.r_grid[0]) / hr);
  const int j = 1 + (int)((Z - AX.z_grid[0]) / hz);
  const int ic = i < 1 ? 1 : (i > AX.nr ? AX.nr : i), jc = j < 1 ? 1 : (j > AX.nz ? AX.nz : j);
  const double x = (R - AX.r_grid[ic - 1]) / hr;
  const double y = (Z - AX.z_grid[jc - 1]) / hz;
  return lin_psi_at(i, j) * (1. - x) * (1. - y) + lin_psi_at(i + 1, j) * x * (1. - y) +
         lin_psi_at(i, j + 1) * (1. - x) * y + lin_psi_at(i + 1, j + 1) * x * y;
}
static double lin_getrbphi(double R) { /* :168-184 */
  const double hr = AX.r_grid[1] - AX.r_grid[0];
  const int i = 1 + (int)((R - AX.r_grid[0]) / hr);
  const int ic = i < 1 ? 1 : (i > AX.nr - 1 ? AX.nr - 1 : i);
  const double x = (R - AX.r_grid[ic - 1]) / hr;
  return AX.rb_fspl[ic - 1] * (1. - x) + AX.rb_fspl[ic] * x;
}
static double lin_psiR(double R, double Z) { return (lin_getpsi(R + AX.dR, Z) - lin_getpsi(R - AX.dR, Z)) / 2. / AX.dR; }
static double lin_psiZ(double R, double Z) { return (lin_getpsi(R, Z + AX.dZ) - lin_getpsi(R, Z - AX.dZ)) / 2. / AX.dZ; }
static double lin_psiRR(double R, double Z) {
  return (lin_getpsi(R + 2. * AX.dR, Z) - 2. * lin_getpsi(R, Z) + lin_getpsi(R - 2. * AX.dR, Z)) / AX.dR / AX.dR;
}
static double lin_psiZZ(double R, double Z) {
  return (lin_getpsi(R, Z + 2. * AX.dZ) - 2. * lin_getpsi(R, Z) + lin_getpsi(R, Z - 2. * AX.dZ)) / AX.dZ / AX.dZ;
}
static double lin_psiRZ(double R, double Z) {
  const double R1 = R - AX.dR, R2 = R + AX.dR, Z1 = Z - AX.dZ, Z2 = Z + AX.dZ;
  return (lin_getpsi(R2, Z2) - lin_getpsi(R1, Z2) - lin_getpsi(R2, Z1) + lin_getpsi(R1, Z1)) / 4. / AX.dR / AX.dZ;
}
static double lin_rbphiR(double R) { return (lin_getrbphi(R + AX.dR) - lin_getrbphi(R - AX.dR)) / 2. / AX.dR; }
/* eqdsk_magnetics_lin_interp (:146-214) */
static void lin_magnetics(const rays_params_t* P, const double rvec[3], double bvec[3], double gbt[3][3],
                          double* psiN, double gpN[3]) {
  const double x = rvec[0], y = rvec[1], z = rvec[2];
  const double r = sqrt(x * x + y * y);
  const double psi = lin_getpsi(r, z);
  const double br = -lin_psiZ(r, z) / r;
  const double bz = lin_psiR(r, z) / r;
  const double bphi = lin_getrbphi(r) / r;
  const double gradpsi[3] = {x * bz, y * bz, -r * br};
  *psiN = psi / P->axisym.psiB;
  gpN[0] = gradpsi[0] / P->axisym.psiB; gpN[1] = gradpsi[1] / P->axisym.psiB; gpN[2] = gradpsi[2] / P->axisym.psiB;
  const double dbrdr = -br / r - lin_psiRZ(r, z) / r;
  const double dbrdz = -lin_psiZZ(r, z) / r;
  const double dbzdr = -bz / r + lin_psiRR(r, z) / r;
  const double dbzdz = lin_psiRZ(r, z) / r;
  const double dbphidr = (lin_rbphiR(r) - bphi) / r;
  bvec[0] = br * x / r - bphi * y / r;
  bvec[1] = br * y / r + bphi * x / r;
  bvec[2] = bz;
  gbt[0][0] = (dbrdr * sq(x) + br * sq(y) / r + (-dbphidr + bphi / r) * x * y) / sq(r);
  gbt[1][0] = ((dbrdr - br / r) * x * y - dbphidr * sq(y) - bphi * sq(x) / r) / sq(r);
  gbt[2][0] = dbrdz * x / r;
  gbt[0][1] = ((dbrdr - br / r) * x * y + dbphidr * sq(x) + bphi * sq(y) / r) / sq(r);
  gbt[1][1] = (dbrdr * sq(y) + br * sq(x) / r + (dbphidr - bphi / r) * x * y) / sq(r);
  gbt[2][1] = dbrdz * y / r;
  gbt[0][2] = dbzdr * x / r;
  gbt[1][2] = dbzdr * y / r;
  gbt[2][2] = dbzdz;
}

static int axisym_eq(const rays_params_t* P, const double rvec[3], double bvec[3], double gbt[3][3],
                     double* ns, double (*gradns)[3], double* ts, double (*gradts)[3], int check_box) {
  const rays_axisym_params_t* S = &P->axisym;
  const int nspec = P->nspec;
  const double Tiny = 10.0e-14;
  int err = 0;
  const double x = rvec[0], y = rvec[1], z = rvec[2];
  const double r = sqrt(x * x + y * y);
  if (r < S->box_rmin - Tiny || r > S->box_rmax + Tiny) err = RAYS_STOP_AXI_R_OUT_OF_BOX;
  if (z < S->box_zmin - Tiny || z > S->box_zmax + Tiny) err = RAYS_STOP_AXI_Z_OUT_OF_BOX;
  if (err && check_box) return err;
  err = 0;
  double psiN_m, gpN_m[3];
  int mag_err = 0;
  if (S->magnetics_model == RAYS_AXI_MAG_SOLOVEV) {
    mag_err = solovev_magnetics(P, rvec, bvec, gbt, &psiN_m, gpN_m);
    if (mag_err && check_box) return mag_err; /* (the reference goes on with an undefined psiN: within 1e-13 of the box) */
  }
  double psiN = 0., gpN[3] = {0., 0., 0.};
  if (S->magnetics_model == RAYS_AXI_MAG_SOLOVEV) {
    psiN = psiN_m;
    gpN[0] = gpN_m[0]; gpN[1] = gpN_m[1]; gpN[2] = gpN_m[2];
  } else if (S->magnetics_model == RAYS_AXI_MAG_EQDSK_LIN) {
    lin_magnetics(P, rvec, bvec, gbt, &psiN, gpN);
  } else { /* eqdsk_magnetics_spline_interp */
    double f6[6], RBphi, RBphiR;
    spl2_fpp(r, z, f6);
    const double psi = f6[0], PsiR = f6[1], PsiZ = f6[2], PsiRR = f6[3], PsiRZ = f6[4], PsiZZ = f6[5];
    spl1_fp(AX.rb_grid, AX.rb_fspl, AX.n_rb, r, &RBphi, &RBphiR);
    const double br = PsiZ / r, bz = -PsiR / r, bphi = RBphi / r;
    const double gradpsi[3] = {-x * bz, -y * bz, r * br};
    psiN = psi / S->psiB;
    gpN[0] = gradpsi[0] / S->psiB; gpN[1] = gradpsi[1] / S->psiB; gpN[2] = gradpsi[2] / S->psiB;
    const double dbrdr = -br / r + PsiRZ / r;
    const double dbrdz = PsiZZ / r;
    const double dbzdr = -bz / r - PsiRR / r;
    const double dbzdz = -PsiRZ / r;
    const double dbphidr = (RBphiR - bphi) / r;
    bvec[0] = br * x / r - bphi * y / r;
    bvec[1] = br * y / r + bphi * x / r;
    bvec[2] = bz;
    gbt[0][0] = (dbrdr * sq(x) + br * sq(y) / r + (-dbphidr + bphi / r) * x * y) / sq(r);
    gbt[1][0] = ((dbrdr - br / r) * x * y - dbphidr * sq(y) - bphi * sq(x) / r) / sq(r);
    gbt[2][0] = dbrdz * x / r;
    gbt[0][1] = ((dbrdr - br / r) * x * y + dbphidr * sq(x) + bphi * sq(y) / r) / sq(r);
    gbt[1][1] = (dbrdr * sq(y) + br * sq(x) / r + (dbphidr - bphi / r) * x * y) / sq(r);
    gbt[2][1] = dbrdz * y / r;
    gbt[0][2] = dbzdr * x / r;
    gbt[1][2] = dbzdr * y / r;
    gbt[2][2] = dbzdz;
  }
  if (psiN > S->plasma_psi_limit) err = RAYS_STOP_OUT_OF_PLASMA; /* :288 */

  /* density :290-312 */
  if (S->density_prof_model == RAYS_AXI_N_CONSTANT) {
    for (int is = 0; is <= nspec; is++) {
      ns[is] = P->n0s[is];
      gradns[is][0] = gradns[is][1] = gradns[is][2] = 0.;
    }
  } else {
    double dens = 0., dd_psi = 0.;
    if (S->density_prof_model == RAYS_AXI_N_PARABOLIC) {
      parabolic_prof(psiN, S->d_scrape_off, S->alphan1, S->alphan2, &dens, &dd_psi);
    } else { /* density_spline_interp_m.f90:109-130 (dd_psi := 0 where the reference leaves it undefined) */
      if (psiN <= 1.0) spl1_fp(AX.ne_grid, AX.ne_fspl, AX.n_ne, psiN, &dens, &dd_psi);
      if (dens < S->d_scrape_off) {
        dens = S->d_scrape_off;
        dd_psi = 0.;
      }
    }
    for (int is = 0; is <= nspec; is++) {
      ns[is] = P->n0s[is] * dens;
      gradns[is][0] = P->n0s[is] * dd_psi * gpN[0];
      gradns[is][1] = P->n0s[is] * dd_psi * gpN[1];
      gradns[is][2] = P->n0s[is] * dd_psi * gpN[2];
    }
  }
  /* temperature :314-354 */
  for (int is = 0; is <= nspec; is++) {
    ts[is] = 0.;
    gradts[is][0] = gradts[is][1] = gradts[is][2] = 0.;
  }
  for (int is = 0; is <= nspec; is++) {
    const int m = S->t_prof_model[is];
    if (m == RAYS_AXI_T_CONSTANT) {
      ts[is] = P->t0s[is];
      for (int j = 0; j <= nspec; j++) gradts[j][0] = gradts[j][1] = gradts[j][2] = 0.; /* `gradts = 0.` (:328) */
    } else if (m == RAYS_AXI_T_PARABOLIC) {
      double t_prof, dt_dpsi;
      parabolic_prof(psiN, S->T_scrape_off, S->alphat1[is], S->alphat2[is], &t_prof, &dt_dpsi);
      ts[is] = P->t0s[is] * t_prof;
      for (int i = 0; i < 3; i++) gradts[is][i] = P->t0s[is] * dt_dpsi * gpN[i];
    } else if (m == RAYS_AXI_T_SPLINE) {
      double Te = 0., dTe = 0., Ti = 0., dTi = 0.;
      if (psiN <= 1.0) {
        spl1_fp(AX.te_grid, AX.te_fspl, AX.n_te, psiN, &Te, &dTe);
        spl1_fp(AX.ti_grid, AX.ti_fspl, AX.n_ti, psiN, &Ti, &dTi);
      }
      if (Te < S->T_scrape_off) { Te = S->T_scrape_off; dTe = 0.; }
      if (Ti < S->T_scrape_off) { Ti = S->T_scrape_off; dTi = 0.; }
      const double T = is == 0 ? Te : Ti, dT = is == 0 ? dTe : dTi;
      ts[is] = P->t0s[is] * T;
      for (int i = 0; i < 3; i++) gradts[is][i] = P->t0s[is] * dT * gpN[i];
    }
  }
  double mn = ns[0], mt = ts[0];
  for (int is = 1; is <= nspec; is++) {
    if (ns[is] < mn) mn = ns[is];
    if (ts[is] < mt) mt = ts[is];
  }
  if (mn < 0.) err = RAYS_STOP_NEGATIVE_DENS;
  if (mt < 0.) err = RAYS_STOP_NEGATIVE_TEMP;
  return check_box ? err : 0;
}

/* --------------------------------------------------------------------------------------------
 * equilibrium              equilibrium_m.f90:135-272
 * ------------------------------------------------------------------------------------------ */
static void equilibrium(const rays_params_t* P, rf_ctx rf, const double rvec[3], eq_point* eq,
                        int check_box) {
  const int nspec = P->nspec;
  double bvec[3], gbt[3][3], ns[NS0], gradns[NS0][3], ts[NS0], gradts[NS0][3];
  int err;
  if (P->equilib_model == RAYS_EQ_SLAB)
    err = slab_eq(P, rvec, bvec, gbt, ns, gradns, ts, gradts, check_box);
  else if (P->equilib_model == RAYS_EQ_SOLOVEV)
    err = solovev_eq(P, rvec, bvec, gbt, ns, gradns, ts, gradts, check_box);
  else
    err = axisym_eq(P, rvec, bvec, gbt, ns, gradns, ts, gradts, check_box);
  eq->err = err;
  if (err) return; /* :198-202: eq is left otherwise undefined */

  memset(eq->ns, 0, sizeof eq->ns); memset(eq->gradns, 0, sizeof eq->gradns);
  memset(eq->ts, 0, sizeof eq->ts); memset(eq->gradts, 0, sizeof eq->gradts);
  memset(eq->omgc, 0, sizeof eq->omgc); memset(eq->omgp2, 0, sizeof eq->omgp2);
  memset(eq->alpha, 0, sizeof eq->alpha); memset(eq->gamma, 0, sizeof eq->gamma);
  for (int i = 0; i < 3; i++) {
    eq->bvec[i] = bvec[i];
    for (int j = 0; j < 3; j++) eq->gradbtensor[i][j] = gbt[i][j];
  }
  for (int is = 0; is <= nspec; is++) {
    eq->ns[is] = ns[is];
    eq->ts[is] = ts[is];
    for (int i = 0; i < 3; i++) {
      eq->gradns[is][i] = gradns[is][i];
      eq->gradts[is][i] = gradts[is][i];
    }
  }
  /* :238-241 */
  double bmag = sqrt((0. + sq(bvec[0]) + sq(bvec[1])) + sq(bvec[2]));
  double bunit[3] = {bvec[0] / bmag, bvec[1] / bmag, bvec[2] / bmag};
  eq->bmag = bmag;
  /* :244-246 */
  double gradbmag[3];
  for (int i = 0; i < 3; i++) {
    double s = 0.;
    for (int j = 0; j < 3; j++) s += gbt[i][j] * bunit[j];
    gradbmag[i] = s;
  }
  for (int i = 0; i < 3; i++) {
    eq->bunit[i] = bunit[i];
    eq->gradbmag[i] = gradbmag[i];
    for (int j = 0; j < 3; j++) /* :254-257 */
      eq->gradbunit[i][j] = (gbt[i][j] - gradbmag[i] * bunit[j]) / bmag;
  }
  /* :262-265 */
  for (int is = 0; is <= nspec; is++) {
    eq->omgc[is] = P->qs[is] * bmag / P->ms[is];
    eq->omgp2[is] = ns[is] * sq(P->qs[is]) / (P->eps0 * P->ms[is]);
    eq->alpha[is] = eq->omgp2[is] / sq(rf.omgrf);
    eq->gamma[is] = eq->omgc[is] / rf.omgrf;
  }
}

/* --------------------------------------------------------------------------------------------
 * deriv_cold               deriv_cold.f90:1-228
 * ------------------------------------------------------------------------------------------ */
static void deriv_cold(const rays_params_t* P, rf_ctx rf, const eq_point* eq, const double nvec[3],
                       double dddx[3], double dddk[3], double* dddw) {
  const int n = P->nspec + 1;
  const double omgrf = rf.omgrf, k0 = rf.k0;
  double alpha[NS0], gamma[NS0];
  for (int is = 0; is < n; is++) {
    alpha[is] = eq->alpha[is];
    gamma[is] = eq->gamma[is];
  }
  /* :45-46 */
  double n3 = 0.;
  for (int i = 0; i < 3; i++) n3 += nvec[i] * eq->bunit[i];
  double s = 0.;
  for (int i = 0; i < 3; i++) s += sq(nvec[i] - n3 * eq->bunit[i]);
  double n1 = sqrt(s);
  /* :50-51 */
  double dn3dk[3], dn12dk[3], dn3dx[3], dn12dx[3];
  for (int i = 0; i < 3; i++) {
    dn3dk[i] = eq->bunit[i] / k0;
    dn12dk[i] = (2. / k0) * (nvec[i] - n3 * eq->bunit[i]);
  }
  /* :54-57 */
  for (int i = 0; i < 3; i++) {
    double t = 0.;
    for (int j = 0; j < 3; j++) t += eq->gradbunit[i][j] * nvec[j];
    dn3dx[i] = t;
  }
  for (int i = 0; i < 3; i++) dn12dx[i] = -2. * n3 * dn3dx[i];
  /* :59-67 */
  double dadx[3][NS0], dgdx[3][NS0];
  for (int i = 0; i < 3; i++)
    for (int is = 0; is < n; is++) {
      dadx[i][is] = eq->alpha[is] * eq->gradns[is][i] / eq->ns[is];
      dgdx[i][is] = gamma[is] * eq->gradbmag[i] / eq->bmag;
    }
  /* :72-75 */
  double dn3dw = -n3 / omgrf;
  double dn12dw = (-2. / omgrf) * sq(n1);
  double dadw[NS0], dgdw[NS0];
  for (int is = 0; is < n; is++) {
    dadw[is] = -2. / omgrf * alpha[is];
    dgdw[is] = -1. / omgrf * gamma[is];
  }
  /* :78-79 */
  double sa = 0.;
  for (int is = 0; is < n; is++) sa += alpha[is];
  double p = 1. - sa;
  double t = 1.;
  for (int is = 0; is < n; is++) t *= (1. - sq(gamma[is]));
  /* :83-91 */
  double dq1da[NS0], dq2da[NS0];
  for (int is1 = 0; is1 < n; is1++) {
    dq1da[is1] = 1.;
    dq2da[is1] = 1.;
    for (int is = 0; is < n; is++)
      if (is != is1) {
        dq1da[is1] = dq1da[is1] * (1. + gamma[is]);
        dq2da[is1] = dq2da[is1] * (1. - gamma[is]);
      }
  }
  /* :94-101 */
  double q1 = 0., q2 = 0., su = 0.;
  for (int is = 0; is < n; is++) q1 += alpha[is] * dq1da[is];
  for (int is = 0; is < n; is++) q2 += alpha[is] * dq2da[is];
  for (int is = 0; is < n; is++) su += alpha[is] * dq1da[is] * dq2da[is];
  double u = t - su;
  double q = 2. * u - t + q1 * q2;
  /* :104-112 */
  double duda[NS0], dqda[NS0], ddda[NS0];
  const double n3_2 = sq(n3), n3_4 = pow4(n3), n1_2 = sq(n1), n1_4 = pow4(n1);
  for (int is = 0; is < n; is++) {
    duda[is] = -dq1da[is] * dq2da[is];
    dqda[is] = 2. * duda[is] + dq1da[is] * q2 + q1 * dq2da[is];
    ddda[is] = -t * n3_4 + (2. * (u - p * duda[is]) + (-t + duda[is]) * n1_2) * n3_2 - q +
               p * dqda[is] - (dqda[is] - u + p * duda[is]) * n1_2 + duda[is] * n1_4;
  }
  /* :116-125  gp(is1,is2) */
  double gp[NS0][NS0], gm[NS0][NS0], gpm[NS0][NS0];
  for (int is1 = 0; is1 < n; is1++)
    for (int is2 = 0; is2 < n; is2++) {
      gp[is1][is2] = 1.;
      gm[is1][is2] = 1.;
      for (int is = 0; is < n; is++)
        if (is != is1 && is != is2) {
          gp[is1][is2] = gp[is1][is2] * (1. + gamma[is]);
          gm[is1][is2] = gm[is1][is2] * (1. - gamma[is]);
        }
    }
  for (int is1 = 0; is1 < n; is1++)
    for (int is2 = 0; is2 < n; is2++) gpm[is1][is2] = gp[is1][is2] * gm[is1][is2];
  /* :128-154 */
  double dtdg[NS0], dudg[NS0], dq1dg[NS0], dq2dg[NS0], dqdg[NS0], dddg[NS0];
  for (int is = 0; is < n; is++) dtdg[is] = 2. * gamma[is] * duda[is];
  for (int is = 0; is < n; is++) {
    double a = 0.;
    for (int j = 0; j < n; j++) a += alpha[j] * gpm[j][is];
    dudg[is] = a;
  }
  for (int is = 0; is < n; is++) dudg[is] = dtdg[is] + 2. * gamma[is] * (dudg[is] + alpha[is] * duda[is]);
  for (int is = 0; is < n; is++) {
    double a = 0.;
    for (int j = 0; j < n; j++) a += alpha[j] * gp[j][is];
    dq1dg[is] = a;
  }
  for (int is = 0; is < n; is++) dq1dg[is] = dq1dg[is] - alpha[is] * dq1da[is];
  for (int is = 0; is < n; is++) {
    double a = 0.;
    for (int j = 0; j < n; j++) a += alpha[j] * gm[j][is];
    dq2dg[is] = a;
  }
  for (int is = 0; is < n; is++) dq2dg[is] = -dq2dg[is] + alpha[is] * dq2da[is];
  for (int is = 0; is < n; is++) {
    dqdg[is] = 2. * dudg[is] - dtdg[is] + dq1dg[is] * q2 + q1 * dq2dg[is];
    dddg[is] = dtdg[is] * p * n3_4 + (-2. * p * dudg[is] + (dtdg[is] * p + dudg[is]) * n1_2) * n3_2 +
               p * dqdg[is] - (dqdg[is] + p * dudg[is]) * n1_2 + dudg[is] * n1_4;
  }
  /* :157-158 */
  double dddn3 = (4. * t * p * n3_2 + 2. * (-2. * p * u + (t * p + u) * n1_2)) * n3;
  double dddn12 = (t * p + u) * n3_2 - (q + p * u) + 2. * u * n1_2;
  /* :162-171 */
  for (int i = 0; i < 3; i++) dddk[i] = dddn3 * dn3dk[i] + dddn12 * dn12dk[i];
  for (int i = 0; i < 3; i++) {
    double a = 0.;
    for (int is = 0; is < n; is++) a += ddda[is] * dadx[i][is] + dddg[is] * dgdx[i][is];
    dddx[i] = a;
  }
  for (int i = 0; i < 3; i++) dddx[i] = dddx[i] + dddn3 * dn3dx[i] + dddn12 * dn12dx[i];
  double a = 0.;
  for (int is = 0; is < n; is++) a += ddda[is] * dadw[is] + dddg[is] * dgdw[is];
  *dddw = a + dddn3 * dn3dw + dddn12 * dn12dw;
}

/* --------------------------------------------------------------------------------------------
 * complex helpers mirroring flang's lowering
 * ------------------------------------------------------------------------------------------ */
typedef struct { double re, im; } cplx;
static inline cplx c_make(double re, double im) { cplx z = {re, im}; return z; }
static inline cplx c_add(cplx a, cplx b) { return c_make(a.re + b.re, a.im + b.im); }
static inline cplx c_sub(cplx a, cplx b) { return c_make(a.re - b.re, a.im - b.im); }
static inline cplx c_neg(cplx a) { return c_make(-a.re, -a.im); }
static inline cplx c_conj(cplx a) { return c_make(a.re, -a.im); }
static inline cplx c_mul(cplx a, cplx b) { /* (ac-bd, bc+ad) */
  return c_make(a.re * b.re - a.im * b.im, a.im * b.re + a.re * b.im);
}
static inline cplx c_rmul(double r, cplx a) { return c_make(r * a.re, r * a.im); }
/* compiler-rt __divdc3 for finite operands (the non-finite recovery branches cannot trigger
 * here without the inputs already being NaN, in which case the plain formula propagates it). */
static cplx c_div(cplx x, cplx y) {
  double a = x.re, b = x.im, c = y.re, d = y.im;
  int ilogbw = 0;
  double logbw = logb(fmax(fabs(c), fabs(d)));
  if (isfinite(logbw)) {
    ilogbw = (int)logbw;
    c = scalbn(c, -ilogbw);
    d = scalbn(d, -ilogbw);
  }
  double denom = c * c + d * d;
  double re = scalbn((a * c + b * d) / denom, -ilogbw);
  double im = scalbn((b * c - a * d) / denom, -ilogbw);
  return c_make(re, im);
}

/* suscep_cold + dielectric_cold      suscep_m.f90:53-86, 142-176 */
static void dielectric_cold(const rays_params_t* P, const eq_point* eq, cplx eps[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) eps[i][j] = c_make(0., 0.);
  for (int is = 0; is <= P->nspec; is++) {
    double alphas = eq->alpha[is], gammas = eq->gamma[is];
    cplx chi[3][3];
    chi[0][0] = c_make(-alphas / (1. - sq(gammas)), 0.); /* :72 */
    chi[1][1] = chi[0][0];
    chi[2][2] = c_make(-alphas, 0.);
    /* :75  -zi*alphas*gammas/(1.-gammas**2): flang forms zi*alphas*gammas with complex
     * multiplies against (alphas,0),(gammas,0), divides by ((1-g^2),0) via __divdc3, negates. */
    cplx zi = c_make(0., 1.);
    cplx t1 = c_mul(zi, c_make(alphas, 0.));
    cplx t2 = c_mul(t1, c_make(gammas, 0.));
    cplx t3 = c_div(t2, c_make(1. - sq(gammas), 0.));
    chi[0][1] = c_neg(t3);
    chi[0][2] = c_make(0., 0.);
    chi[1][2] = c_make(0., 0.);
    chi[1][0] = c_neg(chi[0][1]);
    chi[2][0] = chi[0][2];
    chi[2][1] = c_neg(chi[1][2]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) eps[i][j] = c_add(eps[i][j], chi[i][j]);
  }
  for (int i = 0; i < 3; i++) eps[i][i] = c_add(eps[i][i], c_make(1.0, 0.));
}

/* epsn and its determinant: shared by determ (deriv_num.f90:99-153) and residual
 * (check_save.f90:163-235).  eps_norm (check_save.f90:211) is optional. */
static cplx epsn_det(const rays_params_t* P, const eq_point* eq, const double n[3],
                     double eps_norm[3][3]) {
  cplx eps[3][3], eps_h[3][3], epsn[3][3];
  dielectric_cold(P, eq, eps);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) eps_h[i][j] = c_rmul(.5, c_add(eps[i][j], c_conj(eps[j][i])));
  double nsq = (0. + sq(n[0]) + sq(n[1])) + sq(n[2]);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      int kd = ((i + 1) / (j + 1)) * ((j + 1) / (i + 1)); /* int(i/j)*int(j/i) */
      epsn[i][j] = c_make(eps_h[i][j].re + n[i] * n[j] - kd * nsq, eps_h[i][j].im);
      if (eps_norm) eps_norm[i][j] = hypot(eps_h[i][j].re, eps_h[i][j].im) + fabs(n[i] * n[j]);
    }
  cplx t1 = c_mul(epsn[2][2], c_sub(c_mul(epsn[0][0], epsn[1][1]), c_mul(epsn[1][0], epsn[0][1])));
  cplx t2 = c_mul(epsn[2][1], c_sub(c_mul(epsn[0][0], epsn[1][2]), c_mul(epsn[1][0], epsn[0][2])));
  cplx t3 = c_mul(epsn[2][0], c_sub(c_mul(epsn[0][1], epsn[1][2]), c_mul(epsn[1][1], epsn[0][2])));
  return c_add(c_sub(t1, t2), t3);
}

/* determ                   deriv_num.f90:99-153   (ray_dispersion_model == 'cold') */
static double determ(const rays_params_t* P, rf_ctx rf, const eq_point* eq, const double kvec[3]) {
  double k3 = 0.;
  for (int i = 0; i < 3; i++) k3 += kvec[i] * eq->bunit[i];
  double s = 0.;
  for (int i = 0; i < 3; i++) s += sq(kvec[i] - k3 * eq->bunit[i]);
  double k1 = sqrt(s);
  double n[3] = {k1 / rf.k0, 0., k3 / rf.k0};
  cplx ctmp = epsn_det(P, eq, n, NULL);
  /* :137 abs(Im) > 1e-7 -> `stop 1`: Im is exactly 0 (or NaN) for the cold tensor. */
  double pr = 1.;
  for (int is = 0; is < NS0; is++) pr *= (1. - sq(eq->gamma[is])); /* product over 0:nspec0 */
  return ctmp.re * pr;
}

/* --------------------------------------------------------------------------------------------
 * deriv_num                deriv_num.f90:1-155
 * delta = 1.e-6 is a single-precision literal widened to double (:37).
 * Perturbed-position equilibria that fall outside the box are evaluated without the box test
 * (the reference reads an undefined eq_point there); see DESIGN.md.
 * ------------------------------------------------------------------------------------------ */
static void deriv_num(const rays_params_t* P, rf_ctx rf0, const eq_point* eq0, const double* v,
                      double dddx[3], double dddk[3], double* dddw) {
  const double delta = (double)1.e-6f;
  double rvec0[3] = {v[0], v[1], v[2]}, kvec0[3] = {v[3], v[4], v[5]};
  double omgrf0 = rf0.omgrf;
  eq_point eq_plus, eq_minus;
  for (int i = 0; i < 3; i++) {
    double rvec[3] = {rvec0[0], rvec0[1], rvec0[2]};
    double change = delta;
    rvec[i] = rvec0[i] + change;
    equilibrium(P, rf0, rvec, &eq_plus, 0);
    rvec[i] = rvec0[i] - change;
    equilibrium(P, rf0, rvec, &eq_minus, 0);
    double det_plus = determ(P, rf0, &eq_plus, kvec0);
    double det_minus = determ(P, rf0, &eq_minus, kvec0);
    dddx[i] = (det_plus - det_minus) / (2. * change);
  }
  for (int i = 0; i < 3; i++) {
    double kvec[3] = {kvec0[0], kvec0[1], kvec0[2]};
    double change = fmax(delta, fabs(delta * kvec[i])) / 2.;
    kvec[i] = kvec0[i] + change;
    double det_plus = determ(P, rf0, eq0, kvec);
    kvec[i] = kvec0[i] - change;
    double det_minus = determ(P, rf0, eq0, kvec);
    dddk[i] = (det_plus - det_minus) / (2. * change);
  }
  rf_ctx rfp, rfm;
  rfp.omgrf = omgrf0 * (1. + delta / 2.);
  rfp.k0 = rfp.omgrf / P->clight;
  equilibrium(P, rfp, rvec0, &eq_plus, 0);
  double det_plus = determ(P, rfp, &eq_plus, kvec0);
  rfm.omgrf = omgrf0 * (1. - delta / 2.);
  rfm.k0 = rfm.omgrf / P->clight;
  equilibrium(P, rfm, rvec0, &eq_minus, 0);
  double det_minus = determ(P, rfm, &eq_minus, kvec0);
  *dddw = (det_plus - det_minus) / (omgrf0 * delta);
  /* :83-84 resets k0 = omgrf/clight: see rays_oracle_trace (k0 after the first deriv_num call) */
}

/* --------------------------------------------------------------------------------------------
 * Z function of a real argument by cubic spline: zfun_real_arg_spline_D / zfun0_real_arg_D
 * (math_functions_lib/zfunctions_m.f90:351-432) with cspeval/cspevx/cspevfn on a uniform grid
 * (splines_lib/cspeval.f90).  The table fsplRe(4,nx) is built by the host
 * (initialize_spline_coeffs, :436-466) and handed over with rays_oracle_set_zfun_table.
 * ------------------------------------------------------------------------------------------ */
static int zf_nx = 0;
static double zf_xmin = 0., zf_xmax = 0.;
static double* zf_fspl = NULL; /* [nx][4] */

int rays_oracle_set_zfun_table(const double* fspl_re, int nx, double x_min, double x_max) {
  free(zf_fspl);
  zf_fspl = (double*)malloc(sizeof(double) * 4 * (size_t)nx);
  if (!zf_fspl) return 1;
  memcpy(zf_fspl, fspl_re, sizeof(double) * 4 * (size_t)nx);
  zf_nx = nx;
  zf_xmin = x_min;
  zf_xmax = x_max;
  return 0;
}

static inline double zf_x(int i) { /* x_grid(i), 1-based (zfunctions_m.f90:444) */
  return zf_xmin + (double)(i - 1) * (zf_xmax - zf_xmin) / (double)(zf_nx - 1);
}

/* real**integer as flang lowers it (llvm.powi -> compiler-rt __powidf2: square and multiply) */
static double powi_rt(double a, int b) {
  const int recip = b < 0;
  double r = 1.;
  for (;;) {
    if (b & 1) r = r * a;
    b /= 2;
    if (b == 0) break;
    a = a * a;
  }
  return recip ? 1. / r : r;
}

static cplx zfun_real_arg_spline(double z) {
  const double spline_range = 10.0;
  double re;
  if (fabs(z) <= spline_range) {
    /* cspevx, ilinx = 1 (cspeval.f90:138-146) */
    const int nxm = zf_nx - 1;
    const double x1 = zf_x(1), xn = zf_x(zf_nx);
    int ii = (int)(1 + nxm * (z - x1) / (xn - x1));
    int i = ii < nxm ? ii : nxm;
    if (z < zf_x(i)) i = i - 1;
    else if (z > zf_x(i + 1)) i = i + 1;
    const double dx = z - zf_x(i);
    const double* f = zf_fspl + 4 * (size_t)(i - 1);
    re = f[0] + dx * (f[1] + dx * (f[2] + dx * f[3])); /* cspevfn :239 */
  } else { /* asymptotic expansion :408-414 (unreachable from damp_fund_ECH: |xi| <= 5) */
    static const double A[6] = {1., 1. / 2., 3. / 4., 15. / 8., 105. / 16., 945. / 32.};
    const double z_inv = 1.0 / z;
    re = 0.;
    for (int i = 1; i <= 6; i++) re = re - powi_rt(z_inv, 2 * i - 1) * A[i - 1];
  }
  const double sqrt_pi = sqrt(3.14159265358979323846);
  return c_make(re, sqrt_pi * exp(-(z * z)));
}

static cplx zfun0_real_arg(double z, double kz) { /* :351-372 */
  if (kz > 0.) return zfun_real_arg_spline(z);
  return c_neg(zfun_real_arg_spline(-z));
}

/* --------------------------------------------------------------------------------------------
 * DAMP_FUND_ECH             damp_fund_ECH.f90:2-128
 * D_WARM and DELTA are default COMPLEX (single precision, :36): both assignments truncate.
 * ------------------------------------------------------------------------------------------ */
static void damp_fund_ech(const rays_params_t* P, rf_ctx rf, const eq_point* eq, const double* v,
                          const double vg[3], double* ki) {
  *ki = 0.;
  const double k0 = rf.k0;
  double kvec[3] = {v[3], v[4], v[5]};
  double nvec[3] = {kvec[0] / k0, kvec[1] / k0, kvec[2] / k0};
  double k3 = 0.;
  for (int i = 0; i < 3; i++) k3 += kvec[i] * eq->bunit[i];
  double s = 0.;
  for (int i = 0; i < 3; i++) s += sq(kvec[i] - k3 * eq->bunit[i]);
  const double k1 = sqrt(s);
  const double R3 = k3 / k0, R1 = k1 / k0;
  const double R1S = sq(R1), R3S = sq(R3), RS = R1S + R3S;
  const double B1 = eq->gamma[0], BETAE = sq(B1);
  if (R3 == 0.) return;
  const double vth = sqrt(2. * eq->ts[0] / P->ms[0]);
  const double VT = vth / P->clight;
  const double xi = (rf.omgrf + eq->omgc[0]) / (k3 * vth);
  if (fabs(xi) > 5.) return; /* also true for NaN? no: NaN > 5 is false -> continues, as in Fortran */
  const cplx zf = zfun0_real_arg(xi, k3);
  const double Pa = eq->alpha[0];
  const double Q = Pa / 2. / (1 - B1);
  const double L1 = (1. - Q) * RS * R1S + (1. - Pa) * RS * R3S - (1. - Q) * (1. - Pa) * (RS + R3S) -
                    (1 - 2. * Q) * R1S + (1 - 2 * Q) * (1 - Pa);
  const double L2 = -(Pa / B1 * (RS * R1S - (1. - 2. * Q) * R1S)) +
                    Pa * Pa / 4. / BETAE * R1S / R3S * (RS + R3S - 2. * (1. - 2. * Q));
  const double L5 = Pa * (RS * R3S - (1. - Q) * (RS + R3S) + (1. - 2. * Q));
  const double F = (1. - B1) * R3 * VT * (L1 + L2 + R1S / 2. / R3 / BETAE * VT * xi * L5);
  const cplx zinv = c_div(c_make(1., 0.), zf);
  const cplx par = c_make(xi + zinv.re, 0. + zinv.im);
  const cplx dw8 = c_neg(c_mul(c_make(F, 0.), par));
  const float dw_re = (float)dw8.re, dw_im = (float)dw8.im; /* D_WARM is COMPLEX(4) */
  const double A = 1. - Pa - BETAE;
  const double B = -((1. - Pa) * A + sq(1. - Pa) - BETAE) + (A + (1. - Pa) * (1. - BETAE)) * R3S;
  const double DDNX2 = 2. * A * R1S + B;
  const double DDNZ = 2. * R3 * ((A + (1. - Pa) * (1. - BETAE)) * R1S + (1 - Pa) * (2. * (1. - BETAE) * R3S - 2. * A));
  double DDN[3], vgu[3];
  const double nvg = sqrt((0. + sq(vg[0]) + sq(vg[1])) + sq(vg[2]));
  double dot = 0.;
  for (int i = 0; i < 3; i++) {
    DDN[i] = DDNX2 * (2 * (nvec[i] - R3 * eq->bunit[i])) + DDNZ * eq->bunit[i];
    vgu[i] = vg[i] / nvg;
  }
  for (int i = 0; i < 3; i++) dot += DDN[i] * vgu[i];
  const cplx d8 = c_div(c_make(-(double)dw_re, -(double)dw_im), c_make(dot, 0.));
  const float delta_im = (float)d8.im; /* DELTA is COMPLEX(4) */
  *ki = k0 * (double)delta_im;          /* ksi(0) = k0*aimag(DELTA); ki = ksi(0) */
}

/* --------------------------------------------------------------------------------------------
 * eqn_ray                  eqn_ray.f90:1-236     returns stop code (0 = ok)
 * ------------------------------------------------------------------------------------------ */
static int eqn_ray(const rays_params_t* P, rf_ctx rf, const double* v, double* dvds) {
  const int nv = P->nv;
  double rvec[3] = {v[0], v[1], v[2]}, kvec[3] = {v[3], v[4], v[5]};
  double nvec[3] = {kvec[0] / rf.k0, kvec[1] / rf.k0, kvec[2] / rf.k0};
  eq_point eq;
  equilibrium(P, rf, rvec, &eq, 1);
  if (eq.err) return eq.err; /* :90-102 */
  double dddx[3], dddk[3], dddw;
  if (P->ray_deriv == RAYS_DERIV_COLD)
    deriv_cold(P, rf, &eq, nvec, dddx, dddk, &dddw);
  else
    deriv_num(P, rf, &eq, v, dddx, dddk, &dddw);
  double vg[3], vg0, vg_unit[3];
  if (dddw != 0.) { /* :133  true for NaN */
    for (int i = 0; i < 3; i++) vg[i] = -dddk[i] / dddw;
    vg0 = sqrt((0. + sq(vg[0]) + sq(vg[1])) + sq(vg[2]));
    for (int i = 0; i < 3; i++) vg_unit[i] = vg[i] / vg0;
  } else
    return RAYS_STOP_INFINITE_VG_RHS;
  double dsd;
  if (P->ray_param == RAYS_PARAM_ARCL) { /* :150-170 */
    if (dddk[0] != 0. || dddk[1] != 0. || dddk[2] != 0.) {
      double sgn = copysign(1.0, dddw); /* sign(1.,dddw) */
      double nk = sqrt((0. + sq(dddk[0]) + sq(dddk[1])) + sq(dddk[2]));
      for (int i = 0; i < 3; i++) dvds[i] = -sgn * dddk[i] / nk;
      for (int i = 0; i < 3; i++) dvds[3 + i] = sgn * dddx[i] / nk;
      dsd = 1.;
    } else
      return RAYS_STOP_RAY_STALLED;
  } else { /* 'time' :172-181 */
    for (int i = 0; i < 3; i++) dvds[i] = -dddk[i] / dddw;
    for (int i = 0; i < 3; i++) dvds[3 + i] = dddx[i] / dddw;
    dsd = vg0;
  }
  dvds[6] = dsd; /* :190 */
  int nv0 = 7;
  if (P->damping_model != RAYS_DAMP_NONE) { /* :196-213 */
    double ki;
    damp_fund_ech(P, rf, &eq, v, vg, &ki);
    dvds[7] = dsd * 2. * ki * (1. - v[7]);
    nv0 = 8;
    if (P->multi_spec_damping) { /* :207-212: ksi(0) = ki, ksi(1:nspec) = 0 (damp_fund_ECH.f90:122-124) */
      for (int is = 0; is <= P->nspec; is++) {
        const double ksi = is == 0 ? ki : 0.;
        dvds[nv0 + is] = dsd * 2. * ksi * (1. - v[7]);
      }
      nv0 = nv0 + 1 + P->nspec;
    }
  }
  if (P->integrate_eq_gradients) { /* :217-229 */
    for (int j = 0; j < 3; j++) {
      double a = 0.;
      for (int i = 0; i < 3; i++) a += dsd * vg_unit[i] * eq.gradbtensor[i][j];
      dvds[nv0 + j] = a;
    }
    double a = 0., b = 0.;
    for (int i = 0; i < 3; i++) a += dsd * vg_unit[i] * eq.gradns[0][i];
    for (int i = 0; i < 3; i++) b += dsd * vg_unit[i] * eq.gradts[0][i];
    dvds[nv0 + 3] = a;
    dvds[nv0 + 4] = b;
  }
  (void)nv;
  return 0;
}

/* --------------------------------------------------------------------------------------------
 * check_save + residual    check_save.f90:1-237
 * Returns: stop code to latch into ray_stop (0 = none); *stop = stop_ode.
 * ------------------------------------------------------------------------------------------ */
static int check_save(const rays_params_t* P, rf_ctx rf, const double* v, double* resid, int* stop) {
  int flag = 0;
  *stop = 0;
  eq_point eq;
  double rvec[3] = {v[0], v[1], v[2]};
  equilibrium(P, rf, rvec, &eq, 1);
  if (eq.err) { /* :41-43: flag text only, stop_ode untouched, eq undefined in the reference */
    flag = eq.err;
    equilibrium(P, rf, rvec, &eq, 0);
  }
  double kvec[3] = {v[3], v[4], v[5]};
  double k3 = 0.;
  for (int i = 0; i < 3; i++) k3 += kvec[i] * eq.bunit[i];
  double s = 0.;
  for (int i = 0; i < 3; i++) s += sq(kvec[i] - k3 * eq.bunit[i]);
  double k1 = sqrt(s);
  double nvec[3] = {kvec[0] / rf.k0, kvec[1] / rf.k0, kvec[2] / rf.k0};
  /* residual :163-235 */
  double n[3] = {k1 / rf.k0, 0., k3 / rf.k0};
  double en[3][3];
  cplx ctmp = epsn_det(P, &eq, n, en);
  double den = en[2][2] * (en[0][0] * en[1][1]) + en[2][2] * (en[1][0] * en[0][1]) +
               en[2][1] * (en[0][0] * en[1][2]) + en[2][1] * (en[1][0] * en[0][2]) +
               en[2][0] * (en[0][1] * en[1][2]) + en[2][0] * (en[1][1] * en[0][2]);
  /* eps_norm is declared complex(KIND=rkind) (check_save.f90:186): the quotient real/complex goes
   * through __divdc3 and the real part is assigned to the result. */
  *resid = c_div(c_make(hypot(ctmp.re, ctmp.im), 0.), c_make(den, 0.)).re;
  if (*resid > P->dispersion_resid_limit) { /* :68-71 */
    *stop = 1;
    flag = RAYS_STOP_DISP_RESIDUAL;
  }
  double dddx[3], dddk[3], dddw;
  deriv_cold(P, rf, &eq, nvec, dddx, dddk, &dddw); /* :82 */
  if (fabs(dddw) > 2.2250738585072014e-308) {     /* :90 tiny(dddw) */
    /* vg, vg0 only feed messages */
  } else {
    *stop = 1;
    flag = RAYS_STOP_INFINITE_VG_CHECK; /* :107-108 */
  }
  if (P->damping_model != RAYS_DAMP_NONE) { /* :114-125 (the damping() call only feeds messages) */
    if (v[7] > P->total_damping_limit) {
      *stop = 1;
      flag = RAYS_STOP_TOTAL_ABSORPTION;
    }
  }
  return flag;
}

/* --------------------------------------------------------------------------------------------
 * RK4_ode                  RK4_ode_m.f90:59-94
 * ------------------------------------------------------------------------------------------ */
static int rk4_ode(const rays_params_t* P, rf_ctx rf, double* v, double* s, double sout) {
  const int nv = P->nv;
  double f1[RAYS_ORACLE_NV_MAX], f2[RAYS_ORACLE_NV_MAX], f3[RAYS_ORACLE_NV_MAX],
      f4[RAYS_ORACLE_NV_MAX], w[RAYS_ORACLE_NV_MAX];
  double ds = sout - *s;
  int st;
  if ((st = eqn_ray(P, rf, v, f1))) return st;
  for (int i = 0; i < nv; i++) w[i] = v[i] + ds * f1[i] / 2.0;
  if ((st = eqn_ray(P, rf, w, f2))) return st;
  for (int i = 0; i < nv; i++) w[i] = v[i] + ds * f2[i] / 2.0;
  if ((st = eqn_ray(P, rf, w, f3))) return st;
  for (int i = 0; i < nv; i++) w[i] = v[i] + ds * f3[i];
  if ((st = eqn_ray(P, rf, w, f4))) return st;
  for (int i = 0; i < nv; i++) v[i] = v[i] + ds * (f1[i] + 2.0 * f2[i] + 2.0 * f3[i] + f4[i]) / 6.0;
  *s = sout;
  return 0;
}

/* SG: see rays_oracle_sg.c */
int rays_oracle_sg_ode(const rays_params_t* P, rays_oracle_rhs_fn f, void* ctx, double* v, double* s,
                       double* sout, double* rel_err, double* abs_err, int* nrhs);

typedef struct { const rays_params_t* P; rf_ctx rf; } rhs_ctx;
static int rhs_thunk(void* c, const double* v, double* dvds) {
  rhs_ctx* r = (rhs_ctx*)c;
  return eqn_ray(r->P, r->rf, v, dvds);
}

/* --------------------------------------------------------------------------------------------
 * initialize_ode_vector    initialize_ode_vector.f90:25-54
 * ------------------------------------------------------------------------------------------ */
static void initialize_ode_vector(const rays_params_t* P, rf_ctx rf, const double* r0,
                                  const double* n0, double* v) {
  for (int i = 0; i < 3; i++) v[i] = r0[i];
  for (int i = 0; i < 3; i++) v[3 + i] = rf.k0 * n0[i];
  v[6] = 0.;
  int nv0 = 7;
  if (P->damping_model != RAYS_DAMP_NONE) {
    v[7] = 0.;
    nv0 = 8;
    if (P->multi_spec_damping) { /* initialize_ode_vector.f90:36-39 */
      for (int is = 0; is <= P->nspec; is++) v[8 + is] = 0.;
      nv0 = nv0 + 1 + P->nspec;
    }
  }
  if (P->integrate_eq_gradients) {
    eq_point eq;
    equilibrium(P, rf, v, &eq, 0);
    for (int i = 0; i < 3; i++) v[nv0 + i] = eq.bvec[i];
    v[nv0 + 3] = eq.ns[0];
    v[nv0 + 4] = eq.ts[0];
  }
}

int rays_oracle_check_params(const rays_params_t* P) {
  if (P->abi_version != RAYS_ABI_VERSION) return 1;
  if (P->nspec < 0 || P->nspec > RAYS_NSPEC0) return 2;
  if (P->multi_spec_damping && !P->damping_model) return 5; /* rows 9.. would never be set (eqn_ray.f90:196) */
  if (P->nv != 7 + (P->damping_model ? 1 : 0) + (P->multi_spec_damping ? 1 + P->nspec : 0) +
                   (P->integrate_eq_gradients ? 5 : 0))
    return 3;
  if (P->damping_model == RAYS_DAMP_FUND_ECH && !zf_fspl) return 6;
  if (P->equilib_model == RAYS_EQ_AXISYM && P->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_SPLINE &&
      (!AX.psi_fspl || !AX.rb_fspl || AX.lin))
    return 7;
  if (P->equilib_model == RAYS_EQ_AXISYM && P->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_LIN &&
      (!AX.psi_fspl || !AX.rb_fspl || !AX.lin))
    return 7;
  if (P->nv > RAYS_ORACLE_NV_MAX) return 3;
  if (P->equilib_model == RAYS_EQ_SOLOVEV)
    for (int is = 0; is <= P->nspec; is++)
      if (P->solovev.t_prof_model[is] != RAYS_SOLOVEV_T_ZERO &&
          P->solovev.t_prof_model[is] != RAYS_SOLOVEV_T_PARABOLIC)
        return 4;
  return 0;
}

/* --------------------------------------------------------------------------------------------
 * trace_rays               ray_tracing.f90:1-290 (one ray)
 * ------------------------------------------------------------------------------------------ */
static void trace_one(const rays_params_t* P, const double* r0, const double* n0, double* ray_vec,
                      double* residual, int* npoints, int* stop_code, double* end_ray_vec,
                      double* end_resid, double* max_resid, long long* nrhs_total) {
  const int nv = P->nv;
  rf_ctx rf = {P->omgrf, P->k0};
  double v[RAYS_ORACLE_NV_MAX];
  int nstep = 0, flag = 0, stop = 0;
  double s = 0., sout = 0., resid = 0.;
  double rel_err = P->rel_err0, abs_err = P->abs_err0; /* SG_ode_m.f90:81-82 */
  initialize_ode_vector(P, rf, r0, n0, v);
  for (int i = 0; i < nv; i++) ray_vec[i] = v[i]; /* :92 */
  residual[0] = resid;                            /* :93 */
  flag = check_save(P, rf, v, &resid, &stop);     /* :100 */
  if (stop) {                                     /* :101-112: summary fields stay zero */
    *stop_code = flag;
    *npoints = 1;
    return;
  }
  int rk0_fix = 0;
  for (;;) {
    s = sout;
    sout = sout + P->ds; /* :118-119 running sum */
    if (sout > P->s_max) { flag = RAYS_STOP_SOUT_GT_SMAX; break; } /* :128-147 */
    if (nstep + 1 > P->nstep_max) {                                 /* :150-172 */
      flag = RAYS_STOP_NSTEP_MAX;
      nstep = P->nstep_max;
      break;
    }
    int st;
    if (P->ode_solver == RAYS_ODE_RK4) {
      st = rk4_ode(P, rf, v, &s, sout);
    } else {
      rhs_ctx c = {P, rf};
      int nrhs = 0;
      st = rays_oracle_sg_ode(P, rhs_thunk, &c, v, &s, &sout, &rel_err, &abs_err, &nrhs);
      if (nrhs_total) *nrhs_total += nrhs;
    }
    /* deriv_num.f90:83-84: after the first numerical-derivative RHS the module k0 has been
     * reset to omgrf/clight, which can differ from rf_m's k0 = omgrf/clight only if the two
     * divisions differ -- they are the same expression, so k0 is unchanged. */
    (void)rk0_fix;
    if (st) { flag = st; break; } /* :177-197 */
    flag = check_save(P, rf, v, &resid, &stop); /* :212 */
    if (stop) break;                            /* :214-234 */
    nstep = nstep + 1;                          /* :237-243 */
    for (int i = 0; i < nv; i++) ray_vec[(size_t)nstep * nv + i] = v[i];
    residual[nstep] = resid;
  }
  *npoints = nstep + 1; /* :252 */
  *stop_code = flag;
  for (int i = 0; i < nv; i++) end_ray_vec[i] = v[i]; /* :260 */
  /* :255 end_residuals = residual(nstep) (1-based: the point BEFORE the last); nstep = 0 reads
   * residual(0,iray), out of bounds in the reference -> defined as 0 here. */
  *end_resid = nstep >= 1 ? residual[nstep - 1] : 0.;
  /* :256 maxval(abs(residual(1:nstep))): empty -> -huge */
  double m = -1.7976931348623157e308;
  for (int i = 0; i < nstep; i++)
    if (fabs(residual[i]) > m) m = fabs(residual[i]);
  *max_resid = m;
}

int rays_oracle_trace(const rays_params_t* P, int nray, const double* rvec0,
                      const double* rindex_vec0, double* ray_vec, double* residual,
                      int32_t* npoints, int32_t* stop_code, double* end_ray_vec,
                      double* end_residuals, double* max_residuals, int nthreads,
                      long long* nrhs_total) {
  int rc = rays_oracle_check_params(P);
  if (rc) return rc;
  const size_t npt = (size_t)P->nstep_max + 1, nv = (size_t)P->nv;
  long long nrhs = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : nrhs)
#endif
  for (int ir = 0; ir < nray; ir++) {
    double er = 0., mr = 0.;
    double ev[RAYS_ORACLE_NV_MAX];
    memset(ev, 0, sizeof ev);
    int np = 0, sc = 0;
    long long local = 0;
    trace_one(P, rvec0 + 3 * (size_t)ir, rindex_vec0 + 3 * (size_t)ir, ray_vec + (size_t)ir * npt * nv,
              residual + (size_t)ir * npt, &np, &sc, ev, &er, &mr, &local);
    nrhs += local;
    npoints[ir] = np;
    stop_code[ir] = sc;
    if (end_ray_vec) memcpy(end_ray_vec + (size_t)ir * nv, ev, nv * sizeof(double));
    if (end_residuals) end_residuals[ir] = er;
    if (max_residuals) max_residuals[ir] = mr;
  }
  if (nrhs_total) *nrhs_total = nrhs;
  return 0;
}

/* --------------------------------------------------------------------------------------------
 * probe: evaluate the RHS pieces at one state (unit parity against the reference dump)
 * eq_out: bvec3 bmag gradbmag3 bunit3 gradbunit9 gradbtensor9 (Fortran column-major order, as
 * written by ref_dump_driver.f90) then per species ns gradns3 ts gradts3 omgc omgp2 alpha gamma.
 * ------------------------------------------------------------------------------------------ */
void rays_oracle_probe(const rays_params_t* P, const double* v, double* eq_out, double* cold7,
                       double* num7, double* dvds, double* resid, int32_t* codes) {
  rf_ctx rf = {P->omgrf, P->k0};
  eq_point eq;
  double rvec[3] = {v[0], v[1], v[2]};
  equilibrium(P, rf, rvec, &eq, 1);
  codes[0] = eq.err;
  if (eq.err) equilibrium(P, rf, rvec, &eq, 0);
  double* o = eq_out;
  for (int i = 0; i < 3; i++) *o++ = eq.bvec[i];
  *o++ = eq.bmag;
  for (int i = 0; i < 3; i++) *o++ = eq.gradbmag[i];
  for (int i = 0; i < 3; i++) *o++ = eq.bunit[i];
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++) *o++ = eq.gradbunit[i][j];
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++) *o++ = eq.gradbtensor[i][j];
  for (int is = 0; is <= P->nspec; is++) {
    *o++ = eq.ns[is];
    for (int i = 0; i < 3; i++) *o++ = eq.gradns[is][i];
    *o++ = eq.ts[is];
    for (int i = 0; i < 3; i++) *o++ = eq.gradts[is][i];
    *o++ = eq.omgc[is];
    *o++ = eq.omgp2[is];
    *o++ = eq.alpha[is];
    *o++ = eq.gamma[is];
  }
  double nvec[3] = {v[3] / rf.k0, v[4] / rf.k0, v[5] / rf.k0};
  deriv_cold(P, rf, &eq, nvec, cold7, cold7 + 3, cold7 + 6);
  deriv_num(P, rf, &eq, v, num7, num7 + 3, num7 + 6);
  for (int i = 0; i < P->nv; i++) dvds[i] = 0.;
  codes[1] = eqn_ray(P, rf, v, dvds);
  int stop;
  codes[2] = check_save(P, rf, v, resid, &stop);
  codes[3] = stop;
}
