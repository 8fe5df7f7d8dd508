/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the RAYS hot path (see rays_oracle.c).
 * Shares only the POD parameter block / stop codes with the product header. */
#ifndef RAYS_ORACLE_H
#define RAYS_ORACLE_H
#include "../include/rays_hip.h"

#define RAYS_ORACLE_NV_MAX 19 /* 7 + 1 + (1 + nspec0) + 5: ode_m.f90:160-173 */

#ifdef __cplusplus
extern "C" {
#endif

typedef int (*rays_oracle_rhs_fn)(void* ctx, const double* v, double* dvds);

int rays_oracle_check_params(const rays_params_t* P);

/* Z-function spline table for damp_fund_ECH (same meaning as rays_hip_set_zfun_table). */
int rays_oracle_set_zfun_table(const double* fspl_re, int nx, double x_min, double x_max);
/* axisym_toroid spline tables (same meaning as rays_hip_set_axisym_tables). */
int rays_oracle_set_axisym_tables(const rays_axisym_tables_t* t);
/* tables of magnetics_model = 'eqdsk_magnetics_lin_interp' (same meaning as rays_hip_set_eqdsk_lin_tables) */
int rays_oracle_set_eqdsk_lin_tables(const rays_axisym_tables_t* t, double dR, double dZ);

/* Same argument meaning as rays_hip_trace (include/rays_hip.h); host pointers; nthreads <= 0 =
 * all OpenMP threads.  nrhs_total (optional) counts eqn_ray calls made by the SG stepper. */
int rays_oracle_trace(const rays_params_t* P, int nray, const double* rvec0,
                      const double* rindex_vec0, double* ray_vec, double* residual,
                      int32_t* npoints, int32_t* stop_code, double* end_ray_vec,
                      double* end_residuals, double* max_residuals, int nthreads,
                      long long* nrhs_total);

/* One-state probe of equilibrium / deriv_cold / deriv_num / eqn_ray / check_save.
 * eq_out needs 28 + 12*(nspec+1) doubles; cold7/num7 = dddx(3) dddk(3) dddw; codes[4]. */
void rays_oracle_probe(const rays_params_t* P, const double* v, double* eq_out, double* cold7,
                       double* num7, double* dvds, double* resid, int32_t* codes);

#ifdef __cplusplus
}
#endif
#endif
