// rays_deposition.hpp -- deposition profiles on the device: the step right
// after the hot path, applied to the trajectories while they are still resident in HBM.
//
// Reference path restated here:
//   calculate_deposition_profiles / bin_a_ray   post_process_lib/deposition_profiles_m.f90:228-292
//   Ptotal_axisym_psi_evaluator / _rho_         deposition_profiles_m.f90:458-503
//   binner_real (bin_to_uniform_grid)           math_functions_lib/bin_to_uniform_grid_m.f90
//   eqdsk_magnetics_spline_interp_psi / _rho    eqdsk_magnetics_spline_interp_m.f90:286-352
//
// One thread bins one ray (the binner walks the ray's points in order) into a row held in LDS and
// writes it to work[bin][ray] (bin-major, so both this write and the reduction read coalesce).
// The profile is then the sum over rays IN RAY ORDER -- the order of the reference's sum(work, 2);
// floating-point addition is not associative -- one wave per bin: the lanes load 64 consecutive
// rays at a time and the running sum walks through them with v_readlane.  It can be continued from
// another rank's partial sums, so the multi-GPU result is bit-identical to the single-process one.
#pragma once

#include "rays_device.hpp"

namespace rays {

struct DepArgs {
  int which;  // RAYS_DEP_PTOTAL_PSI | RAYS_DEP_PTOTAL_RHO | RAYS_DEP_PTOTAL_X
  int n_bins, nray, nv, npt;
  double grid_min, grid_max;
  const double* ray_vec;   // [nray][npt][nv]
  const int* npoints;      // [nray]
  const double* power;     // initial_ray_power[nray]
  const double* rho_grid;  // rho(psiN) spline (Ptotal_rho)
  const double* rho_fspl;
  int n_rho;
  double* work;            // [n_bins][nray]
};

// grid value of a ray point: psiN or rho(psiN) at (x, y, z); x itself for the slab's Ptotal_x
RAYS_DEV double dep_grid_value(const DevParams& P, const DepArgs& D, const double* v) {
  const double x = v[0], y = v[1], z = v[2];
  if (D.which == 2) return x;  // Ptotal_x_slab_evaluator (deposition_profiles_m.f90:449)
  const double r = sqrt(x * x + y * y);
  double f6[6];
  double psiN;
  if (P.a_mag_model == RAYS_AXI_MAG_SOLOVEV) {  // axisym_toroid_psi -> solovev_magnetics_psi (solovev_magnetics_m.f90:199-207)
    const double psi = P.half_bp0 * (sq(r * z / P.rk) + sq(r * r - P.rmaj2) / P.rmaj2 / 4.);
    psiN = psi / P.psiB;
  } else if (P.a_mag_model == RAYS_AXI_MAG_EQDSK_LIN) {  // eqdsk_magnetics_lin_interp_psi: GetPsi / PSIBOUND
    psiN = eqlin_getpsi(P, r, z) / P.a_psiB;
  } else {
    spl2_fpp(P, r, z, f6);
    psiN = f6[0] / P.a_psiB;  // psiN = Psi/PSIBOUND (:314)
  }
  if (D.which == 0) return psiN;
  double rho, drho;
  spl1_fp(D.rho_grid, D.rho_fspl, D.n_rho, psiN, rho, drho);
  return rho;
}

// bin_a_ray + binner_real for ray `iray` into row[b * stride], b = 0..n_bins-1 (zeroed here, as the
// binner does)
template <class RowPtr>
RAYS_DEV void deposit_ray(const DevParams& P, const DepArgs& D, int iray, RowPtr row_base, int stride) {
  const int n_bins = D.n_bins;
  struct Row {
    RowPtr p;
    int st;
    RAYS_DEV auto& operator[](int b) const { return p[b * st]; }
  } row{row_base, stride};
  for (int b = 0; b < n_bins; b++) row[b] = 0.;
  const int np = D.npoints[iray];
  const double* rv = D.ray_vec + (long long)iray * D.npt * D.nv;
  const double pw = D.power[iray];
  const double xmin = D.grid_min, xmax = D.grid_max;
  const double x_bin_width = (xmax - xmin) / (double)n_bins;
  double x_prev = 0., q_prev = 0.;
  for (int is = 0; is < np; is++) {
    const double* v = rv + (long long)is * D.nv;
    const double xq = dep_grid_value(P, D, v);
    const double q = v[7] * pw;  // ray_vec(8)*initial_ray_power
    if (is > 0) {
      double x_low = fmin(x_prev, xq), x_high = fmax(x_prev, xq);
      double ix_low = (x_low - xmin) / x_bin_width, ix_high = (x_high - xmin) / x_bin_width;
      const double delta_ix = ix_high - ix_low;
      int index_low = (int)floor(ix_low) + 1, index_high = (int)floor(ix_high) + 1;
      if (x_high >= xmax) index_high = n_bins;
      int delta_i = index_high - index_low;
      double delta_Q = q - q_prev;
      const double Q_density = delta_Q / delta_ix;
      bool skip = fabs(delta_Q) < 4.0 * 2.2250738585072014e-308;  // 4.0_skind*tiny(delta_Q)
      if (x_high < xmin || x_low > xmax) skip = true;
      if (!skip) {
        if (x_low < xmin) {
          delta_Q = delta_Q * (ix_high / delta_ix);
          ix_low = 0.0;
          index_low = 1;
          delta_i = index_high - index_low;
        }
        if (x_high > xmax) {
          delta_Q = delta_Q * (((double)n_bins - ix_low) / delta_ix);
          ix_high = (double)n_bins;
          index_high = n_bins;
          delta_i = index_high - index_low;
        }
        if (delta_i == 0) {
          if (index_low >= 1 && index_low <= n_bins) row[index_low - 1] = row[index_low - 1] + delta_Q;
        } else if (delta_i > 0) {
          const double Q_incrL = delta_Q * (((double)index_low - ix_low) / delta_ix);
          row[index_low - 1] = row[index_low - 1] + Q_incrL;
          const double Q_incrH = delta_Q * ((ix_high - (double)(index_high - 1)) / delta_ix);
          row[index_high - 1] = row[index_high - 1] + Q_incrH;
          for (int i = index_low + 1; i <= index_high - 1; i++) row[i - 1] = row[i - 1] + Q_density;
        }
      }
    }
    x_prev = xq;
    q_prev = q;
  }
}

}  // namespace rays
