// TEST INFRASTRUCTURE ONLY: runs the product trace kernels lane-by-lane on the CPU (see
// hip/hip_runtime.h in this directory).  Exposes one C entry used by tests/test_cpu_kernel_emul.py.
#define RAYS_EMUL_HANDOVER 1   // the tolerance kernels' hand-over of ill-conditioned steps + rk4_resume_ray, on the host
#include <hip/hip_runtime.h>
#include <vector>
RAYS_EMUL_DEFINE_GLOBALS
#include "../../rays_amd/csrc/rays_rk4.hpp"
#include "../../rays_amd/csrc/rays_sg.hpp"
#include "../../rays_amd/csrc/rays_ray_init.hpp"
#include "../../rays_amd/csrc/rays_deposition.hpp"

// make_dev_params is host code in rays_capi.hip; reuse its text through a small include shim
#include "emul_dev_params.inc"
#include "../../rays_amd/csrc/rays_fan_setup.inc"

// Instantiated here: two species with nv = 7 | 8 for every equilibrium; nv = 12 | 13 (integrate_eq_gradients)
// for the slab and Solovev; for the slab also one and three species (nv = 7).
template <int EQ, int DERIV, int NS, int NV>
static int run1(int solver, const rays::DevParams& D, const rays::TraceArgs& A) {
  threadIdx.x = 0; blockIdx.x = 0; blockDim.x = 1; gridDim.x = 1;
  if (solver == 0) {
    rays::rk4_trace_kernel<EQ, NS, DERIV, NV>(D, A);
    for (int ray = 0; ray < A.nray; ray++)   // what rays_capi.hip launches behind a tolerance kernel: rk4_resume_kernel
      if (A.stop_code[ray] == rays::kStopResumeExact) {
        rays::rays_emul_redo_steps++;
        rays::rk4_resume_ray<EQ, NS, DERIV, NV>(D, A, ray);
      }
  } else {
    rays::sg_trace_kernel<EQ, NS, DERIV, NV>(D, A);
  }
  return 0;
}
template <int EQ, int DERIV>
static int run(int solver, int ns, int nv, const rays::DevParams& D, const rays::TraceArgs& A) {
  if (ns == 2 && nv == 7) return run1<EQ, DERIV, 2, 7>(solver, D, A);
  if (ns == 2 && nv == 8) return run1<EQ, DERIV, 2, 8>(solver, D, A);
  if (ns == 2 && nv == 12) return run1<EQ, DERIV, 2, 12>(solver, D, A);
  if (ns == 2 && nv == 13) return run1<EQ, DERIV, 2, 13>(solver, D, A);
  if constexpr ((EQ & 3) == 0) {
    if (ns == 1 && nv == 7) return run1<EQ, DERIV, 1, 7>(solver, D, A);
    if (ns == 3 && nv == 7) return run1<EQ, DERIV, 3, 7>(solver, D, A);
    if (ns == 6 && nv == 7) return run1<EQ, DERIV, 6, 7>(solver, D, A);
  }
  if constexpr ((EQ & 3) == 1) {
    if (ns == 4 && nv == 7) return run1<EQ, DERIV, 4, 7>(solver, D, A);
  }
  return 1;
}
// multi_spec_damping kernels (EQ | kEqMultiSpec): Solovev nv = 10, slab nv = 15 (two species)
template <int EQ, int DERIV>
static int run_ms(int solver, int ns, int nv, const rays::DevParams& D, const rays::TraceArgs& A) {
  if constexpr ((EQ & 3) == 1) {
    if (ns == 2 && nv == 10) return run1<EQ, DERIV, 2, 10>(solver, D, A);
  }
  if constexpr ((EQ & 3) == 0) {
    if (ns == 2 && nv == 15) return run1<EQ, DERIV, 2, 15>(solver, D, A);
  }
  return 1;
}

extern "C" long long rays_emul_redo_steps(int reset) {
  const long long n = rays::rays_emul_redo_steps.load();
  if (reset) rays::rays_emul_redo_steps = 0;
  return n;
}

static std::vector<double> g_zfun;
static int g_zf_nx = 0;
static double g_zf_xmin = 0., g_zf_xmax = 0.;
extern "C" int rays_emul_set_zfun_table(const double* f, int nx, double x_min, double x_max) {
  g_zfun.assign(f, f + 4 * (size_t)nx);
  g_zf_nx = nx; g_zf_xmin = x_min; g_zf_xmax = x_max;
  return 0;
}

static std::vector<double> g_axi[11];
static int g_axi_n[6];
static int g_axi_lin = 0;
static double g_axi_dR = 0., g_axi_dZ = 0.;
extern "C" int rays_emul_set_axisym_tables(const rays_axisym_tables_t* t, int lin, double dR, double dZ) {
  g_axi_lin = lin; g_axi_dR = dR; g_axi_dZ = dZ;
  const double* src[11] = {t->r_grid, t->z_grid, t->psi_fspl, t->rb_grid, t->rb_fspl, t->ne_grid, t->ne_fspl,
                           t->te_grid, t->te_fspl, t->ti_grid, t->ti_fspl};
  const size_t len[11] = {(size_t)t->nr, (size_t)t->nz, (size_t)(lin ? 1 : 16) * t->nr * t->nz, lin ? (size_t)0 : (size_t)t->n_rb,
                          (size_t)(lin ? 1 : 4) * t->n_rb,
                          (size_t)t->n_ne, (size_t)4 * t->n_ne, (size_t)t->n_te, (size_t)4 * t->n_te,
                          (size_t)t->n_ti, (size_t)4 * t->n_ti};
  for (int k = 0; k < 11; k++) g_axi[k].assign(src[k] ? src[k] : nullptr, src[k] ? src[k] + len[k] : nullptr);
  g_axi_n[0] = t->nr; g_axi_n[1] = t->nz; g_axi_n[2] = t->n_rb; g_axi_n[3] = t->n_ne; g_axi_n[4] = t->n_te; g_axi_n[5] = t->n_ti;
  return 0;
}

// nray = total rays of the launch.  v0 / s0: optional starting states (rays_hip_ode_step_device);
// ds_run / rays_per_run: a fused `ds` scan (rays_hip_scan_device).
extern "C" int rays_emul_trace_ex(const rays_params_t* p, int nray, const double* rvec0,
                                  const double* rindex_vec0, double* ray_vec, double* residual,
                                  int32_t* npoints, int32_t* stop_code, double* end_ray_vec,
                                  double* end_residuals, double* max_residuals, const double* v0,
                                  const double* s0, const double* ds_run, int rays_per_run);
extern "C" int rays_emul_trace(const rays_params_t* p, int nray, const double* rvec0,
                               const double* rindex_vec0, double* ray_vec, double* residual,
                               int32_t* npoints, int32_t* stop_code, double* end_ray_vec,
                               double* end_residuals, double* max_residuals) {
  return rays_emul_trace_ex(p, nray, rvec0, rindex_vec0, ray_vec, residual, npoints, stop_code, end_ray_vec,
                            end_residuals, max_residuals, nullptr, nullptr, nullptr, 0);
}
extern "C" int rays_emul_trace_ex(const rays_params_t* p, int nray, const double* rvec0,
                                  const double* rindex_vec0, double* ray_vec, double* residual,
                                  int32_t* npoints, int32_t* stop_code, double* end_ray_vec,
                                  double* end_residuals, double* max_residuals, const double* v0,
                                  const double* s0, const double* ds_run, int rays_per_run) {
  unsigned counter = 0;
  rays::TraceArgs A = rays::TraceArgs();
  A.v0 = v0; A.s0 = s0; A.ds_run = ds_run; A.rays_per_run = rays_per_run;
  std::vector<double> sg_far(512, 0.0);  // one lane: the SG kernels' upper-tier workspace
  A.sg_far = sg_far.data(); A.sg_far_lanes = 1;
  A.nray = nray; A.rvec0 = rvec0; A.rindex_vec0 = rindex_vec0; A.ray_vec = ray_vec;
  A.residual = residual; A.npoints = npoints; A.stop_code = stop_code; A.end_ray_vec = end_ray_vec;
  A.end_residuals = end_residuals; A.max_residuals = max_residuals; A.next_ray = &counter;
  rays::DevParams D = make_dev_params(*p);
  if (p->damping_model) {
    if (g_zfun.empty()) return 2;
    D.zf_fspl = g_zfun.data(); D.zf_nx = g_zf_nx; D.zf_xmin = g_zf_xmin; D.zf_xmax = g_zf_xmax;
  }
  if (p->equilib_model == RAYS_EQ_AXISYM) {
    if (p->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_SPLINE && (g_axi[2].empty() || g_axi_lin)) return 3;
    if (p->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_LIN && (g_axi[2].empty() || !g_axi_lin)) return 3;
    D.a_lin_dR = g_axi_dR; D.a_lin_dZ = g_axi_dZ;
    D.a_nr = g_axi_n[0]; D.a_nz = g_axi_n[1]; D.a_n_rb = g_axi_n[2]; D.a_n_ne = g_axi_n[3]; D.a_n_te = g_axi_n[4]; D.a_n_ti = g_axi_n[5];
    D.a_r_grid = g_axi[0].data(); D.a_z_grid = g_axi[1].data(); D.a_psi_fspl = g_axi[2].data();
    D.a_rb_grid = g_axi[3].data(); D.a_rb_fspl = g_axi[4].data(); D.a_ne_grid = g_axi[5].data(); D.a_ne_fspl = g_axi[6].data();
    D.a_te_grid = g_axi[7].data(); D.a_te_fspl = g_axi[8].data(); D.a_ti_grid = g_axi[9].data(); D.a_ti_fspl = g_axi[10].data();
    set_spline_axes(D, D.a_r_grid, D.a_z_grid, D.a_rb_grid, D.a_ne_grid, D.a_te_grid, D.a_ti_grid);
  }
  // same kernel selection as rays_capi.hip: find_kernel (EQ = model | kEqUnitExp)
  const int e = p->equilib_model | (unit_exponents(*p) ? rays::kEqUnitExp : 0), d = p->ray_deriv, s = p->ode_solver;
  const rays::DevParams& D_ = D;
  if (p->multi_spec_damping) {
    if (e == 5 && d == 0) return run_ms<5 | rays::kEqMultiSpec, 0>(s, p->nspec + 1, p->nv, D_, A);
    if (e == 4 && d == 0) return run_ms<4 | rays::kEqMultiSpec, 0>(s, p->nspec + 1, p->nv, D_, A);
    return 4;
  }
#define RAYS_EMUL_CASE(E, D) if (e == E && d == D) return run<E, D>(s, p->nspec + 1, p->nv, D_, A); else
  RAYS_EMUL_CASE(0, 0) RAYS_EMUL_CASE(0, 1) RAYS_EMUL_CASE(1, 0) RAYS_EMUL_CASE(1, 1) RAYS_EMUL_CASE(2, 0) RAYS_EMUL_CASE(2, 1)
  RAYS_EMUL_CASE(4, 0) RAYS_EMUL_CASE(4, 1) RAYS_EMUL_CASE(5, 0) RAYS_EMUL_CASE(5, 1) RAYS_EMUL_CASE(6, 0) RAYS_EMUL_CASE(6, 1)
  return 4;
#undef RAYS_EMUL_CASE
  return 0;
}

// Ray initialisation (rays_ray_init.hpp: fan_member) run sequentially in the reference's loop order.
template <int EQ>
static bool emul_member(const rays::DevParams& D, const rays::FanArgs& F, const double* rvec, int ia, int ib, double* ri) {
  if constexpr (EQ == 0) {
    if (D.nspec == 2) return rays::fan_member<EQ, 3>(D, F, rvec, ia, ib, ri);
    if (D.nspec == 0) return rays::fan_member<EQ, 1>(D, F, rvec, ia, ib, ri);
    if (D.nspec == 5) return rays::fan_member<EQ, 6>(D, F, rvec, ia, ib, ri);
  }
  if constexpr (EQ == 1) {
    if (D.nspec == 3) return rays::fan_member<EQ, 4>(D, F, rvec, ia, ib, ri);
  }
  return rays::fan_member<EQ, 2>(D, F, rvec, ia, ib, ri);
}
extern "C" int rays_emul_ray_init(const rays_params_t* p, const rays_fan_t* fan, int nray_max, double* rvec0,
                                  double* rindex_vec0, int32_t* nray) {
  if (p->nspec != 1 && !((p->nspec == 2 || p->nspec == 0 || p->nspec == 5) && p->equilib_model == 0) &&
      !(p->nspec == 3 && p->equilib_model == 1)) return 1;
  rays::FanArgs F;
  std::vector<double> launch;
  int per_r = 0;
  const char* why = "";
  if (fan_setup(p, fan, nray_max, &F, &launch, &per_r, &why)) return 5;
  F.launch = launch.data();
  rays::DevParams D = make_dev_params(*p);
  if (p->equilib_model == RAYS_EQ_AXISYM) {
    if (p->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_SPLINE && (g_axi[2].empty() || g_axi_lin)) return 3;
    if (p->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_LIN && (g_axi[2].empty() || !g_axi_lin)) return 3;
    D.a_lin_dR = g_axi_dR; D.a_lin_dZ = g_axi_dZ;
    D.a_nr = g_axi_n[0]; D.a_nz = g_axi_n[1]; D.a_n_rb = g_axi_n[2]; D.a_n_ne = g_axi_n[3]; D.a_n_te = g_axi_n[4]; D.a_n_ti = g_axi_n[5];
    D.a_r_grid = g_axi[0].data(); D.a_z_grid = g_axi[1].data(); D.a_psi_fspl = g_axi[2].data();
    D.a_rb_grid = g_axi[3].data(); D.a_rb_fspl = g_axi[4].data(); D.a_ne_grid = g_axi[5].data(); D.a_ne_fspl = g_axi[6].data();
    D.a_te_grid = g_axi[7].data(); D.a_te_fspl = g_axi[8].data(); D.a_ti_grid = g_axi[9].data(); D.a_ti_fspl = g_axi[10].data();
    set_spline_axes(D, D.a_r_grid, D.a_z_grid, D.a_rb_grid, D.a_ne_grid, D.a_te_grid, D.a_ti_grid);
  }
  int count = 0;
  for (int il = 0; il < F.n_launch; il++)
    for (int ia = 0; ia < F.n_a; ia++)
      for (int ib = 0; ib < F.n_b; ib++) {
        double ri[3];
        const double* rv = &launch[3 * il];
        const bool ok = p->equilib_model == 0 ? emul_member<0>(D, F, rv, ia, ib, ri)
                        : p->equilib_model == 1 ? emul_member<1>(D, F, rv, ia, ib, ri) : emul_member<2>(D, F, rv, ia, ib, ri);
        if (!ok) continue;
        for (int i = 0; i < 3; i++) { rvec0[3 * count + i] = rv[i]; rindex_vec0[3 * count + i] = ri[i]; }
        count++;
      }
  *nray = count;
  return 0;
}

// Deposition profiles (rays_deposition.hpp: deposit_ray) run on the host, rays in order.
extern "C" int rays_emul_deposition(const rays_params_t* p, int which, int n_bins, int nray, const double* ray_vec,
                                    const int32_t* npoints, const double* power, const double* rho_grid,
                                    const double* rho_fspl, int n_rho, double* work, double* profile) {
  rays::DevParams D = make_dev_params(*p);
  if (which != 2) {
    if (p->equilib_model != RAYS_EQ_AXISYM) return 3;
    if (p->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_SPLINE && (g_axi[2].empty() || g_axi_lin)) return 3;
    if (p->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_LIN && (g_axi[2].empty() || !g_axi_lin)) return 3;
    D.a_lin_dR = g_axi_dR; D.a_lin_dZ = g_axi_dZ;
    D.a_nr = g_axi_n[0]; D.a_nz = g_axi_n[1]; D.a_n_rb = g_axi_n[2];
    D.a_r_grid = g_axi[0].data(); D.a_z_grid = g_axi[1].data(); D.a_psi_fspl = g_axi[2].data();
    D.a_rb_grid = g_axi[3].data(); D.a_rb_fspl = g_axi[4].data();
    set_spline_axes(D, D.a_r_grid, D.a_z_grid, D.a_rb_grid, nullptr, nullptr, nullptr);
  }
  rays::DepArgs A;
  A.which = which; A.n_bins = n_bins; A.nray = nray; A.nv = p->nv; A.npt = p->nstep_max + 1;
  A.grid_min = 0.; A.grid_max = 1.;
  if (which == 2) { A.grid_min = p->slab.xmin; A.grid_max = p->slab.xmax; }
  A.ray_vec = ray_vec; A.npoints = npoints; A.power = power; A.work = work;
  A.rho_grid = rho_grid; A.rho_fspl = rho_fspl; A.n_rho = n_rho;
  for (int r = 0; r < nray; r++) rays::deposit_ray(D, A, r, work + (size_t)r * n_bins, 1);
  for (int b = 0; b < n_bins; b++) {
    double s = 0.;
    for (int r = 0; r < nray; r++) s = s + work[(size_t)r * n_bins + b];
    profile[b] = s;
  }
  return 0;
}
