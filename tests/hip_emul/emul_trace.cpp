// TEST INFRASTRUCTURE ONLY: runs the product trace kernels lane-by-lane on the CPU (see
// hip/hip_runtime.h in this directory).  Exposes one C entry used by tests/test_cpu_kernel_emul.py.
#include <hip/hip_runtime.h>
#include <vector>
RAYS_EMUL_DEFINE_GLOBALS
#include "../../rays_amd/csrc/rays_rk4.hpp"
#include "../../rays_amd/csrc/rays_sg.hpp"

// make_dev_params is host code in rays_capi.hip; reuse its text through a small include shim
#include "emul_dev_params.inc"

template <int EQ, int DERIV>
static void run(int solver, int nv, const rays::DevParams& D, const rays::TraceArgs& A) {
  threadIdx.x = 0; blockIdx.x = 0; blockDim.x = 1; gridDim.x = 1;
  if (nv == 7) {
    if (solver == 0) rays::rk4_trace_kernel<EQ, 2, DERIV, 7, 8>(D, A);
    else rays::sg_trace_kernel<EQ, 2, DERIV, 7, 8>(D, A);
  } else {
    if (solver == 0) rays::rk4_trace_kernel<EQ, 2, DERIV, 8, 8>(D, A);
    else rays::sg_trace_kernel<EQ, 2, DERIV, 8, 8>(D, A);
  }
}

static std::vector<double> g_zfun;
static int g_zf_nx = 0;
static double g_zf_xmin = 0., g_zf_xmax = 0.;
extern "C" int rays_emul_set_zfun_table(const double* f, int nx, double x_min, double x_max) {
  g_zfun.assign(f, f + 4 * (size_t)nx);
  g_zf_nx = nx; g_zf_xmin = x_min; g_zf_xmax = x_max;
  return 0;
}

extern "C" int rays_emul_trace(const rays_params_t* p, int nray, const double* rvec0,
                               const double* rindex_vec0, double* ray_vec, double* residual,
                               int32_t* npoints, int32_t* stop_code, double* end_ray_vec,
                               double* end_residuals, double* max_residuals) {
  if (p->nspec != 1 || (p->nv != 7 && p->nv != 8)) return 1;  // emulation instantiates NS = 2, nv = 7 | 8
  unsigned counter = 0;
  rays::TraceArgs A;
  A.nray = nray; A.rvec0 = rvec0; A.rindex_vec0 = rindex_vec0; A.ray_vec = ray_vec;
  A.residual = residual; A.npoints = npoints; A.stop_code = stop_code; A.end_ray_vec = end_ray_vec;
  A.end_residuals = end_residuals; A.max_residuals = max_residuals; A.next_ray = &counter;
  rays::DevParams D = make_dev_params(*p);
  if (p->damping_model) {
    if (g_zfun.empty()) return 2;
    D.zf_fspl = g_zfun.data(); D.zf_nx = g_zf_nx; D.zf_xmin = g_zf_xmin; D.zf_xmax = g_zf_xmax;
  }
  const int e = p->equilib_model, d = p->ray_deriv, s = p->ode_solver;
  if (e == 0 && d == 0) run<0, 0>(s, p->nv, D, A);
  else if (e == 0 && d == 1) run<0, 1>(s, p->nv, D, A);
  else if (e == 1 && d == 0) run<1, 0>(s, p->nv, D, A);
  else run<1, 1>(s, p->nv, D, A);
  return 0;
}
