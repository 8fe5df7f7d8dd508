// TEST INFRASTRUCTURE ONLY: the lane-group Shampine-Gordon kernel (rays_amd/csrc/rays_sg_group.hpp) on the host wave
// emulator (hip/hip_wave_emul.h: 64 lanes per wave as fibers, __shfl / __any / __ballot are rendezvous), compared with
// the reference fixtures and the oracle by tests/test_cpu_group_emul.py.  Built as its own library: with
// RAYS_EMUL_WAVE the cross-lane operations are collective, so only kernels that issue them from wave-uniform control
// flow can run here (the one-lane entries of emul_trace.cpp are compiled along for their table plumbing, not called).
#define RAYS_EMUL_WAVE 1
#include "emul_trace.cpp"
#include "../../rays_amd/csrc/rays_sg_group.hpp"

static int attach_axisym_tables(const rays_params_t* p, rays::DevParams& D) {
  if (p->equilib_model != RAYS_EQ_AXISYM) return 0;
  if (p->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_SPLINE && (g_axi[2].empty() || g_axi_lin)) return 3;
  if (p->axisym.magnetics_model == RAYS_AXI_MAG_EQDSK_LIN && (g_axi[2].empty() || !g_axi_lin)) return 3;
  D.a_lin_dR = g_axi_dR; D.a_lin_dZ = g_axi_dZ;
  D.a_nr = g_axi_n[0]; D.a_nz = g_axi_n[1]; D.a_n_rb = g_axi_n[2]; D.a_n_ne = g_axi_n[3]; D.a_n_te = g_axi_n[4]; D.a_n_ti = g_axi_n[5];
  D.a_r_grid = g_axi[0].data(); D.a_z_grid = g_axi[1].data(); D.a_psi_fspl = g_axi[2].data();
  D.a_rb_grid = g_axi[3].data(); D.a_rb_fspl = g_axi[4].data(); D.a_ne_grid = g_axi[5].data(); D.a_ne_fspl = g_axi[6].data();
  D.a_te_grid = g_axi[7].data(); D.a_te_fspl = g_axi[8].data(); D.a_ti_grid = g_axi[9].data(); D.a_ti_fspl = g_axi[10].data();
  set_spline_axes(D, D.a_r_grid, D.a_z_grid, D.a_rb_grid, D.a_ne_grid, D.a_te_grid, D.a_ti_grid);
  return 0;
}

template <int EQ, int NS, int G>
static int run_group(const rays::DevParams& D, const rays::TraceArgs& A, int resident_blocks) {
  typedef rays::GrpGeom<G> GEO;
  const int need = (A.nray + GEO::kRaysPerBlock - 1) / GEO::kRaysPerBlock;
  const int blocks = need < resident_blocks ? (need < 1 ? 1 : need) : resident_blocks;
  gridDim.x = (unsigned)blocks;
  blockDim.x = 256;
  std::vector<double> far((size_t)rays::sg_group_far_doubles_per_lane<G>() * 256 * (size_t)blocks, 0.0);
  rays::TraceArgs A2 = A;
  A2.sg_far = far.data();
  A2.sg_far_lanes = 256ll * blocks;
  for (int b = 0; b < blocks; b++) {
    blockIdx.x = (unsigned)b;
    for (int w = 0; w < 4; w++)
      wave_emul::run_wave(64u * (unsigned)w, [&] { rays::sg_group_kernel<EQ, NS, G>(D, A2); }, threadIdx);
  }
  return 0;
}
template <int EQ, int NS>
static int run_group_g(int G, const rays::DevParams& D, const rays::TraceArgs& A, int resident_blocks) {
  if (G == 4) return run_group<EQ, NS, 4>(D, A, resident_blocks);
  if (G == 8) return run_group<EQ, NS, 8>(D, A, resident_blocks);
  if (G == 16) return run_group<EQ, NS, 16>(D, A, resident_blocks);
  return 1;
}

// SG + ray_deriv_name = 'numerical', nv = 7 only.  resident_blocks: blocks launched (fewer than the fan needs:
// finished groups pull the remaining rays).
extern "C" int rays_emul_trace_group(const rays_params_t* p, int G, int resident_blocks, int nray, const double* rvec0,
                                     const double* rindex_vec0, double* ray_vec, double* residual, int32_t* npoints,
                                     int32_t* stop_code, double* end_ray_vec, double* end_residuals, double* max_residuals) {
  if (p->ode_solver != RAYS_ODE_SG || p->ray_deriv != RAYS_DERIV_NUM || p->nv != 7 || p->multi_spec_damping) return 1;
  unsigned counter = 0;
  rays::TraceArgs A = rays::TraceArgs();
  A.nray = nray; A.rvec0 = rvec0; A.rindex_vec0 = rindex_vec0; A.ray_vec = ray_vec;
  A.residual = residual; A.npoints = npoints; A.stop_code = stop_code; A.end_ray_vec = end_ray_vec;
  A.end_residuals = end_residuals; A.max_residuals = max_residuals; A.next_ray = &counter;
  rays::DevParams D = make_dev_params(*p);
  if (int rc = attach_axisym_tables(p, D)) return rc;
  const int e = p->equilib_model | (unit_exponents(*p) ? rays::kEqUnitExp : 0), ns = p->nspec + 1;
#define RAYS_GRP_CASE(E, N) if (e == E && ns == N) return run_group_g<E, N>(G, D, A, resident_blocks);
  RAYS_GRP_CASE(0, 2) RAYS_GRP_CASE(4, 2) RAYS_GRP_CASE(0, 3) RAYS_GRP_CASE(4, 3)
  RAYS_GRP_CASE(1, 2) RAYS_GRP_CASE(5, 2) RAYS_GRP_CASE(2, 2) RAYS_GRP_CASE(6, 2)
#undef RAYS_GRP_CASE
  return 4;
}

// The one-ray-per-lane kernels on whole waves: `nwaves` blocks of one 64-lane wave each.  Fewer lanes than rays: lanes
// whose ray has ended are parked, written out and refilled by the wave's batched pass (rays_rk4_body.inc) -- the
// control flow a single emulated lane cannot exercise (its ballots have one bit).
static int g_rk4_w2_body = 0;  // run the body of the two-waves-per-SIMD build (one loop, index order, residual window)
template <int EQ, int NS, int NV>
static int run_rk4_waves(const rays::DevParams& D, const rays::TraceArgs& A, int nwaves) {
  gridDim.x = (unsigned)nwaves;
  blockDim.x = 64;
  for (int b = 0; b < nwaves; b++) {
    blockIdx.x = (unsigned)b;
    if (g_rk4_w2_body) wave_emul::run_wave(0u, [&] { rays::rk4_trace_kernel_w2<EQ, NS, 0, NV>(D, A); }, threadIdx);
    else wave_emul::run_wave(0u, [&] { rays::rk4_trace_kernel<EQ, NS, 0, NV>(D, A); }, threadIdx);
  }
  // what rays_capi.hip launches behind a tolerance kernel (this build hands ill-conditioned steps over like one:
  // emul_trace.cpp defines RAYS_EMUL_HANDOVER): rk4_resume_kernel for the rays that came back with the internal stop code
  for (int ray = 0; ray < A.nray; ray++)
    if (A.stop_code[ray] == rays::kStopResumeExact) rays::rk4_resume_ray<EQ, NS, 0, NV>(D, A, ray);
  return 0;
}
extern "C" void rays_emul_rk4_waves_use_w2_body(int on) { g_rk4_w2_body = on; }
// stride > 1: the "long rays first" hand-out order (rays_trace.hpp: take_rays) with that neighbourhood size
extern "C" int rays_emul_trace_rk4_waves(const rays_params_t* p, int nwaves, int stride, int nray, const double* rvec0,
                                         const double* rindex_vec0, double* ray_vec, double* residual, int32_t* npoints,
                                         int32_t* stop_code, double* end_ray_vec, double* end_residuals, double* max_residuals) {
  if (p->ode_solver != RAYS_ODE_RK4 || p->ray_deriv != RAYS_DERIV_COLD || p->multi_spec_damping || nwaves < 1) return 1;
  unsigned counter = 0;
  rays::TraceArgs A = rays::TraceArgs();
  A.nray = nray; A.rvec0 = rvec0; A.rindex_vec0 = rindex_vec0; A.ray_vec = ray_vec;
  A.residual = residual; A.npoints = npoints; A.stop_code = stop_code; A.end_ray_vec = end_ray_vec;
  A.end_residuals = end_residuals; A.max_residuals = max_residuals; A.next_ray = &counter;
  std::vector<unsigned> sched;
  if (stride > 1) {
    sched.assign(4 + (size_t)rays::sched_pilots((unsigned)nray, stride), 0u);
    A.sched = sched.data();
    A.sched_stride = stride;
  }
  rays::DevParams D = make_dev_params(*p);
  if (int rc = attach_axisym_tables(p, D)) return rc;
  if (p->damping_model) {
    if (g_zfun.empty()) return 2;
    D.zf_fspl = g_zfun.data(); D.zf_nx = g_zf_nx; D.zf_xmin = g_zf_xmin; D.zf_xmax = g_zf_xmax;
  }
  const int e = p->equilib_model | (unit_exponents(*p) ? rays::kEqUnitExp : 0), ns = p->nspec + 1, nv = p->nv;
#define RAYS_RK4W_CASE(E, N, V) if (e == E && ns == N && nv == V) return run_rk4_waves<E, N, V>(D, A, nwaves);
  RAYS_RK4W_CASE(0, 2, 7) RAYS_RK4W_CASE(4, 2, 7) RAYS_RK4W_CASE(1, 2, 7) RAYS_RK4W_CASE(5, 2, 7)
  RAYS_RK4W_CASE(2, 2, 8) RAYS_RK4W_CASE(6, 2, 8)
#undef RAYS_RK4W_CASE
  return 4;
}

// The one-ray-per-lane Shampine-Gordon kernel (rays_sg.hpp: sg_trace_kernel) on `nwaves` blocks of one whole wave: the
// phase / interval votes with 64 lanes, rays that end on different trips, lanes that pull the next ray in index order.
template <int EQ, int NS, int NV>
static int run_sg_waves(const rays::DevParams& D, const rays::TraceArgs& A, int nwaves) {
  gridDim.x = (unsigned)nwaves;
  blockDim.x = 64;
  std::vector<double> far((size_t)rays::sg_far_doubles_per_lane<NV>() * 64 * (size_t)nwaves, 0.0);
  rays::TraceArgs A2 = A;
  A2.sg_far = far.data();
  A2.sg_far_lanes = 64ll * nwaves;
  for (int b = 0; b < nwaves; b++) {
    blockIdx.x = (unsigned)b;
    wave_emul::run_wave(0u, [&] { rays::sg_trace_kernel<EQ, NS, 0, NV>(D, A2); }, threadIdx);
  }
  return 0;
}
extern "C" int rays_emul_trace_sg_waves(const rays_params_t* p, int nwaves, int nray, const double* rvec0,
                                        const double* rindex_vec0, double* ray_vec, double* residual, int32_t* npoints,
                                        int32_t* stop_code, double* end_ray_vec, double* end_residuals, double* max_residuals) {
  if (p->ode_solver != RAYS_ODE_SG || p->ray_deriv != RAYS_DERIV_COLD || p->multi_spec_damping || nwaves < 1) return 1;
  unsigned counter = 0;
  rays::TraceArgs A = rays::TraceArgs();
  A.nray = nray; A.rvec0 = rvec0; A.rindex_vec0 = rindex_vec0; A.ray_vec = ray_vec;
  A.residual = residual; A.npoints = npoints; A.stop_code = stop_code; A.end_ray_vec = end_ray_vec;
  A.end_residuals = end_residuals; A.max_residuals = max_residuals; A.next_ray = &counter;
  rays::DevParams D = make_dev_params(*p);
  if (int rc = attach_axisym_tables(p, D)) return rc;
  if (p->damping_model) {
    if (g_zfun.empty()) return 2;
    D.zf_fspl = g_zfun.data(); D.zf_nx = g_zf_nx; D.zf_xmin = g_zf_xmin; D.zf_xmax = g_zf_xmax;
  }
  const int e = p->equilib_model | (unit_exponents(*p) ? rays::kEqUnitExp : 0), ns = p->nspec + 1, nv = p->nv;
#define RAYS_SGW_CASE(E, N, V) if (e == E && ns == N && nv == V) return run_sg_waves<E, N, V>(D, A, nwaves);
  RAYS_SGW_CASE(1, 2, 7) RAYS_SGW_CASE(5, 2, 7) RAYS_SGW_CASE(2, 2, 8) RAYS_SGW_CASE(6, 2, 8)
#undef RAYS_SGW_CASE
  return 4;
}
