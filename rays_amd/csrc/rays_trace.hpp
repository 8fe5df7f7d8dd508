// rays_trace.hpp -- the batched ray launcher shared by the RK4 and SG kernels.
//
// Replaces the reference's outer loops (ray_tracing.f90:62-264: OpenMP `parallel do` over rays,
// sequential `trajectory` loop over output steps) with persistent wave64 workers:
//
//   * one ray per lane; the ODE state (v, stage accumulator, next RHS input) lives in VGPRs;
//   * every trip of the wave loop performs exactly ONE evaluation of the ray-equation RHS for all
//     live lanes (the expensive, convergent part); the integrator around it is a small per-lane
//     state machine, so lanes in different stages / different rays never serialise the RHS;
//   * a lane whose ray terminates pulls the next ray index from a global counter (lane refill),
//     so a wave stays full while rays of very different length (99..380 steps in the Solovev
//     fan) are in flight;
//   * a recorded point is written by its lane straight into the reference layout
//     ray_vec(nv, nstep_max+1, nray) / residual(nstep_max+1, nray) (ray_results_m.f90:44-46).
#pragma once

#include "rays_device.hpp"

namespace rays {

constexpr int kWave = 64;

struct TraceArgs {
  int nray;
  const double* __restrict__ rvec0;        // [nray][3]
  const double* __restrict__ rindex_vec0;  // [nray][3]
  double* __restrict__ ray_vec;            // [nray][nstep_max+1][nv]
  double* __restrict__ residual;           // [nray][nstep_max+1]
  int* __restrict__ npoints;               // [nray]
  int* __restrict__ stop_code;             // [nray]
  double* __restrict__ end_ray_vec;        // [nray][nv]      (may be null)
  double* __restrict__ end_residuals;      // [nray]          (may be null)
  double* __restrict__ max_residuals;      // [nray]          (may be null)
  unsigned int* __restrict__ next_ray;     // refill counter, zeroed before launch
};

// The trace kernels' signature is (DevParams, TraceArgs).  The eleven pointers of TraceArgs are
// only needed where a ray starts or ends (the two trajectory arrays: where a point is recorded); as plain kernel
// arguments they are loaded once and then occupy 24 of the ~100 SGPRs across the whole wave loop,
// which the hot RHS pays for with spill reloads (v_readlane_b32).  cold_args() re-reads the block
// from the kernarg segment at the point of use instead (scalar loads behind an opaque zero offset,
// so LLVM cannot hoist them back out of the loop).
#ifdef RAYS_HOST_EMUL
RAYS_DEV const TraceArgs& cold_args(const TraceArgs& A) { return A; }
#else
RAYS_DEV const TraceArgs& cold_args(const TraceArgs&) {
  typedef const __attribute__((address_space(4))) char* kconst_ptr;
  constexpr unsigned kOffset = (sizeof(DevParams) + alignof(TraceArgs) - 1) / alignof(TraceArgs) * alignof(TraceArgs);
  unsigned z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  kconst_ptr kb = (kconst_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  return *reinterpret_cast<const TraceArgs*>((const char*)(kb + kOffset + z));
}
#endif

// One recorded trajectory point, written by its own lane: ray_vec(:, pt) and residual(pt).
template <int NV>
RAYS_DEV void record_point(const TraceArgs& A, long long pt, const double v[NV], double resid) {
#pragma unroll
  for (int c = 0; c < NV; c++) A.ray_vec[pt * NV + c] = v[c];
  A.residual[pt] = resid;
}

}  // namespace rays
