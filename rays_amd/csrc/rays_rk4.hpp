// rays_rk4.hpp -- RK4 ray-trace kernel (ode_solver_name = 'RK4_ODE').
//
// Reference path restated here:
//   trace_rays   ray_tracing.f90:67-264   (per-ray bookkeeping, stop logic, recording)
//   RK4_ode      RK4_ode_m.f90:59-94      (4 stages, early return leaves v untouched)
//   eqn_ray / check_save                  (rays_device.hpp)
//
// Per lane the integrator is a 4-state machine around ONE RHS evaluation per wave-loop trip:
//   stage 0,1,2 : evaluate f at w (= v + ds*f1/2, v + ds*f2/2, v + ds*f3)  -> f2, f3, f4
//   stage 3     : w = v + ds*(f1 + 2 f2 + 2 f3 + f4)/6 ; check_save(w) fused with the next step's
//                 f1 = eqn_ray(w) (same equilibrium + dispersion derivatives, evaluated once)
// A new ray starts in stage 3 with w = v0 (`first`), which is exactly the reference's initial
// check_save call (ray_tracing.f90:100) and also yields the first step's f1.
//
// Recorded points: the one-wave-per-SIMD kernel with nv = 7 passes them through a per-lane LDS window
// and writes whole 64-byte sectors (PointWindow, rays_trace.hpp: HBM write traffic 1.06x the
// algorithmic bytes for 1.3 % of the pass; with nv = 8 only residual(:) needs it, a ray_vec record being
// one sector); the other builds store each point directly from its lane
// (record_point: 2.1x the bytes through partly written sectors, no LDS).  The kernel is bound by
// instruction issue, not by HBM, so what matters is the issue slots the recording
// costs: ~50 per step for the window, 14 for direct stores, ~500 for an earlier cooperative flush of
// 512-byte runs (12 % of the pass).  -DRAYS_RK4_DIRECT_STORES builds the kernel without the window.
#pragma once

#include "rays_trace.hpp"

namespace rays {

#ifndef RAYS_HOST_EMUL
// One wave per SIMD (all 256 VGPRs): fastest while the fan has at most one wave per SIMD (<= 64k rays).
template <int EQ, int NS, int DERIV, int NV>
__global__ void __launch_bounds__(256)
rk4_trace_kernel(const DevParams P_kernarg, const TraceArgs A_hot) {
#ifdef RAYS_RK4_DIRECT_STORES
#define RAYS_RK4_USE_WINDOW 0
#else
#define RAYS_RK4_USE_WINDOW 1
#endif
#define RAYS_RK4_LONG_FIRST 1  // pass loop around the trip loop, rays handed out long-first (rays_rk4_body.inc)
#include "rays_rk4_body.inc"
#undef RAYS_RK4_LONG_FIRST
#undef RAYS_RK4_USE_WINDOW
}

// Two waves per SIMD (<= 256 combined registers): from two waves' worth of rays on this build wins,
// because two waves share a SIMD's issue slots better than one (tools/ubench: 4.6 vs 5.5 clocks per
// FP64 op).  Measured on the Solovev fan: -2 % at 64k rays, +6 % at 128k, +18 % at 1M.
// rays_capi.hip: find_kernel picks by fan size.  (The launch bound must be a literal: hipcc ignores
// a template-dependent second argument.)
template <int EQ, int NS, int DERIV, int NV>
__global__ void __launch_bounds__(256, 2)
rk4_trace_kernel_w2(const DevParams P_kernarg, const TraceArgs A_hot) {
#ifdef RAYS_RK4_W2_DIRECT_STORES
#define RAYS_RK4_USE_WINDOW 0
#else
#define RAYS_RK4_USE_WINDOW 2  // residual(:) through a 30 KB window, ray_vec stored directly (rays_trace.hpp)
#endif
#define RAYS_RK4_LONG_FIRST 0  // one loop, rays in index order: what fits this build's 256 registers (rays_rk4_body.inc)
#include "rays_rk4_body.inc"
#undef RAYS_RK4_LONG_FIRST
#undef RAYS_RK4_USE_WINDOW
}
#else
// (host emulation: the same two bodies, for the CPU test tier)
template <int EQ, int NS, int DERIV, int NV>
void rk4_trace_kernel_w2(const DevParams P_kernarg, const TraceArgs A_hot) {
#define RAYS_RK4_USE_WINDOW 2
#define RAYS_RK4_LONG_FIRST 0
#include "rays_rk4_body.inc"
#undef RAYS_RK4_LONG_FIRST
#undef RAYS_RK4_USE_WINDOW
}
template <int EQ, int NS, int DERIV, int NV>
void rk4_trace_kernel(const DevParams P_kernarg, const TraceArgs A_hot) {
#ifdef RAYS_RK4_DIRECT_STORES
#define RAYS_RK4_USE_WINDOW 0
#else
#define RAYS_RK4_USE_WINDOW 1
#endif
#define RAYS_RK4_LONG_FIRST 1  // pass loop around the trip loop, rays handed out long-first (rays_rk4_body.inc)
#include "rays_rk4_body.inc"
#undef RAYS_RK4_LONG_FIRST
#undef RAYS_RK4_USE_WINDOW
}
#endif

}  // namespace rays
