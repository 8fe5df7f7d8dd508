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
// Recorded points are written by their own lane, eight 8-byte stores per step (record_point).  An
// earlier version staged 8 points per lane in LDS and flushed them as 512-byte runs per ray; the
// kernel is bound by instruction issue, not by HBM, and the flush's ~500 issue
// slots per step cost 12 % of the 64k-fan pass against 14 for the direct stores (measured:
// 5.05 -> 4.49 ms; HBM write traffic per launch in profiles/).
#pragma once

#include "rays_trace.hpp"

namespace rays {

// The kernel body; the two __global__ wrappers below differ only in their launch bounds.
template <int EQ, int NS, int DERIV, int NV>
RAYS_DEV void rk4_trace_body(const DevParams& P_kernarg, const TraceArgs& A_hot) {
  DevParams P;  // working copy: scalarised by the compiler, hot constants in vector registers
  hot_params<EQ, NS>(P_kernarg, P);

  const unsigned total_lanes = gridDim.x * blockDim.x;
  const long long npt = (long long)P.nstep_max + 1;

  // ---- per-lane ray state -------------------------------------------------------------------
  int ray = blockIdx.x * blockDim.x + threadIdx.x;
  bool alive = ray < A_hot.nray;
  bool need_init = alive;
  int j = 3;            // stage
  int first = 1;        // stage-3 evaluation is the initial check_save (int, not bool: see rays_sg.hpp)
  int nstep = 0;
  double s = 0., sout = 0., dsl = 0.;
  double v[NV], w[NV], acc[NV];
  double last_resid = 0., prev_resid = 0., maxr = -1.7976931348623157e308;
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = w[i] = acc[i] = 0.;

  const Recip R6 = const_recip(6.0, 1.0 / 6.0);  // RN(1/6); div() = the correctly rounded quotient

  while (__any(alive)) {
    if (need_init) {  // initialize_ode_vector + per-ray resets (ray_tracing.f90:77-93)
      const TraceArgs& A = cold_args(A_hot);
      initialize_ode_vector<EQ, NS, NV>(P, A.rvec0 + 3ll * ray, A.rindex_vec0 + 3ll * ray, v);
#pragma unroll
      for (int i = 0; i < NV; i++) w[i] = v[i];
      j = 3;
      first = 1;
      nstep = 0;
      s = 0.;
      sout = 0.;
      last_resid = 0.;
      prev_resid = 0.;
      maxr = -1.7976931348623157e308;
      need_init = false;
    }

    // ---- the one RHS evaluation of this trip -------------------------------------------------
    double f[NV], resid = 0.;
    int code = 0, cs_flag = 0;
    bool cs_stop = false;
    if (alive) rhs_eval<EQ, NS, DERIV, NV>(P, w, j == 3, resid, cs_flag, cs_stop, code, f);

    // ---- per-lane integrator state machine ---------------------------------------------------
    int stop = 0;        // 0 = keep going
    int done = 0;        // ray finished this trip
    if (alive) {
      if (j < 3) {
        if (code) {  // RK4_ode_m.f90:83-89: stage stopped, v untouched
          stop = code;
          done = 1;
        } else if (j == 0) {
#pragma unroll
          for (int i = 0; i < NV; i++) {
            acc[i] = acc[i] + 2.0 * f[i];
            w[i] = v[i] + dsl * f[i] * 0.5;  // (ds*f2)/2.0, exact scaling
          }
          j = 1;
        } else if (j == 1) {
#pragma unroll
          for (int i = 0; i < NV; i++) {
            acc[i] = acc[i] + 2.0 * f[i];
            w[i] = v[i] + dsl * f[i];
          }
          j = 2;
        } else {
#pragma unroll
          for (int i = 0; i < NV; i++) {
            acc[i] = acc[i] + f[i];
            w[i] = v[i] + div(dsl * acc[i], R6);  // RK4_ode_m.f90:91  (ds*(...))/6.0
          }
          j = 3;
        }
      } else {
        // stage 3: w is the new state; check_save decides whether the step is recorded
        if (first) {
          // ray_tracing.f90:92-112: point 1 = initial state, residual(1) = 0
          record_point<NV>(A_hot, (long long)ray * npt, v, 0.);
          if (cs_stop) {  // ray did not start: npoints = 1, summary fields stay zero
            const TraceArgs& A = cold_args(A_hot);
            A.npoints[ray] = 1;
            A.stop_code[ray] = cs_flag;
            if (A.end_ray_vec)
#pragma unroll
              for (int i = 0; i < NV; i++) A.end_ray_vec[(long long)ray * NV + i] = 0.;
            if (A.end_residuals) A.end_residuals[ray] = 0.;
            if (A.max_residuals) A.max_residuals[ray] = 0.;
            done = 1;
            stop = -1;  // summary already written
          }
          first = 0;
        } else {
#pragma unroll
          for (int i = 0; i < NV; i++) v[i] = w[i];  // RK4_ode_m.f90:91-92
          s = sout;
          if (cs_stop) {  // ray_tracing.f90:214-234: step not recorded, v is the new state
            stop = cs_flag;
            done = 1;
          } else {  // :237-243
            nstep = nstep + 1;
            record_point<NV>(A_hot, (long long)ray * npt + nstep, v, resid);
            if (fabs(last_resid) > maxr) maxr = fabs(last_resid);
            prev_resid = last_resid;
            last_resid = resid;
          }
        }
        if (!done) {  // top of the next trajectory trip, ray_tracing.f90:118-172
          s = sout;
          sout = sout + P.ds;
          if (sout > P.s_max) {
            stop = RAYS_STOP_SOUT_GT_SMAX;
            done = 1;
          } else if (nstep + 1 > P.nstep_max) {
            stop = RAYS_STOP_NSTEP_MAX;
            done = 1;
          } else if (code) {  // first RK4 stage of the next step stops (RK4_ode_m.f90:82-83)
            stop = code;
            done = 1;
          } else {
            dsl = sout - s;  // RK4_ode_m.f90:81
#pragma unroll
            for (int i = 0; i < NV; i++) {
              acc[i] = f[i];
              w[i] = v[i] + dsl * f[i] * 0.5;
            }
            j = 0;
          }
        }
      }
      if (done) {
        if (stop >= 0) {  // ray_tracing.f90:252-260
          const TraceArgs& A = cold_args(A_hot);
          A.npoints[ray] = nstep + 1;
          A.stop_code[ray] = stop;
          if (A.end_ray_vec)
#pragma unroll
            for (int i = 0; i < NV; i++) A.end_ray_vec[(long long)ray * NV + i] = v[i];
          if (A.end_residuals) A.end_residuals[ray] = nstep >= 1 ? prev_resid : 0.;
          if (A.max_residuals) A.max_residuals[ray] = maxr;
        }
      }
    }

    // ---- refill finished lanes ---------------------------------------------------------------
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

#ifndef RAYS_HOST_EMUL
// One wave per SIMD (all 256 VGPRs): fastest while the fan has at most one wave per SIMD (<= 64k rays).
template <int EQ, int NS, int DERIV, int NV>
__global__ void __launch_bounds__(256)
rk4_trace_kernel(const DevParams P, const TraceArgs A) {
  rk4_trace_body<EQ, NS, DERIV, NV>(P, A);
}

// Two waves per SIMD (128 VGPRs, spills): from two waves' worth of rays on this build wins, because
// two waves share a SIMD's issue slots better than one (tools/ubench: 4.6 vs 5.5 clocks per FP64 op).
// Measured on the Solovev fan: -2 % at 64k rays, +4 % at 128k, +11 % at 1M.  rays_capi.hip:
// find_kernel picks by fan size.  (The launch bound must be a literal: hipcc ignores a
// template-dependent second argument.)
template <int EQ, int NS, int DERIV, int NV>
__global__ void __launch_bounds__(256, 2)
rk4_trace_kernel_w2(const DevParams P, const TraceArgs A) {
  rk4_trace_body<EQ, NS, DERIV, NV>(P, A);
}
#else
template <int EQ, int NS, int DERIV, int NV>
void rk4_trace_kernel(const DevParams P, const TraceArgs A) {
  rk4_trace_body<EQ, NS, DERIV, NV>(P, A);
}
#endif

}  // namespace rays
