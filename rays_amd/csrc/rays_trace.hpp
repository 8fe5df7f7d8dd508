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
//   * recorded points are staged per wave in LDS (K points x (nv+1) doubles x 64 lanes, rows
//     padded to 65 doubles) and flushed as contiguous runs of up to K*nv*8 bytes per ray into the
//     reference layout ray_vec(nv, nstep_max+1, nray) / residual(nstep_max+1, nray)
//     (ray_results_m.f90:44-46), instead of 64 scattered 56-byte stores per step.
#pragma once

#include "rays_device.hpp"

namespace rays {

#ifndef RAYS_FLUSH_G
#define RAYS_FLUSH_G 4
#endif
constexpr int kWave = 64;
constexpr int kRowStride = 65;  // doubles; 65 -> conflict-free b64 column reads at flush
#ifdef RAYS_HOST_EMUL  // tests/hip_emul: one lane stands in for the whole wave
constexpr int kFlushStride = 1;
#else
constexpr int kFlushStride = kWave;
#endif

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
// only needed where a ray starts or ends and where staged points are flushed; as plain kernel
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

// Per-wave LDS staging of recorded trajectory points.
template <int NV, int K>
struct PointStage {
  static constexpr int kRows = K * (NV + 1);
  static constexpr int kDoublesPerWave = kRows * kRowStride;
  double* base;  // this wave's region
  int lane;

  RAYS_DEV void put(int slot, const double v[NV], double resid) {
    double* p = base + (slot * (NV + 1)) * kRowStride + lane;
#pragma unroll
    for (int c = 0; c < NV; c++) p[c * kRowStride] = v[c];
    p[NV * kRowStride] = resid;
  }

  // One lane writes out its own staged points (ray termination).
  RAYS_DEV void drain_own(const TraceArgs& A_, int nbuf, long long first_pt) {
    const TraceArgs& A = cold_args(A_);
    for (int k = 0; k < nbuf; k++) {
      const double* p = base + (k * (NV + 1)) * kRowStride + lane;
#pragma unroll
      for (int c = 0; c < NV; c++) A.ray_vec[(first_pt + k) * NV + c] = p[c * kRowStride];
      A.residual[first_pt + k] = p[NV * kRowStride];
    }
  }

  // All 64 lanes call this together.  nbuf = points this lane has staged, first_pt = global point
  // index (ray*(nstep_max+1) + step) of its slot 0.  Four rays per pass: their LDS column reads are
  // issued together and waited for once (with one ray per pass the wave sat out one LDS latency
  // per ray, 64 times per flush).
  RAYS_DEV void flush(const TraceArgs& A_, int nbuf, long long first_pt) {
    const TraceArgs& A = cold_args(A_);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int PER = NV + 1;                                      // doubles per point
    constexpr int E = (K * PER + kFlushStride - 1) / kFlushStride;   // elements per lane and ray
    constexpr int G = RAYS_FLUSH_G;                                  // rays per pass
#pragma unroll 1
    for (int r0 = 0; r0 < kWave; r0 += G) {
      int total[G];
      long long pt0[G];
      double val[G][E];
#pragma unroll
      for (int j = 0; j < G; j++) {
        const int r = r0 + j;
        total[j] = __builtin_amdgcn_readlane(nbuf, r) * PER;
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)(first_pt & 0xffffffffll), r);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)first_pt >> 32), r);
        pt0[j] = (long long)(((unsigned long long)hi << 32) | lo);
      }
#pragma unroll
      for (int j = 0; j < G; j++)
#pragma unroll
        for (int q = 0; q < E; q++) {
          const int e = lane + q * kFlushStride;
          val[j][q] = e < total[j] ? base[e * kRowStride + r0 + j] : 0.;
        }
#pragma unroll
      for (int j = 0; j < G; j++)
#pragma unroll
        for (int q = 0; q < E; q++) {
          const int e = lane + q * kFlushStride;
          if (e < total[j]) {
            const int k = e / PER, c = e - k * PER;
            if (c < NV)
              A.ray_vec[(pt0[j] + k) * NV + c] = val[j][q];
            else
              A.residual[pt0[j] + k] = val[j][q];
          }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

}  // namespace rays
