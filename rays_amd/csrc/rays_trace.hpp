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
//   * a recorded point is written by its lane into the reference layout
//     ray_vec(nv, nstep_max+1, nray) / residual(nstep_max+1, nray) (ray_results_m.f90:44-46), directly
//     (record_point) or through a per-lane LDS window that emits whole 64-byte sectors (PointWindow).
#pragma once

#include "rays_device.hpp"

namespace rays {

constexpr int kWave = 64;
// Internal stop code of the tolerance-flavour RK4 kernels: "this ray's next step is ill-conditioned; rk4_resume_kernel
// continues it in the reference's arithmetic" (rays_rk4_body.inc).  Never leaves the library: every launch of such a
// kernel is followed by the resume kernel, which overwrites it with the ray's real stop code.
constexpr int kStopResumeExact = 1000;

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
  // Optional per-ray starting conditions (null = the reference's: v0 from rvec0 / rindex_vec0, s = 0):
  const double* __restrict__ v0;           // [nray][nv]  full ODE vectors to start from (rays_hip_ode_step_device)
  const double* __restrict__ s0;           // [nray]      ray parameter at the start
  // Fused ray_scan (ray_scan.f90:33-49): run r = ray / rays_per_run traces fan member ray % rays_per_run
  // with step ds_run[r]; rays_per_run = 0: one run, ds from the parameter block.
  const double* __restrict__ ds_run;       // [nrun]
  int rays_per_run;
  // SG kernels: workspace for the integrator's rarely used upper storage tiers (rays_sg.hpp), [doubles per
  // lane][sg_far_lanes]; the launcher never starts more lanes than sg_far_lanes
  double* __restrict__ sg_far;
  long long sg_far_lanes;
  // RK4 kernels, fans of several rays per lane: hand-out order "long rays first" (take_rays below).  sched_stride = S
  // > 1: sched = [sweep-1 cursor, sweep-2 cursor, 0, 0, state of the sched_pilots(nray, S) neighbourhoods], zeroed before
  // the launch; 0: rays are handed out in index order by next_ray alone.
  unsigned int* __restrict__ sched;
  int sched_stride;
};

// ---- which ray a free lane traces next ---------------------------------------------------------------------------
// With more rays than lanes the pass ends when the last ray does, so the rays handed out last should be the short
// ones -- but a ray's length is only known once it has been traced.  Neighbouring rays of a fan are alike, though
// (cfg 5b: standard deviation 3 steps within 64 consecutive rays, 28..54 between such blocks), so one ray in S is
// traced first, as the PILOT of its neighbourhood (S - 1 rays close to it), and the others are handed out in two
// sweeps over the neighbourhoods in index order: sweep 1 skips the neighbourhoods whose pilot has already ended (the
// short ones) and hands out those whose pilot is still running (at that moment the longest rays there are); sweep 2
// hands out what sweep 1 skipped.  tools/refill_model.py on cfg 5b's measured lengths: 773 trips in index order, 589 with
// this order and S = 2 (605 with S = 4 or 8), 589 with the rays sorted by their true lengths (ideal 518).  Every ray is handed out exactly
// once: a pilot by the counter, a neighbour by whoever sets its bit in the neighbourhood's state word first.  The
// order changes no ray's arithmetic.
//
// take_rays is called by ALL lanes of a wave from wave-uniform control flow (`want`: this lane needs a ray).  The
// lanes that want one are served by ONE atomic of the wave per cursor and round (the first of them adds their number,
// each takes base + its rank): a word of HBM-side atomics serves ~90 returning atomics per microsecond, and a fan
// whose rays end together (cfg 4: 131072 lanes, eight times) otherwise queues a whole generation behind it.
// Returns the lane's ray or -1 (none this time: ask again at the next pass); `dry` (wave-uniform) = no ray is left.
constexpr unsigned kSchedDone = 0x80000000u;
// Layout: the fan is cut into blocks of 64 S consecutive rays, S rows of 64; row 0 holds the block's 64 pilots, and
// neighbourhood m = (block, column l) is column l of the block: its pilot and the S - 1 rays 64, 128, ... behind it.
// So the 64 lanes of a wave hold 64 CONSECUTIVE rays whenever they ask together, as in index order (with pilots
// S apart a wave's trajectories spread over S times the address range: the 1 M-ray fan ran 2-5 % slower).
__host__ __device__ inline unsigned sched_pilots(unsigned nray, int S) { return (nray + 64u * (unsigned)S - 1u) / (64u * (unsigned)S) * 64u; }
__host__ __device__ inline unsigned sched_pilot_ray(unsigned m, int S) { return (m >> 6) * 64u * (unsigned)S + (m & 63u); }
__host__ __device__ inline bool sched_is_pilot(unsigned ray, int S) { return ((ray >> 6) % (unsigned)S) == 0u; }
__host__ __device__ inline unsigned sched_neighbourhood(unsigned ray, int S) { return ray / (64u * (unsigned)S) * 64u + (ray & 63u); }
RAYS_DEV unsigned wave_take(unsigned int* word, bool want, unsigned long long mask, int lane) {
  // the lanes of `mask` get consecutive values of *word (the wave adds their number once)
  const int n = __popcll(mask);
  int leader = 0;
  while (!((mask >> leader) & 1ull)) leader++;
  unsigned base = 0;
  if (want && lane == leader) base = atomicAdd(word, (unsigned)n);
  base = (unsigned)__shfl((int)base, leader);
  return base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
}
RAYS_DEV int take_rays(const TraceArgs& A, unsigned total_lanes, int S, bool want, int lane, bool& dry) {
  const unsigned nray = (unsigned)A.nray;
  int got = -1;
  dry = false;
  unsigned long long mask = __ballot(want);
  if (!mask) return got;
  if (S <= 1) {  // index order
    const unsigned nxt = wave_take(A.next_ray, want, mask, lane) + total_lanes;
    if (want && nxt < nray) got = (int)nxt;
    dry = __any(want && got < 0);  // the counter is past the last ray
    return got;
  }
  const unsigned M = sched_pilots(nray, S);                   // neighbourhoods = pilots
  const unsigned first = total_lanes < M ? total_lanes : M;   // pilots the lanes started with
  if (first < M) {
    const unsigned p = wave_take(A.next_ray, want, mask, lane) + first;
    if (want && p < M && sched_pilot_ray(p, S) < nray) got = (int)sched_pilot_ray(p, S);
    mask = __ballot(want && got < 0);
    if (!mask) return got;
  }
  unsigned int* state = A.sched + 4;
  const unsigned W = (unsigned)S - 1u, Q = M * W;  // neighbour slots
  int rounds = 6;
  for (int sweep = 0; sweep < 2; sweep++) {
    for (;;) {
      if (--rounds < 0) return got;
      const bool w = want && got < 0;
      const unsigned q = wave_take(A.sched + sweep, w, mask, lane);
      if (w && q < Q) {
        // consecutive slots are consecutive rays: slot -> (block of 64 S rays, row k = 1..S-1 of it, column l)
        const unsigned blk = q >> 6, l = q & 63u, B = blk / W, k = blk - B * W + 1u;
        const unsigned m = (B << 6) + l, r = ((B * (unsigned)S + k) << 6) + l;
        // sweep 1 leaves the neighbourhoods whose pilot has ended (the short ones) to sweep 2
        if (r < nray && !(sweep == 0 && (*(volatile unsigned int*)(state + m) & kSchedDone))) {
          const unsigned old = atomicOr(state + m, 1u << k);
          if (!(old & (1u << k))) got = (int)r;
        }
      }
      const bool past = __any(w && q >= Q);  // this cursor has run off its end
      mask = __ballot(want && got < 0);
      if (!mask) return got;
      if (past) break;
    }
  }
  dry = true;  // both sweeps are through and lanes still want: nothing is left
  return got;
}

// Start of a ray: initialize_ode_vector (or the caller's v0), the ray parameter and the run's step.
template <int EQ, int NS, int NV>
RAYS_DEV void start_ray(const DevParams& P, const TraceArgs& A, int ray, double v[NV], double& s_start, double& ds_ray) {
  int member = ray;
  ds_ray = P.ds;
  if (RAYS_RARE(A.rays_per_run > 0)) {
    const int run = ray / A.rays_per_run;
    member = ray - run * A.rays_per_run;
    ds_ray = A.ds_run[run];
  }
  if (RAYS_RARE(A.v0 != nullptr)) {
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = A.v0[(long long)ray * NV + i];
  } else {
    initialize_ode_vector<EQ, NS, NV>(P, A.rvec0 + 3ll * member, A.rindex_vec0 + 3ll * member, v);
  }
  s_start = A.s0 ? A.s0[ray] : 0.;
}

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

// The same for the parameter block (the kernels' first argument): re-derived where a phase of the trip starts, its
// constants are loaded there (scalar loads the other waves of the SIMD hide) instead of being kept in -- and spilled
// from -- scalar registers across the whole wave loop.  For the kernels built for several waves per SIMD.
#ifdef RAYS_HOST_EMUL
RAYS_DEV const DevParams& cold_params(const DevParams& P) { return P; }
#else
RAYS_DEV const DevParams& cold_params(const DevParams&) {
  typedef const __attribute__((address_space(4))) char* kconst_ptr;
  unsigned z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  kconst_ptr kb = (kconst_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  return *reinterpret_cast<const DevParams*>((const char*)(kb + z));
}
#endif

// One recorded trajectory point, written by its own lane: ray_vec(:, pt) and residual(pt).
template <int NV>
RAYS_DEV void record_point(const TraceArgs& A, long long pt, const double v[NV], double resid) {
#pragma unroll
  for (int c = 0; c < NV; c++) A.ray_vec[pt * NV + c] = v[c];
  A.residual[pt] = resid;
}

// ---- PointWindow: recorded points leave the chip as whole 64-byte sectors ------------------------
// record_point's stores are 56 + 8 bytes per step and lane, to addresses no other lane shares.  The
// L2 does not keep such a partly written sector until the lane's next step (9 us later) completes it:
// it is written back in between, so HBM sees most sectors two or three times (tools/ubench/
// store_pattern.hip: 2.4x the bytes; 2.1x measured in the RK4 kernel).  PointWindow holds a lane's
// last recorded doubles in LDS and writes a sector only once all eight of its doubles exist.
//
// Sector boundaries are those of the global address, so they fall differently for every ray (a ray's
// slab starts at ray * 56056 bytes): `phase` = the slab's offset into its sector, in doubles.  The
// window is laid out in SECTOR coordinates: row u holds the double that goes to (sector-aligned)
// g[u], where g = slab + 56*(k-1) - phase for the k-th group of eight points.  A point p of the group
// therefore lands on rows 7*(p & 7) + phase + (0..6), i.e. 0..62; after the eighth point rows 0..55
// are seven complete sectors, written with 16-byte stores from fixed rows, and rows 56..62 (the
// doubles beyond the last boundary) move down to 0..6 as the next group's head.  residual(:) gets the
// same treatment with one sector per group.  Only a ray's first sector (shared with the previous
// ray's slab) and its last points go out as single doubles.  Rows are row-major over the block's
// lanes: every LDS access is conflict free whatever row each lane is at.
#ifdef RAYS_HOST_EMUL
typedef double* trace_lds_ptr;
#else
typedef __attribute__((address_space(3))) double* trace_lds_ptr;
#endif
struct alignas(16) SectorPair { double a, b; };

// NV = 7: both streams pass through the window.  NV = 8: a ray_vec record IS one sector (the slabs are
// 64-byte aligned when the array is), so only residual(:) does.  Other NV: no window (kAny = false).
// RES_ONLY (the two-waves-per-SIMD build, whose two blocks per CU cannot both hold the 156 KB window): residual(:)
// alone goes through the window (30 KB) -- it is the stream that suffers most from partly written sectors, eight
// bytes per step into a sector that is otherwise written back eight times -- and ray_vec is stored directly.
template <int NV, bool RES_ONLY = false>
struct PointWindow {
  static constexpr bool kVec = NV == 7 && !RES_ONLY, kRes = NV == 7 || NV == 8, kAny = kRes;
  static constexpr int kVecRows = kVec ? 63 : 0, kResRows = kRes ? 15 : 0;
  static constexpr int kStride = 256;  // lanes per block
  static constexpr size_t kLdsBytes = (size_t)(kVecRows + kResRows) * kStride * sizeof(double);
  trace_lds_ptr vec, res;  // this lane's columns

  RAYS_DEV void attach(double* lds, int lane_in_block) {
    vec = (trace_lds_ptr)(lds + lane_in_block);
    res = (trace_lds_ptr)(lds + (size_t)kVecRows * kStride + lane_in_block);
  }
  // offsets of ray's slabs into their sectors, in doubles (both < 8)
  RAYS_DEV static void phases(const TraceArgs& A, int ray, long long npt, int& pv, int& pr) {
    const unsigned bv = (unsigned)((unsigned long long)A.ray_vec >> 3), br = (unsigned)((unsigned long long)A.residual >> 3);
    pv = (int)((bv + (unsigned)ray * (unsigned)((npt * NV) & 7)) & 7u);
    pr = (int)((br + (unsigned)ray * (unsigned)(npt & 7)) & 7u);
  }
  // point pt of the ray whose slabs start at point index pt0
  RAYS_DEV void put(const TraceArgs& A, long long pt0, int pt, int pv, int pr, const double v[NV], double resid) {
    if constexpr (kVec) {
      const trace_lds_ptr p = vec + ((pt & 7) * NV + pv) * kStride;
#pragma unroll
      for (int c = 0; c < NV; c++) p[c * kStride] = v[c];
    } else {
#pragma unroll
      for (int c = 0; c < NV; c++) A.ray_vec[(pt0 + pt) * NV + c] = v[c];
    }
    res[((pt & 7) + pr) * kStride] = resid;
  }
  // one sector from rows r0..r0+7 of col to g (64-byte aligned); `from` > 0: only elements >= from
  RAYS_DEV static void sector(double* g, trace_lds_ptr col, int r0, int from) {
    double e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = col[(r0 + i) * kStride];
    if (RAYS_RARE(from > 0)) {
#pragma unroll
      for (int i = 1; i < 8; i++)
        if (i >= from) g[i] = e[i];
    } else {
      SectorPair* o = reinterpret_cast<SectorPair*>(g);
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] = SectorPair{e[2 * i], e[2 * i + 1]};
    }
  }
  // after the put of point 8k-1: write the group's complete sectors, keep the rest as the next head
  RAYS_DEV void flush(const TraceArgs& A, long long pt0, int k, int pv, int pr) {
    if constexpr (kVec) {
      double* g = A.ray_vec + pt0 * NV + (56 * (k - 1) - pv);
      sector(g, vec, 0, k == 1 ? pv : 0);  // a ray's first sector belongs in part to the ray before it
#pragma unroll
      for (int j = 1; j < 7; j++) sector(g + 8 * j, vec, 8 * j, 0);
#pragma unroll
      for (int i = 0; i < 7; i++) vec[i * kStride] = vec[(56 + i) * kStride];
    }
    sector(A.residual + pt0 + (8 * (k - 1) - pr), res, 0, k == 1 ? pr : 0);
#pragma unroll
    for (int i = 0; i < 7; i++) res[i * kStride] = res[(8 + i) * kStride];
  }
  // the ray ended with n points: write what the window still holds.  Runs once per pass of the wave over its
  // parked lanes (rays_rk4_body.inc), in chunks of eight rows: the LDS reads of a chunk are in flight together
  // (element by element it was one LDS round trip per double, up to 56 of them per ray).
  RAYS_DEV void finish(const TraceArgs& A, long long pt0, int n, int pv, int pr) {
    const int K = n >> 3, m = n & 7;
    if constexpr (kVec) {
      double* g = A.ray_vec + pt0 * NV + (56 * K - pv);
      const int u0 = K ? 0 : pv, u1 = pv + NV * m;  // u1 <= 7 + 49: rows 0..55, a chunk reads up to row 63
      static_assert(kVecRows + kResRows >= 64, "finish reads whole chunks of eight rows");
#pragma nounroll
      for (int b = u0 & ~7; b < u1; b += 8) {
        double e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = vec[(b + i) * kStride];
#pragma unroll
        for (int i = 0; i < 8; i++)
          if (b + i >= u0 && b + i < u1) g[b + i] = e[i];
      }
    }
    double* gr = A.residual + pt0 + (8 * K - pr);
    const int u0 = K ? 0 : pr, u1 = pr + m;  // u1 <= 14
    double e[kResRows];
#pragma unroll
    for (int i = 0; i < kResRows; i++) e[i] = res[i * kStride];
#pragma unroll
    for (int i = 0; i < kResRows; i++)
      if (i >= u0 && i < u1) gr[i] = e[i];
  }
};

// LDS for the staged 1-D spline tables of the eqdsk equilibrium (DevParams::a_lds_tab), behind the point
// window: kernels whose window leaves room (nv = 8: residual(:) only; nv = 12 | 13: none) get 32 KB (+ a pad that
// keeps the block off LDS address 0, which means "not staged").
constexpr size_t kEqTabPad = 64, kEqTabBytes = 32768;
template <int EQ, int NV>
constexpr size_t eq_tab_lds_bytes() {
  return ((EQ & 3) == RAYS_EQ_AXISYM && PointWindow<NV>::kLdsBytes + kEqTabPad + kEqTabBytes <= 160 * 1024)
             ? kEqTabPad + kEqTabBytes : 0;
}
// ... and for the Z-function spline table of the damping (DevParams::zf_lds; 2001 x 4 doubles = 64 KB in RAYS):
// kernels with the absorbed-power row whose window leaves room for it
constexpr size_t kZfTabBytes = 65536 + 64;
template <int EQ, int NS, int NV>
constexpr size_t zf_tab_lds_bytes() {
  return (RayVec<(EQ & kEqMultiSpec) != 0, NS, NV>::DAMP &&
          PointWindow<NV>::kLdsBytes + eq_tab_lds_bytes<EQ, NV>() + kZfTabBytes <= 160 * 1024) ? kZfTabBytes : 0;
}

}  // namespace rays
