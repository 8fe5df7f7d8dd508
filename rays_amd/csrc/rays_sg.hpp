// rays_sg.hpp -- Shampine-Gordon ray-trace kernel (ode_solver_name = 'SG_ODE').
//
// Reference path restated here:
//   trace_rays        ray_tracing.f90:67-264
//   SG_ode            SG_ode_m.f90:89-159      (iflag = 1 every call, tolerance-inflation retry loop)
//   ode/de/step/intrp ode_RAYS.f90:1-230 / 232-593 / 595-1234 / 1235-1362
//                     (Burkardt's F90 of Shampine & Gordon's variable-order Adams PECE code)
//
// MI355X design.  `step` calls the RHS at three places (start, predictor, corrector).  Run as
// written, lanes that sit at different call sites would serialise three inlined copies of the
// (expensive) RHS.  Here the integrator is a per-lane coroutine: every trip of the wave loop does
// ONE convergent rhs_eval for the lanes of the phase being served, then each of them runs the
// continuation of its own integrator (segments SEG_*, in pipeline order) up to its next RHS
// request.  The integrator state that is indexed by the lane's order k lives in tiers (LDS /
// registers for the low orders, private memory above: see "Per-lane integrator storage" below);
// the ODE vectors y, p, wt are register-resident.
//
// Because SG_ode restarts the integrator on every output interval (fresh work arrays, iflag = 1),
// the first RHS of `step` (start = true) is evaluated at the state check_save just saw, so it is
// fused with check_save exactly as in the RK4 kernel (rhs_eval with do_check).
#pragma once

#include "rays_trace.hpp"

namespace rays {

// Developer build (-DRAYS_SG_PROFILE): wave-clock spent per section of the wave loop, summed over all
// waves into g_sg_prof (read back by rays_hip_debug_sg_profile).  Not compiled into the product.
#ifdef RAYS_SG_PROFILE
__device__ unsigned long long g_sg_prof[32];
#define SG_PROF_DECL unsigned long long prof_t = clock64(), prof_acc[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SG_PROF(slot)                              \
  do {                                             \
    const unsigned long long now_ = clock64();     \
    prof_acc[slot] += now_ - prof_t;               \
    prof_t = now_;                                 \
  } while (0)
#define SG_PROF_FLUSH                                                          \
  if ((threadIdx.x & 63) == 0)                                                 \
    for (int i_ = 0; i_ < 32; i_++) atomicAdd(&g_sg_prof[i_], prof_acc[i_]);
#else
#define SG_PROF_DECL
#define SG_PROF(slot) ((void)0)
#define SG_PROF_FLUSH
#endif

enum : int {
  PC_CHECK = 0,  // rhs at the recorded state: check_save + start-of-interval f
  PC_F1 = 1,     // start f after a crash/restart inside an interval (ode_RAYS.f90:860)
  PC_F2 = 2,     // predictor evaluation (:1017)
  PC_F3 = 3      // corrector evaluation (:1142)
};

// Early exit of a predicated loop INSIDE divergent control flow: true while any of the lanes that are active there
// still has work (the hardware's vote counts the active lanes only).  On the host emulator, where a vote is a rendezvous
// of all 64 lanes, the loop just runs to its end (the predicates keep the arithmetic the same).
#ifdef RAYS_HOST_EMUL
#define RAYS_ANY_ACTIVE(x) true
#else
#define RAYS_ANY_ACTIVE(x) __any(x)
#endif

enum : unsigned { FL_START = 1u, FL_PHASE1 = 2u, FL_NORND = 4u, FL_STIFF = 8u, FL_FIRST = 16u };

enum : int {
  SEG_WAIT = 0,
  SEG_DE_BEGIN,    // ode/de entry for interval [t, tout]
  SEG_DE_TOP,      // top of de's step loop
  SEG_START_DONE,  // have f(x, yy) for start = true
  SEG_COEF,        // coefficients + predictor
  SEG_AFTER_F2,
  SEG_AFTER_F3,
  SEG_CRASH,       // iflag = 3 return to SG_ode
  SEG_STOP         // ray finished with code `stop`
};

// ---------------------------------------------------------------------------------------------
// Per-lane integrator storage.
//
// `step` keeps the coefficient vectors psi, alpha, beta, sig, g, v (<= 13 entries each) and the
// divided differences phi(neqn,16) between calls, indexed by the lane's own order k.  As private
// arrays they are scratch memory, and with one wave per SIMD every scratch access that is waited
// for costs a full memory round trip (~1-2 us; 64k lanes x 2 KB of scratch live in L2/MALL):
// the first version of this kernel spent ~100 us per trip there against ~3 us in the RHS.
//
// SG_ode restarts the integrator on every output interval (SG_ode_m.f90:118-122), so the order
// stays low: k <= 4 in 99.7 % of the step attempts of the Solovev fan with finite-difference dD (cfg 3), k <= 4 in
// 84 % and k <= 6 in 96.7 % on the eqdsk fan with damping (cfg 5; counted with the C restatement).
// Storage is therefore tiered by index:
//   coefficient entries 1..6          LDS, lane-interleaved (element e of lane L at col[e * 64])
//   phi rows 1..2                     registers: every loop over rows is unrolled over q with the lane's
//                                     bounds as predicates and a wave-uniform skip
//   phi rows 3..2+LR (LR = 44/nv)     LDS
//   everything above, the round-off   the launch's global workspace (TraceArgs::sg_far, [slot][lane], so a wave's
//   rows 15-16 and y of SG_ode        access coalesces); reached in < 1 % of the steps on the Solovev fans and in
//                                     ~9 % on the eqdsk fan, except rows 15-16, which are read and written once per step
// No private arrays: a dynamically indexed private array is scratch memory, and with it came 0.9-1.1 KB of
// scratch per lane, 115-180 SGPR spills and an HBM write stream 12x the trajectory's (round 1).  Now the kernels
// have 0-100 B of scratch per lane.  LDS per lane: 36 + LR x nv <= 80 doubles = the CU's 160 KB for its four waves,
// so the SG kernel stages no trajectory points in LDS: it records one point per ~30 trips and the lane writes it
// straight to HBM.
// ---------------------------------------------------------------------------------------------
// (the RAYS_SG_* overrides exist for tests/hip_emul, which shrinks the fast tiers so that ordinary
// rays cross every tier boundary)
#ifndef RAYS_SG_TIER
#define RAYS_SG_TIER 6  // entries 1..k+1 (k+2 in intrp) are touched at order k
// measured on MI355X (cfg5 eqdsk 256k rays nv = 8 | cfg3 Solovev 64k nv = 7, ms per pass), round-off rows in the
// workspace: tier 5 -> 112.6 | 336.2, tier 6 -> 111.0 | 318.6, tier 8 -> 114.3 | 342.8; round-off rows in LDS instead
// (fewer phi rows fit): tier 5 -> 118.3 | 333.5, tier 7 -> 135.6 | 343.9.  What decides is how many phi rows fit.
#endif
// phi rows in registers: 2, and 3 for nv = 8 (the eqdsk + damping fans reach order 6 in 5 % of their step attempts;
// the third register row moves the LDS rows up to 4..8, so that order stays out of the workspace).  Measured with the
// final control flow (cfg 5 nv = 8 | cfg 3 nv = 7, ms, one box): 2 rows 97.1-97.5 | 301-304, 3 rows 95.0 | 308.5,
// 4 rows 95.3-95.5 | 309-320, 6 rows 104-105 | 319.5.
constexpr int kSgTier = RAYS_SG_TIER;  // coefficient entries held in LDS
// LDS pointers carry their address space explicitly: the tiered accessors choose between an LDS
// and a private location, and with generic pointers LLVM may merge the two loads into one load
// through a selected flat pointer (slow, and hipcc 7.2 then fails in instruction selection).
#ifdef RAYS_HOST_EMUL
typedef double* sg_lds_ptr;
#else
typedef __attribute__((address_space(3))) double* sg_lds_ptr;
#endif
template <int NV>
constexpr int sg_phi_lds_rows() {
#ifdef RAYS_SG_LDS_ROWS
  return RAYS_SG_LDS_ROWS;
#else
  // what the CU's LDS leaves: 80 doubles per lane - 6 coefficient arrays
  return (80 - 6 * kSgTier) / NV < 1 ? 1 : (80 - 6 * kSgTier) / NV;
#endif
}
template <int NV>
constexpr int sg_phi_reg_rows() {
#ifdef RAYS_SG_REG_ROWS
  return RAYS_SG_REG_ROWS;
#else
  return NV == 8 ? 3 : 2;
#endif
}

// This lane's column of the launch's upper-tier workspace (TraceArgs::sg_far, element e at column[e * stride]).
// Recomputed where it is needed (rare) from the kernel arguments instead of being carried in registers.
RAYS_DEV double* sg_far_column(const TraceArgs& A_hot) {
  return cold_args(A_hot).sg_far + ((long long)blockIdx.x * blockDim.x + threadIdx.x);
}
RAYS_DEV long long sg_far_stride() { return (long long)gridDim.x * blockDim.x; }

struct SgCoef {
  enum { PSI = 0, ALPHA, BETA, SIG, G, V, kArrays };
  sg_lds_ptr col;               // this lane's LDS column
  // entries above the LDS tier (.. 14): the lane's column of the launch's global workspace (TraceArgs::sg_far),
  // element e at far[e * far_stride].  Not private memory: a dynamically indexed private array is scratch,
  // and its mere presence cost the hot loop address arithmetic and spill slots next to it.
  const TraceArgs* args;
  static constexpr int kFarPerArray = 14 - kSgTier;
  static constexpr int kFarDoubles = kArrays * kFarPerArray;

  struct Ref {  // reads / writes one entry through the tier it lives in
    const SgCoef* s;
    int arr, i;
    RAYS_DEV operator double() const {
      if (RAYS_RARE(i > kSgTier)) return sg_far_column(*s->args)[(arr * kFarPerArray + i - kSgTier - 1) * sg_far_stride()];
      return s->col[(arr * kSgTier + i - 1) * kWave];
    }
    RAYS_DEV const Ref& operator=(double x) const {
      if (RAYS_RARE(i > kSgTier))
        sg_far_column(*s->args)[(arr * kFarPerArray + i - kSgTier - 1) * sg_far_stride()] = x;
      else
        s->col[(arr * kSgTier + i - 1) * kWave] = x;
      return *this;
    }
    RAYS_DEV const Ref& operator=(const Ref& o) const { return *this = (double)o; }  // copies the VALUE
  };
  RAYS_DEV Ref psi(int i) const { return Ref{this, PSI, i}; }      // i = 1..12
  RAYS_DEV Ref alpha(int i) const { return Ref{this, ALPHA, i}; }  // i = 1..12
  RAYS_DEV Ref beta(int i) const { return Ref{this, BETA, i}; }    // i = 1..12
  RAYS_DEV Ref sig(int i) const { return Ref{this, SIG, i}; }      // i = 1..13
  RAYS_DEV Ref g(int i) const { return Ref{this, G, i}; }          // i = 1..13
  RAYS_DEV Ref v(int i) const { return Ref{this, V, i}; }          // i = 1..12
  // intrp runs when the interval is complete and the integrator is about to be restarted, so its
  // work vectors w(1:14), g(1:13) (ode_RAYS.f90:1300-1301) reuse the alpha and sig storage.
  RAYS_DEV Ref wi(int i) const { return Ref{this, ALPHA, i}; }  // i = 1..14
  RAYS_DEV Ref gi(int i) const { return Ref{this, SIG, i}; }    // i = 1..13
};

template <int NV>
struct SgLds {  // LDS doubles per lane / per wave
  static constexpr int kPhiBase = SgCoef::kArrays * kSgTier;
  static constexpr int kPerLane = kPhiBase + sg_phi_lds_rows<NV>() * NV;
  static constexpr int kDoublesPerWave = kPerLane * kWave;
  // gstr(0:13) as a 16-double table behind the four waves' columns, where the 160 KB leave room for it
  static constexpr int kGstrBase = 4 * kDoublesPerWave;
  static constexpr bool kGstrTable = (size_t)(kGstrBase + 16) * sizeof(double) <= 160 * 1024;
  static constexpr size_t kLdsBytes = (size_t)(kGstrBase + (kGstrTable ? 16 : 0)) * sizeof(double);
};

// Divided differences phi(neqn,16) (ode_RAYS.f90:668), tiered as described above.
template <int NV, int R, int LR>
struct SgPhi {
  static_assert(R >= 2 && LR >= 1, "rows 1 and 2 are addressed directly (start of an interval)");
  double lo[R][NV];  // rows 1..R: registers (static indices only)
  sg_lds_ptr mid;    // rows R+1..R+LR: LDS, element (row, l) of this lane at mid[((row-R-1)*NV + l) * 64]
  // phi(:,15) and phi(:,16) carry the propagated round-off of `step` (ode_RAYS.f90:1006-1011, 1131-1136) whenever
  // the tolerance is within 100x of the round-off level -- every step of the BASELINE runs (1e-9).  They live in
  // the workspace behind rows .. 14: one read and one write of each per step, issued well ahead of their use;
  // LDS is worth more as phi rows (measurements at RAYS_SG_TIER above).
  RAYS_DEV double* rnd_col() const { return sg_far_column(*args) + (SgCoef::kFarDoubles + kFarDoubles) * sg_far_stride(); }
  // (take rnd_col() ONCE per block of accesses: it re-reads the workspace pointer from the kernel arguments)
  RAYS_DEV static double& rnd15(double* col, int l) { return col[l * sg_far_stride()]; }
  RAYS_DEV static double& rnd16(double* col, int l) { return col[(NV + l) * sg_far_stride()]; }
  const TraceArgs* args;  // rows R+LR+1..14: global workspace behind the coefficient tiers (see SgCoef)
  static constexpr int kFarRows = 14 - R - LR;
  static constexpr int kFarDoubles = kFarRows * NV;
  RAYS_DEV double& far_at(int row, int l) const {
    return sg_far_column(*args)[(SgCoef::kFarDoubles + (row - R - LR - 1) * NV + l) * sg_far_stride()];
  }

  RAYS_DEV void load_far(int i, double out[NV]) const {  // i > R
    const int r = i - R - 1;
    if (r < LR) {
#pragma unroll
      for (int l = 0; l < NV; l++) out[l] = mid[(r * NV + l) * kWave];
    } else {
#pragma unroll
      for (int l = 0; l < NV; l++) out[l] = far_at(i, l);
    }
  }
  RAYS_DEV void store_far(int i, const double in[NV]) {  // i > R
    const int r = i - R - 1;
    if (r < LR) {
#pragma unroll
      for (int l = 0; l < NV; l++) mid[(r * NV + l) * kWave] = in[l];
    } else {
#pragma unroll
      for (int l = 0; l < NV; l++) far_at(i, l) = in[l];
    }
  }
  RAYS_DEV void get(int i, double out[NV]) const {  // out = phi(:, i), i dynamic
#pragma unroll
    for (int l = 0; l < NV; l++) out[l] = 0.;
#pragma unroll
    for (int q = 1; q <= R; q++) {
      if (!RAYS_ANY_ACTIVE(i == q)) continue;
      if (i == q) {
#pragma unroll
        for (int l = 0; l < NV; l++) out[l] = lo[q - 1][l];
      }
    }
    if (i > R) load_far(i, out);
  }
  RAYS_DEV void set(int i, const double in[NV]) {  // phi(:, i) = in, i dynamic
#pragma unroll
    for (int q = 1; q <= R; q++) {
      if (!RAYS_ANY_ACTIVE(i == q)) continue;
      if (i == q) {
#pragma unroll
        for (int l = 0; l < NV; l++) lo[q - 1][l] = in[l];
      }
    }
    if (i > R) store_far(i, in);
  }
  // phi(:, i) = beta(i) * phi(:, i), i = a..b                       (ode_RAYS.f90:992-996)
  template <class Beta>
  RAYS_DEV void scale(int a, int b, Beta beta) {
#pragma unroll
    for (int q = 1; q <= R; q++) {
      const bool on = q >= a && q <= b;
      if (!RAYS_ANY_ACTIVE(on)) continue;
      if (on) {
        const double bq = beta(q);
#pragma unroll
        for (int l = 0; l < NV; l++) lo[q - 1][l] = bq * lo[q - 1][l];
      }
    }
    for (int i = (a > R + 1 ? a : R + 1); i <= b; i++) {
      const double bq = beta(i);
      double row[NV];
      load_far(i, row);
#pragma unroll
      for (int l = 0; l < NV; l++) row[l] = bq * row[l];
      store_far(i, row);
    }
  }
  // predictor (:1003-1011), i = k..1:  p += phi(:,i)*g(i); phi(:,i) += phi(:,i+1).
  // phi(:,k+1) has just been zeroed; `up` carries the updated row above.
  template <class G>
  RAYS_DEV void predict(int k, G g, double pp[NV]) {
    double up[NV];
#pragma unroll
    for (int l = 0; l < NV; l++) up[l] = 0.;
    for (int i = k; i > R; i--) {
      const double gg = g(i);
      double row[NV];
      load_far(i, row);
#pragma unroll
      for (int l = 0; l < NV; l++) {
        pp[l] = pp[l] + row[l] * gg;
        row[l] = row[l] + up[l];
        up[l] = row[l];
      }
      store_far(i, row);
    }
#pragma unroll
    for (int q = R; q >= 1; q--) {
      const bool on = q <= k;
      if (!RAYS_ANY_ACTIVE(on)) continue;
      if (on) {
        const double gg = g(q);
#pragma unroll
        for (int l = 0; l < NV; l++) {
          double r = lo[q - 1][l];
          pp[l] = pp[l] + r * gg;
          r = r + up[l];
          lo[q - 1][l] = r;
          up[l] = r;
        }
      }
    }
  }
  // failed step (:1090-1094), i = 1..k ascending:  phi(:,i) = (phi(:,i) - phi(:,i+1)) / beta(i)
  template <class Beta>
  RAYS_DEV void restore(int k, Beta beta) {
    double nxt[NV];  // phi(:, R+1), not yet modified when row R is restored
#pragma unroll
    for (int l = 0; l < NV; l++) nxt[l] = 0.;
    if (k >= R) load_far(R + 1, nxt);
#pragma unroll
    for (int q = 1; q <= R; q++) {
      const bool on = q <= k;
      if (!RAYS_ANY_ACTIVE(on)) continue;
      if (on) {
        const Recip b = make_recip(beta(q));
#pragma unroll
        for (int l = 0; l < NV; l++) {
          const double above = q < R ? lo[q < R ? q : 0][l] : nxt[l];  // phi(:, q+1), old value
          lo[q - 1][l] = div(lo[q - 1][l] - above, b);
        }
      }
    }
    for (int i = R + 1; i <= k; i++) {
      const Recip b = make_recip(beta(i));
      double row[NV], abv[NV];
      load_far(i, row);
      load_far(i + 1, abv);
#pragma unroll
      for (int l = 0; l < NV; l++) row[l] = div(row[l] - abv[l], b);
      store_far(i, row);
    }
  }
  // phi(:, i) += d, i = 1..k                                         (:1160-1164)
  RAYS_DEV void add(int k, const double d[NV]) {
#pragma unroll
    for (int q = 1; q <= R; q++) {
      const bool on = q <= k;
      if (!RAYS_ANY_ACTIVE(on)) continue;
      if (on) {
#pragma unroll
        for (int l = 0; l < NV; l++) lo[q - 1][l] = lo[q - 1][l] + d[l];
      }
    }
    for (int i = R + 1; i <= k; i++) {
      double row[NV];
      load_far(i, row);
#pragma unroll
      for (int l = 0; l < NV; l++) row[l] = row[l] + d[l];
      store_far(i, row);
    }
  }
  // intrp (:1343-1349), i = ki..1:  yout += g(i) * phi(:, i)
  template <class G>
  RAYS_DEV void interp(int ki, G g, double yout[NV]) const {
    for (int i = ki; i > R; i--) {
      const double gg = g(i);
      double row[NV];
      load_far(i, row);
#pragma unroll
      for (int l = 0; l < NV; l++) yout[l] = yout[l] + gg * row[l];
    }
#pragma unroll
    for (int q = R; q >= 1; q--) {
      const bool on = q <= ki;
      if (!RAYS_ANY_ACTIVE(on)) continue;
      if (on) {
        const double gg = g(q);
#pragma unroll
        for (int l = 0; l < NV; l++) yout[l] = yout[l] + gg * lo[q - 1][l];
      }
    }
  }
};

// doubles per lane of the launch's global workspace (TraceArgs::sg_far): the tiers above LDS
template <int NV>
constexpr int sg_far_doubles_per_lane() {
  return SgCoef::kFarDoubles + SgPhi<NV, sg_phi_reg_rows<NV>(), sg_phi_lds_rows<NV>()>::kFarDoubles + 2 * NV + NV;  // + round-off rows 15, 16 + y (below)
}
// y of SG_ode -- the ray state the reference's `ode` was last entered with or returned (the last completed output
// point, or the state after a tolerance-inflation restart).  It is only READ when a ray stops (end_ray_vec), so
// it lives in the workspace behind the upper tiers: written once per output interval, no registers.
template <int NV>
RAYS_DEV void sg_save_y(const TraceArgs& A_hot, const double y[NV]) {
  double* c = sg_far_column(A_hot) + (sg_far_doubles_per_lane<NV>() - NV) * sg_far_stride();
#pragma unroll
  for (int l = 0; l < NV; l++) c[l * sg_far_stride()] = y[l];
}
template <int NV>
RAYS_DEV void sg_load_y(const TraceArgs& A_hot, double y[NV]) {
  const double* c = sg_far_column(A_hot) + (sg_far_doubles_per_lane<NV>() - NV) * sg_far_stride();
#pragma unroll
  for (int l = 0; l < NV; l++) y[l] = c[l * sg_far_stride()];
}

// gstr(1:13) -- single-precision literals widened to double (ode_RAYS.f90:776-779).  Indexed by the
// lane's order: a select chain over compile-time constants (a table in global memory cost a waited
// ~0.7 us load per use, four uses per step).
RAYS_DEV double gstr(int i) {
  constexpr double t[14] = {
      0., (double)0.50e+00f, (double)0.0833e+00f, (double)0.0417e+00f, (double)0.0264e+00f,
      (double)0.0188e+00f, (double)0.0143e+00f, (double)0.0114e+00f, (double)0.00936e+00f,
      (double)0.00789e+00f, (double)0.00679e+00f, (double)0.00592e+00f, (double)0.00524e+00f,
      (double)0.00468e+00f};
  double r = 0.;
#pragma unroll
  for (int q = 1; q <= 13; q++) r = i == q ? t[q] : r;
  return r;
}

#ifndef RAYS_SG_PATIENCE
#define RAYS_SG_PATIENCE 128
#endif
constexpr int kSgPatienceMax = RAYS_SG_PATIENCE, kSgPatienceProbe = 16, kSgProbeTrips = 512;

template <int EQ, int NS, int DERIV, int NV>
__global__ void __launch_bounds__(256)
sg_trace_kernel(const DevParams P_kernarg, const TraceArgs A_hot) {
  DevParams P;  // working copy with the hot per-species constants in vector registers (rays_device.hpp)
  hot_params<EQ, NS>(P_kernarg, P);
  extern __shared__ double lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  constexpr int RR = sg_phi_reg_rows<NV>(), LR = sg_phi_lds_rows<NV>();
  const sg_lds_ptr lane_lds = (sg_lds_ptr)(lds + wave * SgLds<NV>::kDoublesPerWave + lane);
#ifndef RAYS_HOST_EMUL
  // gstr(1:13) is indexed by the lane's order: a lookup in a small LDS table (where the block's LDS has 128 bytes to
  // spare) instead of the select chain over thirteen constants (~40 instructions a call, four calls per step)
  if constexpr (SgLds<NV>::kGstrTable) {
    if (threadIdx.x < 14) lds[SgLds<NV>::kGstrBase + threadIdx.x] = gstr((int)threadIdx.x);
    __syncthreads();
  }
  const sg_lds_ptr gstr_tab = (sg_lds_ptr)(lds + SgLds<NV>::kGstrBase);
#define RAYS_GSTR(i) (SgLds<NV>::kGstrTable ? (double)gstr_tab[(i)] : gstr(i))
#else
#define RAYS_GSTR(i) gstr(i)
#endif
  SgCoef S;
  S.col = lane_lds;
  SgPhi<NV, RR, LR> F;  // divided differences phi(neqn,16)
  F.mid = lane_lds + SgLds<NV>::kPhiBase * kWave;
  S.args = &A_hot;  // the rarely touched upper tiers live in the launch's workspace (TraceArgs::sg_far)
  F.args = &A_hot;

  const unsigned total_lanes = gridDim.x * blockDim.x;
  const long long npt = (long long)P.nstep_max + 1;
  constexpr double kEps = 2.220446049250313e-16;  // epsilon(1._rkind)
  constexpr double twou = 2.0 * kEps, fouru = 2.0 * twou;
  constexpr int maxnum = 500;  // ode_RAYS.f90:395

  // ---- per-lane ray state -------------------------------------------------------------------
  int ray = blockIdx.x * blockDim.x + threadIdx.x;
  bool alive = ray < A_hot.nray;
  bool need_init = alive;
  int pc = PC_CHECK;
  int nstep = 0;
  double sout = 0.;
  double ds_ray = P.ds;  // output step of this lane's run (a fused scan gives every run its own)
  // (y of SG_ode, the state at the last completed output point, has no registers: between the interpolation
  // that produces it and the `de` entry that consumes it, it IS yy; a copy for the day the ray stops is kept in
  // the workspace, sg_save_y)
  double last_resid = 0., prev_resid = 0., maxr = -1.7976931348623157e308;

  // ---- per-lane integrator state (de / step locals that live across RHS evaluations) ----------
  double yy[NV], pp[NV];  // (yp of `step` is always the f of the current trip: not kept)
  // wt(l) with its reciprocal: ~6 nv quotients x/wt(l) per step share it (rays_device.hpp: Recip)
  Recip wt[NV];
  double t = 0., tout = 0., x = 0., h = 0., hold = 0., eps = 0.;
  double rel_err = 0., abs_err = 0., releps = 0., abseps = 0., absdel = 0., tend = 0.;
  double p5eps = 0., round_ = 0., xold = 0., absh = 0., erk = 0., erkm1 = 0.;
  int k = 1, kold = 0, ns = 0, knew = 1, ifail = 0, nostep = 0, kle4 = 0;
  int resume = SEG_WAIT;  // continuation segment deferred to the next trip (the rare DE_TOP -> CRASH edge)
  int waited = 0;         // trips this lane has waited at PC_CHECK for the rest of its wave (interval alignment)
  int patience = kSgPatienceMax, probe = 0;  // wave-uniform: see "Interval alignment" below
  // Per-lane logicals are kept as bits of ONE integer VGPR rather than as `bool`s: a bool that is
  // live across the divergent continuation loop is a 64-bit lane mask in SGPRs, and with ~10 of
  // them the register allocator spills lane masks inside divergent control flow.
  unsigned fl = FL_START | FL_PHASE1 | FL_NORND | FL_FIRST;
#pragma unroll
  for (int i = 0; i < NV; i++) yy[i] = pp[i] = wt[i].d = wt[i].rc = 0.;

  SG_PROF_DECL
  while (__any(alive)) {
    SG_PROF(0);  // loop overhead, refill
    if (RAYS_RARE(need_init)) {  // ray_tracing.f90:77-93, SG_ode_m.f90:73-85
      const TraceArgs& A = cold_args(A_hot);  // rays_trace.hpp
      start_ray<EQ, NS, NV>(P, A, ray, yy, sout, ds_ray);
      sg_save_y<NV>(A_hot, yy);
      pc = PC_CHECK;
      resume = SEG_WAIT;
      fl |= FL_FIRST;
      nstep = 0;
      t = sout;
      last_resid = 0.;
      prev_resid = 0.;
      maxr = -1.7976931348623157e308;
      rel_err = P.rel_err0;
      abs_err = P.abs_err0;
      need_init = false;
    }

    // ---- the one RHS evaluation of this trip -------------------------------------------------
    double f[NV], resid = 0.;
    int code = 0, cs_flag = 0;
    bool cs_stop = false;
    // Phase alignment.  A lane alternates between the predictor evaluation (PC_F2, continued by
    // the short AFTER_F2 segment) and the corrector / start evaluations (continued by AFTER_F3 ->
    // DE_TOP -> COEF, the long part).  With lanes in both phases the wave would run ALL the
    // continuation code on every trip.  Each trip therefore serves only the phase most lanes are
    // in; the others sit the trip out (their RHS input is unchanged, so nothing is lost but the
    // slot).  Served lanes move to the other phase, so the wave falls into step after one trip and
    // a lane idles about once per output interval (an interval has an odd number of evaluations).
    const bool in_f2 = pc == PC_F2;
    const bool wants_rhs = alive && resume == SEG_WAIT;  // a lane with a deferred segment needs no evaluation
    // Interval alignment.  A lane that has completed its output interval (PC_CHECK) waits until no lane of the
    // wave is inside an interval any more; then ONE trip serves all of them, so check_save's extra work in the RHS
    // and the once-per-interval segments (CHECK, DE_BEGIN, intrp, START_DONE) run once per interval for the wave
    // instead of on nearly every trip for one or two lanes.  Where a lane's interval needs many more evaluations
    // than its neighbours' (step-size thrashing at noise-level tolerances: the Solovev fan with finite-difference
    // dD) waiting would cost more than it saves, so the wave has a PATIENCE: the longest wait it accepts, halved
    // each time it runs out, raised again by alignments that complete; at zero the wave uses the plain two-phase
    // vote (CHECK lanes ride with the corrector phase) and probes again after a while.  Measured (cfg 5 | cfg 3, ms):
    // no alignment 109.7 | 324; fixed patience 4: 114.9 | 348, 32: 103.2 | 306, 64: 95.6 | 342, 128: 93.8 | 433,
    // unbounded: 95.5 | 1199; adaptive with maximum 32: 101.1 | 302.7, 64: 96.0 | 304.1, 128: 93.9 | 296.6.
    const bool at_check = pc == PC_CHECK;
    bool act;
    if (patience > 0) {
      const int n_f2 = __popcll(__ballot(wants_rhs && in_f2));
      const int n_f3 = __popcll(__ballot(wants_rhs && !in_f2 && !at_check));
      const bool all_arrived = n_f2 + n_f3 == 0;
      const bool timed_out = !all_arrived && __any(wants_rhs && at_check && waited >= patience);
      const bool serve_check = all_arrived || timed_out;
      const bool serve_f2 = n_f2 > n_f3;
      act = wants_rhs && (serve_check ? at_check : (!at_check && in_f2 == serve_f2));
      waited = (wants_rhs && at_check && !serve_check) ? waited + 1 : 0;
      if (timed_out) patience = patience / 2;                             // wave-uniform: every lane computes the same
      else if (all_arrived) patience = patience + 8 < kSgPatienceMax ? patience + 8 : kSgPatienceMax;
      probe = 0;
    } else {
      const int n_f2 = __popcll(__ballot(wants_rhs && in_f2)), n_other = __popcll(__ballot(wants_rhs && !in_f2));
      const bool serve_f2 = n_f2 > n_other;
      act = wants_rhs && (in_f2 == serve_f2);
      waited = 0;
      probe = probe + 1;
      if (probe >= kSgProbeTrips) patience = kSgPatienceProbe;
    }
    // RHS input: the recorded state (PC_CHECK), the predicted p (PC_F2), else the current yy
    double win[NV];
#pragma unroll
    for (int l = 0; l < NV; l++) win[l] = pc == PC_F2 ? pp[l] : yy[l];
    SG_PROF(1);  // init + phase vote
#ifdef RAYS_SG_PROFILE
    prof_acc[25] += __popcll(__ballot(act));              // lanes served by this trip's RHS
    prof_acc[26] += 1;                                    // trips
    prof_acc[27] += __popcll(__ballot(alive));            // lanes that hold a ray
    prof_acc[28] += __popcll(__ballot(alive && at_check && !act));  // lanes waiting for their wave at an interval end
#endif
    if (act) rhs_eval<EQ, NS, DERIV, NV>(P, win, pc == PC_CHECK, resid, cs_flag, cs_stop, code, f);
#ifdef RAYS_SG_PROFILE
    SG_PROF(__popcll(__ballot(act)) <= 8 ? 15 : 2);  // RHS (slot 15: trips serving <= 8 lanes)
#else
    SG_PROF(2);  // RHS
#endif

    // ---- per-lane continuation -------------------------------------------------------------------
    int stop = 0;
    int done = 0;
    // Control flow is kept FLAT: the segments below are sequential `if (seg == ...)` blocks on the top level of
    // the wave loop, in pipeline order, and run once per trip.  (An enclosing `if (act)` and a `while (seg !=
    // SEG_WAIT)` around them made every join a merge point of the whole integrator state: ~160 register moves
    // per join and trip.)  The one backward edge, DE_TOP -> CRASH (the step size underflowed or the tolerance is
    // below the round-off level: rare), is deferred to the next trip through `resume`.
    int seg = resume;
    resume = SEG_WAIT;
    int have_f = 0;  // f(x, yy) for start = true is already in f[]
    if (act) {
      if (pc == PC_CHECK) {
        seg = SEG_DE_BEGIN;
        if (RAYS_RARE(fl & FL_FIRST)) {  // ray_tracing.f90:92-112
          record_point<NV>(cold_args(A_hot), (long long)ray * npt, yy, 0.);
          fl &= ~FL_FIRST;
          if (RAYS_RARE(cs_stop)) {
            const TraceArgs& A = cold_args(A_hot);
            A.npoints[ray] = 1;
            A.stop_code[ray] = cs_flag;
            if (A.end_ray_vec)
#pragma unroll
              for (int i = 0; i < NV; i++) A.end_ray_vec[(long long)ray * NV + i] = 0.;
            if (A.end_residuals) A.end_residuals[ray] = 0.;
            if (A.max_residuals) A.max_residuals[ray] = 0.;
            done = 1;
            stop = -1;
            seg = SEG_WAIT;
          }
        } else {
          // the interval completed: yy is y(tout)  (ray_tracing.f90:212-243)
          if (RAYS_RARE(cs_stop)) {
            stop = cs_flag;
            seg = SEG_STOP;
          } else {
            nstep = nstep + 1;
            record_point<NV>(cold_args(A_hot), (long long)ray * npt + nstep, yy, resid);
            if (fabs(last_resid) > maxr) maxr = fabs(last_resid);
            prev_resid = last_resid;
            last_resid = resid;
          }
        }
        if (seg == SEG_DE_BEGIN) {  // ray_tracing.f90:118-172
          t = sout;  // s = sout
          sout = sout + ds_ray;
          tout = sout;
          if (sout > P.s_max) {
            stop = RAYS_STOP_SOUT_GT_SMAX;
            seg = SEG_STOP;
          } else if (nstep + 1 > P.nstep_max) {
            stop = RAYS_STOP_NSTEP_MAX;
            seg = SEG_STOP;
          }
          have_f = 1;
        }
      } else if (pc == PC_F1) {
        have_f = 1;
        seg = SEG_START_DONE;
      } else if (pc == PC_F2) {
        seg = SEG_AFTER_F2;
      } else {
        seg = SEG_AFTER_F3;
      }
    }

    // Continuation segments in pipeline order: a lane falls through AFTER_F3 -> DE_TOP -> COEF (or
    // CHECK -> DE_BEGIN -> DE_TOP -> START_DONE -> COEF) in ONE pass.
    SG_PROF(3);  // CHECK bookkeeping
    {
      {
        if (seg == SEG_AFTER_F2) {
          if (RAYS_RARE(code)) {  // :1020
            stop = code;
            seg = SEG_STOP;
          } else {
            // ---- error estimates (ode_RAYS.f90:1026-1070) ----
            const int kp1 = k + 1, km1 = k - 1, km2 = k - 2;
            double erkm2 = 0.0;
            erkm1 = 0.0;
            erk = 0.0;
            double rk[NV], rkm1[NV];  // phi(:,k), phi(:,k-1)
            F.get(k, rk);
            F.get(km1, rkm1);
            SG_PROF(16);
#pragma unroll
            for (int l = 0; l < NV; l++) {
              const double ph1 = F.lo[0][l];
              if (0 < km2) {
                const double q = div(rkm1[l] + f[l] - ph1, wt[l]);
                erkm2 = erkm2 + q * q;
              }
              if (0 <= km2) {
                const double q = div(rk[l] + f[l] - ph1, wt[l]);
                erkm1 = erkm1 + q * q;
              }
              const double q = div(f[l] - ph1, wt[l]);
              erk = erk + q * q;
            }
            SG_PROF(17);
            if (0 < km2) erkm2 = absh * S.sig(km1) * RAYS_GSTR(km2) * sqrt(erkm2);
            if (0 <= km2) erkm1 = absh * S.sig(k) * RAYS_GSTR(km1) * sqrt(erkm1);
            const double err = absh * sqrt(erk) * (S.g(k) - S.g(kp1));
            erk = absh * sqrt(erk) * S.sig(kp1) * RAYS_GSTR(k);
            knew = k;
            if (0 < km2) {
              if (fmax(erkm1, erkm2) <= erk) knew = km1;
            } else if (0 == km2) {
              if (erkm1 <= 0.5 * erk) knew = km1;
            }
            SG_PROF(18);
            if (err <= eps) {
              // ---- successful: correct (ode_RAYS.f90:1128-1142) ----
              kold = k;
              hold = h;
              const double hg = h * S.g(kp1);
              if (!(fl & FL_NORND)) {
                double* const rnd = F.rnd_col();
#pragma unroll
                for (int l = 0; l < NV; l++) {
                  const double rho = hg * (f[l] - F.lo[0][l]) - F.rnd16(rnd, l);
                  yy[l] = pp[l] + rho;
                  F.rnd15(rnd, l) = (yy[l] - pp[l]) - rho;
                }
              } else {
#pragma unroll
                for (int l = 0; l < NV; l++) yy[l] = pp[l] + hg * (f[l] - F.lo[0][l]);
              }
              pc = PC_F3;
              seg = SEG_WAIT;
              SG_PROF(19);
            } else {
              // ---- failed step: restore, shrink (ode_RAYS.f90:1086-1120) ----
              fl &= ~FL_PHASE1;
              x = xold;
              F.restore(k, [&](int i) { return S.beta(i); });
              for (int i = 2; i <= k; i++) S.psi(i - 1) = S.psi(i) - h;
              ifail = ifail + 1;
              double temp2 = 0.5;
              if (3 < ifail) {
                if (p5eps < 0.25 * erk) temp2 = sqrt(p5eps / erk);
              }
              if (3 <= ifail) knew = 1;
              h = temp2 * h;
              k = knew;
              if (fabs(h) < fouru * fabs(x)) {
                h = copysign(fouru * fabs(x), h);
                eps = eps + eps;
                seg = SEG_CRASH;
              } else {
                seg = SEG_COEF;
              }
            }
          }
        }
        SG_PROF(4);
        if (seg == SEG_AFTER_F3) {
          if (RAYS_RARE(code)) {  // :1145
            stop = code;
            seg = SEG_STOP;
          } else {
            // ---- update differences, choose order and step (ode_RAYS.f90:1151-1231) ----
            const int kp1 = k + 1, kp2 = k + 2, km1 = k - 1;
            double d1[NV], d2[NV];  // new phi(:,kp1), phi(:,kp2)
            F.get(kp2, d2);
#pragma unroll
            for (int l = 0; l < NV; l++) {
              d1[l] = f[l] - F.lo[0][l];
              d2[l] = d1[l] - d2[l];
            }
            F.set(kp1, d1);
            F.set(kp2, d2);
            F.add(k, d1);
            SG_PROF(20);
            double erkp1 = 0.0;
            if (knew == km1 || k == 12) fl &= ~FL_PHASE1;
            if (fl & FL_PHASE1) {
              k = kp1;
              erk = erkp1;
            } else if (knew == km1) {
              k = km1;
              erk = erkm1;
            } else if (kp1 <= ns) {
#pragma unroll
              for (int l = 0; l < NV; l++) {
                const double q = div(d2[l], wt[l]);
                erkp1 = erkp1 + q * q;
              }
              erkp1 = absh * RAYS_GSTR(kp1) * sqrt(erkp1);
              if (k == 1) {
                if (erkp1 < 0.5 * erk) {
                  k = kp1;
                  erk = erkp1;
                }
              } else if (erkm1 <= fmin(erk, erkp1)) {
                k = km1;
                erk = erkm1;
              } else if (erkp1 < erk && k < 12) {
                k = kp1;
                erk = erkp1;
              }
            }
            SG_PROF(21);
            double hnew = h + h;
            if (!(fl & FL_PHASE1)) {
              const double two_k1 = (double)(2 << k);  // two(k+1) = 2**(k+1)
              if (p5eps < erk * two_k1) {
                hnew = h;
                if (p5eps < erk) {
                  const double temp2 = (double)(k + 1);
                  const double r = libm::pow(p5eps / erk, 1.0 / temp2);  // rays_libm.hpp: glibc's pow
                  hnew = absh * fmax(0.5, fmin((double)0.9f, r));
                  hnew = copysign(fmax(hnew, fouru * fabs(x)), h);
                }
              }
            }
            h = hnew;
            // ---- back in de (ode_RAYS.f90:579-588) ----
            nostep = nostep + 1;
            kle4 = kle4 + 1;
            if (4 < kold) kle4 = 0;
            if (50 <= kle4) fl |= FL_STIFF;
            seg = SEG_DE_TOP;
          }
        }
        SG_PROF(5);
        if (RAYS_RARE(seg == SEG_CRASH)) {  // de returns iflag = 3 (ode_RAYS.f90:566-575), SG_ode_m.f90:139-149
          rel_err = eps * releps;
          abs_err = eps * abseps;
          sg_save_y<NV>(A_hot, yy);  // y = yy
          t = x;
          const double total_error = fabs(rel_err) + fabs(abs_err);
          if (total_error > P.sg_error_limit) {
            stop = RAYS_STOP_ODE_TOTAL_ERROR;
            seg = SEG_STOP;
          } else {
            have_f = 0;
            seg = SEG_DE_BEGIN;
          }
        }
        SG_PROF(6);
        if (seg == SEG_DE_BEGIN) {
          // ---- de parameter tests + restart (ode_RAYS.f90:423-505); y == yy, t, tout set ----
          if (t == tout) {
            stop = RAYS_STOP_SG_T_EQ_TOUT;
            seg = SEG_STOP;
          } else if (rel_err < 0.0 || abs_err < 0.0) {
            stop = RAYS_STOP_SG_NEG_ERR;
            seg = SEG_STOP;
          } else {
            eps = fmax(rel_err, abs_err);
            if (eps <= 0.0) {
              stop = RAYS_STOP_SG_EPS_LE_0;
              seg = SEG_STOP;
            } else {
              const double del = tout - t;
              absdel = fabs(del);
              tend = t + 10.0 * del;  // :485
              nostep = 0;
              kle4 = 0;
              fl &= ~FL_STIFF;
              releps = rel_err / eps;
              abseps = abs_err / eps;
              fl |= FL_START;  // :497-505
              x = t;
              // (yy = y: they are the same registers)
              h = copysign(fmax(fabs(tout - x), fouru * fabs(x)), tout - x);
              seg = SEG_DE_TOP;
            }
          }
        }
        SG_PROF(7);
        if (seg == SEG_DE_TOP) {
          if (absdel <= fabs(x - t)) {
            // ---- intrp (ode_RAYS.f90:1235-1362) -> y(tout); interval done (:511-518) ----
            // (the rho recurrence :1331 only feeds ypout, which SG_ode discards)
            const double hi = tout - x;
            const int ki = kold + 1;
            for (int i = 1; i <= ki; i++) S.wi(i) = 1.0 / (double)i;
            S.gi(1) = 1.0;
            double term = 0.0;
            for (int j = 2; j <= ki; j++) {
              const double psijm1 = S.psi(j - 1);
              const Recip rpsi = make_recip(psijm1);
              const double gamma = div(hi + term, rpsi);
              const double eta = div(hi, rpsi);
              for (int i = 1; i <= ki + 1 - j; i++) S.wi(i) = gamma * S.wi(i) - eta * S.wi(i + 1);
              S.gi(j) = S.wi(1);
              term = psijm1;
            }
            double yout[NV];
#pragma unroll
            for (int l = 0; l < NV; l++) yout[l] = 0.0;
            F.interp(ki, [&](int i) { return S.gi(i); }, yout);
#pragma unroll
            for (int l = 0; l < NV; l++) yy[l] = yy[l] + hi * yout[l];  // y = yout
            sg_save_y<NV>(A_hot, yy);
            t = tout;
            pc = PC_CHECK;
            seg = SEG_WAIT;
            SG_PROF(22);
          } else if (maxnum <= nostep) {  // :536-548
            stop = (fl & FL_STIFF) ? RAYS_STOP_SG_STIFF : RAYS_STOP_SG_MAXNUM;
            sg_save_y<NV>(A_hot, yy);  // y = yy; t = x
            t = x;
            seg = SEG_STOP;
          } else {
            h = copysign(fmin(fabs(h), fabs(tend - x)), h);  // :552-553
#pragma unroll
            for (int l = 0; l < NV; l++) wt[l] = make_recip(releps * fabs(yy[l]) + abseps);
            // ---- step entry (ode_RAYS.f90:833-885) ----
            if (fabs(h) < fouru * fabs(x)) {
              h = copysign(fouru * fabs(x), h);
              seg = SEG_CRASH;
            } else {
              p5eps = 0.5 * eps;
              double sm = 0.;
#pragma unroll
              for (int l = 0; l < NV; l++) {
                const double q = div(yy[l], wt[l]);
                sm += q * q;
              }
              round_ = twou * sqrt(sm);  // :844
              SG_PROF(23);
              if (p5eps < round_) {
                eps = 2.0 * round_ * (1.0 + fouru);
                seg = SEG_CRASH;
              } else {
                S.g(1) = 1.0;
                S.g(2) = 0.5;
                S.sig(1) = 1.0;
                if (fl & FL_START) {
                  if (have_f) {
                    seg = SEG_START_DONE;
                  } else {  // f(x, yy) needed (:860)
                    pc = PC_F1;
                    seg = SEG_WAIT;
                  }
                } else {
                  ifail = 0;
                  seg = SEG_COEF;
                }
              }
            }
          }
        }
        SG_PROF(8);
        if (seg == SEG_START_DONE) {
          have_f = 0;
          if (RAYS_RARE(code)) {  // :863 stop inside f: y, t untouched
            stop = code;
            seg = SEG_STOP;
          } else {  // :865-885
            double sm = 0.;
#pragma unroll
            for (int l = 0; l < NV; l++) {
              F.lo[0][l] = f[l];
              F.lo[1][l] = 0.0;
              const double q = div(f[l], wt[l]);
              sm += q * q;
            }
            const double total = sqrt(sm);
            absh = fabs(h);
            if (eps < 16.0 * total * h * h) absh = 0.25 * sqrt(eps / total);
            h = copysign(fmax(absh, fouru * fabs(x)), h);
            hold = 0.0;
            k = 1;
            kold = 0;
            fl &= ~FL_START;
            fl |= FL_PHASE1;
            fl |= FL_NORND;
            if (p5eps <= 100.0 * round_) {
              fl &= ~FL_NORND;
              double* const rnd = F.rnd_col();
#pragma unroll
              for (int l = 0; l < NV; l++) F.rnd15(rnd, l) = 0.0;
            }
            ifail = 0;
            seg = SEG_COEF;
          }
        }
        SG_PROF(9);
        if (seg == SEG_COEF) {
          // ---- coefficients + predictor (ode_RAYS.f90:892-1015) ----
          const int kp1 = k + 1, kp2 = k + 2;
          if (h != hold) ns = 0;
          if (ns <= kold) ns = ns + 1;
          const int nsp1 = ns + 1;
          if (ns <= k) {
            // Order of this block: the reference runs (1) the recurrence psi -> beta -> alpha -> sig for i = ns+1..k, (2) the
            // v / w block, (3) g(i) for i = ns+2..k+1 from alpha(i-1) and w.  (2) reads nothing (1) writes (alpha(2..ns-1),
            // alpha(ns) = 1/ns, v), and pass i of (3) needs exactly the alpha pass i-1 of (1) has produced, over the same
            // range: here (2) comes first and (1), (3) are ONE loop.  beta(i-1), psi(i-1), sig(i), alpha(i) of the
            // recurrences are the values just computed, carried in registers -- a pass waits for one read of its
            // storage (the old psi(i-1)) instead of five dependent ones.  Every operation and its order are unchanged.
            const double alpha_ns = 1.0 / (double)ns;
            S.beta(ns) = 1.0;
            S.alpha(ns) = alpha_ns;
            double temp1 = h * (double)ns;
            S.sig(nsp1) = 1.0;
            // w(1:12) is a work vector of this block only (ode_RAYS.f90:946-986): kept in registers,
            // loops unrolled over the maximum order with per-lane bounds as predicates and a
            // wave-uniform early exit.
            double w[14];
#pragma unroll
            for (int iq = 0; iq < 14; iq++) w[iq] = 0.;
            if (ns <= 1) {
#pragma unroll
              for (int iq = 1; iq <= 12; iq++) {
                if (!RAYS_ANY_ACTIVE(iq <= k)) break;
                if (iq <= k) {
                  const double c = 1.0 / (double)(iq * (iq + 1));
                  S.v(iq) = c;
                  w[iq] = c;
                }
              }
            } else {
              if (kold < k) {
                S.v(k) = 1.0 / (double)(k * kp1);
                for (int j = 1; j <= ns - 2; j++) {
                  const int i = k - j;
                  S.v(i) = S.v(i) - S.alpha(j + 1) * S.v(i + 1);
                }
              }
              const int lim = kp1 - ns;
              double v_cur = S.v(1);
#pragma unroll
              for (int iq = 1; iq <= 12; iq++) {  // ascending: v(iq+1) is still the old value
                if (!RAYS_ANY_ACTIVE(iq <= lim)) break;
                if (iq <= lim) {
                  const double v_nxt = S.v(iq + 1);
                  const double c = v_cur - alpha_ns * v_nxt;
                  S.v(iq) = c;
                  w[iq] = c;
                  v_cur = v_nxt;
                }
              }
              S.g(nsp1) = w[1];
            }
            double beta_c = 1.0, sig_c = 1.0;
            for (int i = nsp1; i <= k; i++) {
              const double temp2 = S.psi(i - 1);
              S.psi(i - 1) = temp1;
              beta_c = beta_c * temp1 / temp2;  // beta(i) = beta(i-1)*psi(i-1)/temp2
              S.beta(i) = beta_c;
              temp1 = temp2 + h;
              const double alpha_i = h / temp1;
              S.alpha(i) = alpha_i;
              sig_c = (double)i * alpha_i * sig_c;  // sig(i+1) = i*alpha(i)*sig(i)
              S.sig(i + 1) = sig_c;
              const int lim = kp1 - i;  // g(i+1): w(iq) = w(iq) - alpha(i)*w(iq+1), iq = 1..kp2-(i+1)
#pragma unroll
              for (int iq = 1; iq <= 12; iq++) {
                if (!RAYS_ANY_ACTIVE(iq <= lim)) break;
                if (iq <= lim) w[iq] = w[iq] - alpha_i * w[iq + 1];
              }
              S.g(i + 1) = w[1];
            }
            S.psi(k) = temp1;
          }
          SG_PROF(12);  // coefficient block
          F.scale(nsp1, k, [&](int i) { return S.beta(i); });
          {
            double row[NV];
            F.get(kp1, row);
            F.set(kp2, row);  // phi(:,kp2) = phi(:,kp1)
#pragma unroll
            for (int l = 0; l < NV; l++) {
              row[l] = 0.0;
              pp[l] = 0.0;
            }
            F.set(kp1, row);  // phi(:,kp1) = 0
          }
          SG_PROF(13);  // scale + shift
          F.predict(k, [&](int i) { return S.g(i); }, pp);
          SG_PROF(14);  // predictor
          if (!(fl & FL_NORND)) {
            double* const rnd = F.rnd_col();
#pragma unroll
            for (int l = 0; l < NV; l++) {
              const double tau = h * pp[l] - F.rnd15(rnd, l);
              pp[l] = yy[l] + tau;
              F.rnd16(rnd, l) = (pp[l] - yy[l]) - tau;
            }
          } else {
#pragma unroll
            for (int l = 0; l < NV; l++) pp[l] = yy[l] + h * pp[l];
          }
          SG_PROF(24);
          xold = x;
          x = x + h;
          absh = fabs(h);
          pc = PC_F2;
          seg = SEG_WAIT;
        }
        SG_PROF(10);
        if (RAYS_RARE(seg == SEG_STOP)) {
          done = 1;
          seg = SEG_WAIT;
        }
        if (RAYS_RARE(seg != SEG_WAIT)) resume = seg;  // SEG_CRASH entered from DE_TOP: next trip
      }

      SG_PROF(11);
      if (RAYS_RARE(done && stop >= 0)) {  // ray_tracing.f90:252-260
        const TraceArgs& A = cold_args(A_hot);
        A.npoints[ray] = nstep + 1;
        A.stop_code[ray] = stop;
        if (A.end_ray_vec) {
          double yend[NV];
          sg_load_y<NV>(A_hot, yend);
#pragma unroll
          for (int i = 0; i < NV; i++) A.end_ray_vec[(long long)ray * NV + i] = yend[i];
        }
        if (A.end_residuals) A.end_residuals[ray] = nstep >= 1 ? prev_resid : 0.;
        if (A.max_residuals) A.max_residuals[ray] = maxr;
      }
    }

    // ---- refill finished lanes ---------------------------------------------------------------
    if (RAYS_RARE(done)) {
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
  SG_PROF_FLUSH
#undef RAYS_GSTR
}

}  // namespace rays
