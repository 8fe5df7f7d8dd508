// rays_inst.hip -- kernel instantiations.  Compiled once per (solver, equilibrium, derivative,
// unit-exponent) group:
//   -DRAYS_INST_SOLVER={0,1} -DRAYS_INST_EQ={0,1,2} -DRAYS_INST_DERIV={0,1} -DRAYS_INST_UE={0,1} -DRAYS_INST_MS={0,1}
//   -DRAYS_INST_EQT=<EQ + 4 UE + 8 MS + 16 TOL>  (the kernels' EQ template argument as a literal, for the kernel names)
//   -DRAYS_INST_TOL=1 -DRAYS_TOL_FLAVOUR -ffp-contract=fast: the tolerance flavour of a cold RK4 group
//   (rays_device.hpp: kEqTol; 1e-10 relative per step instead of bit-identity)
// `make FULL=1` (-DRAYS_INST_FULL): each group instantiates the species counts NS = 1..6 (nspec = 0..5,
// species_m.f90:25) and nv = 7 | 12 (integrate_eq_gradients) and 8 | 13 (+ damping) (ode_m.f90:160-173); the MS = 1
// groups (multi_spec_damping) nv = 8 + NS | 13 + NS.  The DEFAULT build holds electrons + one ion species (NS = 2) in
// every nv, plus the other species counts a BASELINE config or a fixture reaches (listed below): a quarter of the
// kernels, of the build time and of the library's size.  A configuration whose shape is not built is refused with a
// message that names `make FULL=1` (rays_capi.hip: find_kernel).
#include "rays_launch.hpp"
#if RAYS_INST_SOLVER == 0
#include "rays_rk4.hpp"
#else
#include "rays_sg.hpp"
#if RAYS_INST_DERIV == 1 && !RAYS_INST_MS
#include "rays_sg_group.hpp"   // finite-difference dD, nv = 7: one ray per group of lanes
#define RAYS_INST_GROUP 1
#endif
#endif

#ifndef RAYS_INST_MS
#define RAYS_INST_MS 0
#endif
#ifndef RAYS_INST_TOL
#define RAYS_INST_TOL 0
#endif
#if RAYS_INST_TOL && !(defined(RAYS_TOL_FLAVOUR) && RAYS_INST_SOLVER == 0 && RAYS_INST_DERIV == 0 && RAYS_INST_MS == 0)
#error "the tolerance flavour exists for the cold RK4 kernels only, compiled with -DRAYS_TOL_FLAVOUR"
#endif
#if !RAYS_INST_TOL && defined(RAYS_TOL_FLAVOUR)
#error "-DRAYS_TOL_FLAVOUR without -DRAYS_INST_TOL=1: an exact kernel name would get tolerance arithmetic"
#endif
#define RAYS_CAT_(a, b, c, d, e, f) a##_##b##_##c##_##d##_##e##_##f
#define RAYS_CAT(a, b, c, d, e, f) RAYS_CAT_(a, b, c, d, e, f)
// the kernels' EQ template argument, as a literal (it appears in the kernel names rocprof prints)
#ifndef RAYS_INST_EQT
#error "pass -DRAYS_INST_EQT=<RAYS_INST_EQ + 4 RAYS_INST_UE + 8 RAYS_INST_MS + 16 RAYS_INST_TOL>"
#endif
#define RAYS_STR_(x) #x
#define RAYS_STR(x) RAYS_STR_(x)

namespace rays {

namespace {
constexpr int EQ = RAYS_INST_EQT;
static_assert(EQ == (RAYS_INST_EQ | (RAYS_INST_UE ? kEqUnitExp : 0) | (RAYS_INST_MS ? kEqMultiSpec : 0) |
                     (RAYS_INST_TOL ? kEqTol : 0)), "EQ encoding");
constexpr int DERIV = RAYS_INST_DERIV;

template <int NS, int NV, int OCC = 1>
hipError_t launch_one(const DevParams& P, const TraceArgs& A, hipStream_t stream, int* grid_blocks) {
#if RAYS_INST_SOLVER == 0
  // (`if constexpr`: only the listed shapes get a two-waves-per-SIMD build compiled at all)
#ifndef RAYS_HOST_EMUL
  if constexpr (OCC == 2) {
#ifdef RAYS_RK4_W2_DIRECT_STORES
    constexpr size_t lds2 = 0;
#else
    constexpr size_t lds2 = PointWindow<NV, true>::kLdsBytes;  // residual(:) only
#endif
    return launch_persistent(rk4_trace_kernel_w2<EQ, NS, DERIV, NV>, lds2, P, A, stream, grid_blocks);
  } else
#endif
  {
#ifdef RAYS_RK4_DIRECT_STORES
    constexpr size_t lds = 0;
#else
    // rays_trace.hpp: the point window (0 unless nv = 7 | 8) + the eqdsk 1-D tables where there is room
    constexpr size_t lds = PointWindow<NV>::kLdsBytes + eq_tab_lds_bytes<EQ, NV>() + zf_tab_lds_bytes<EQ, NS, NV>();
#endif
    return launch_persistent(rk4_trace_kernel<EQ, NS, DERIV, NV>, lds, P, A, stream, grid_blocks);
  }
#else
  static_assert(kBlock / kWave == 4, "SgLds lays the block's four waves out");
  constexpr size_t lds = SgLds<NV>::kLdsBytes;
  return launch_persistent(sg_trace_kernel<EQ, NS, DERIV, NV>, lds, P, A, stream, grid_blocks);
#endif
}

#if RAYS_INST_SOLVER == 0 && RAYS_INST_DERIV == 0 && !RAYS_INST_TOL && !RAYS_INST_MS
// the continuation of rays the tolerance twin of this shape hands over (rays_rk4.hpp: rk4_resume_kernel)
template <int NS, int NV>
hipError_t resume_one(const DevParams& P, const TraceArgs& A, hipStream_t stream) {
  hipLaunchKernelGGL((rk4_resume_kernel<EQ, NS, DERIV, NV>), dim3((A.nray + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, P, A);
  return hipGetLastError();
}
#define RAYS_RESUME(NS, NV) , &resume_one<NS, NV>
#else
#define RAYS_RESUME(NS, NV)
#endif

#ifdef RAYS_INST_GROUP
#ifndef RAYS_SG_GROUP_LANES
#define RAYS_SG_GROUP_LANES 4
#endif
template <int NS>
hipError_t launch_group(const DevParams& P, const TraceArgs& A, hipStream_t stream, int* grid_blocks) {
  typedef GrpGeom<RAYS_SG_GROUP_LANES> GEO;
  static_assert(GEO::kThreads == kBlock, "the lane-group kernel is written for the launch block size");
  return launch_persistent(sg_group_kernel<EQ, NS, RAYS_SG_GROUP_LANES>, GEO::kLdsBytes, P, A, stream, grid_blocks,
                           GEO::kRaysPerBlock);
}
#define RAYS_ENTRY_GROUP(NS) \
  { RAYS_INST_SOLVER, EQ, NS, DERIV, 7, 1, sg_group_far_doubles_per_lane<RAYS_SG_GROUP_LANES>(), RAYS_SG_GROUP_LANES, "sg_group_kernel<" RAYS_STR(RAYS_INST_EQT) ", " #NS ", " RAYS_STR(RAYS_SG_GROUP_LANES) ">", &launch_group<NS> }
#endif
#if RAYS_INST_SOLVER == 0
#define RAYS_KNAME "rk4_trace_kernel"
#else
#define RAYS_KNAME "sg_trace_kernel"
#endif
#if RAYS_INST_SOLVER == 0
#define RAYS_SG_FAR(NV) 0
#else
#define RAYS_SG_FAR(NV) sg_far_doubles_per_lane<NV>()
#endif
#define RAYS_ENTRY(NS, NV) \
  { RAYS_INST_SOLVER, EQ, NS, DERIV, NV, 1, RAYS_SG_FAR(NV), 1, RAYS_KNAME "<" RAYS_STR(RAYS_INST_EQT) ", " #NS ", " RAYS_STR(RAYS_INST_DERIV) ", " #NV ">", &launch_one<NS, NV> RAYS_RESUME(NS, NV) }
// two-waves-per-SIMD build of an RK4 kernel (large fans; rays_rk4.hpp)
#define RAYS_ENTRY_OCC2(NS, NV) \
  { RAYS_INST_SOLVER, EQ, NS, DERIV, NV, 2, 0, 1, "rk4_trace_kernel_w2<" RAYS_STR(RAYS_INST_EQT) ", " #NS ", " RAYS_STR(RAYS_INST_DERIV) ", " #NV ">", &launch_one<NS, NV, 2> }

const KernelEntry kEntries[] = {
#if RAYS_INST_MS
    // multi_spec_damping: nv = 7 + 1 + (1 + nspec) (+ 5 with integrate_eq_gradients), ode_m.f90:160-173
#if defined(RAYS_INST_FAST) || !defined(RAYS_INST_FULL)
    RAYS_ENTRY(2, 10), RAYS_ENTRY(2, 15),
#else
    RAYS_ENTRY(1, 9), RAYS_ENTRY(2, 10), RAYS_ENTRY(3, 11), RAYS_ENTRY(4, 12), RAYS_ENTRY(5, 13), RAYS_ENTRY(6, 14),
    RAYS_ENTRY(1, 14), RAYS_ENTRY(2, 15), RAYS_ENTRY(3, 16), RAYS_ENTRY(4, 17), RAYS_ENTRY(5, 18), RAYS_ENTRY(6, 19),
#endif
#elif defined(RAYS_INST_FAST)  // developer builds (make FAST=1): electrons + one ion species only
    RAYS_ENTRY(2, 7), RAYS_ENTRY(2, 8),
#ifdef RAYS_INST_GROUP
    RAYS_ENTRY_GROUP(2),
#endif
#if RAYS_INST_SOLVER == 0 && RAYS_INST_EQ != 2 && !defined(RAYS_HOST_EMUL)
    RAYS_ENTRY_OCC2(2, 7),
#endif
#elif !defined(RAYS_INST_FULL)  // the default build: NS = 2 throughout + the shapes configs / fixtures reach
    RAYS_ENTRY(2, 7), RAYS_ENTRY(2, 12), RAYS_ENTRY(2, 8), RAYS_ENTRY(2, 13),
#ifdef RAYS_INST_GROUP
    RAYS_ENTRY_GROUP(2),
#endif
#if RAYS_INST_SOLVER == 0 && RAYS_INST_EQ != 2
    RAYS_ENTRY_OCC2(2, 7),
#endif
#if RAYS_INST_SOLVER == 0 && RAYS_INST_EQ == 0 && RAYS_INST_DERIV == 0 && RAYS_INST_UE == 1
    RAYS_ENTRY(1, 7), RAYS_ENTRY_OCC2(1, 7),   // electrons only (gold_slab_ns1_rk4)
#endif
#if RAYS_INST_SOLVER == 1 && RAYS_INST_EQ == 0 && RAYS_INST_DERIV == 1 && RAYS_INST_UE == 1
    RAYS_ENTRY(3, 7), RAYS_ENTRY_GROUP(3),     // three species (gold_slab_shear_gauss_3spec_sg_num)
#endif
#if RAYS_INST_SOLVER == 0 && RAYS_INST_EQ == 1 && RAYS_INST_DERIV == 1 && RAYS_INST_UE == 1
    RAYS_ENTRY(4, 7),                          // D + T + 3He (gold_solovev64_4spec_rk4_num)
#endif
#if RAYS_INST_SOLVER == 1 && RAYS_INST_EQ == 0 && RAYS_INST_DERIV == 0 && RAYS_INST_UE == 1
    RAYS_ENTRY(6, 7),                          // five ion species (gold_slab_6spec_sg)
#endif
#else
    RAYS_ENTRY(1, 7), RAYS_ENTRY(2, 7), RAYS_ENTRY(3, 7), RAYS_ENTRY(4, 7), RAYS_ENTRY(5, 7), RAYS_ENTRY(6, 7),
    RAYS_ENTRY(1, 12), RAYS_ENTRY(2, 12), RAYS_ENTRY(3, 12), RAYS_ENTRY(4, 12), RAYS_ENTRY(5, 12), RAYS_ENTRY(6, 12),
    // nv = 8 | 13: + total-absorption row (damping_model = 'damp_fund_ECH', ode_m.f90:162-166)
    RAYS_ENTRY(1, 8), RAYS_ENTRY(2, 8), RAYS_ENTRY(3, 8), RAYS_ENTRY(4, 8), RAYS_ENTRY(5, 8), RAYS_ENTRY(6, 8),
    RAYS_ENTRY(1, 13), RAYS_ENTRY(2, 13), RAYS_ENTRY(3, 13), RAYS_ENTRY(4, 13), RAYS_ENTRY(5, 13), RAYS_ENTRY(6, 13),
#ifdef RAYS_INST_GROUP
    // SG + finite-difference dD, nv = 7: one ray per group of lanes (rays_sg_group.hpp); preferred by find_kernel
    RAYS_ENTRY_GROUP(1), RAYS_ENTRY_GROUP(2), RAYS_ENTRY_GROUP(3), RAYS_ENTRY_GROUP(4), RAYS_ENTRY_GROUP(5), RAYS_ENTRY_GROUP(6),
#endif
#if RAYS_INST_SOLVER == 0 && RAYS_INST_EQ != 2
    // the common analytic-equilibrium shapes also built for two waves per SIMD (the eqdsk + damping
    // kernels lose at 128 VGPRs: 4.4 -> 7.0 ms on the 256k-ray eqdsk fan)
    RAYS_ENTRY_OCC2(1, 7), RAYS_ENTRY_OCC2(2, 7), RAYS_ENTRY_OCC2(3, 7),
#endif
#endif
};
}  // namespace

#if defined(RAYS_SG_PROFILE) && RAYS_INST_SOLVER == 1
// developer builds only: per-section wave clocks of this group's SG kernels
extern "C" int RAYS_CAT(rays_debug_sg_profile, RAYS_INST_SOLVER, RAYS_INST_EQ, RAYS_INST_DERIV, RAYS_INST_UE, RAYS_INST_MS)(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sg_prof), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sg_prof), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif

#if RAYS_INST_TOL
#define RAYS_ENTRIES_NAME rays_entries_tol
#else
#define RAYS_ENTRIES_NAME rays_entries
#endif
const KernelEntry* RAYS_CAT(RAYS_ENTRIES_NAME, RAYS_INST_SOLVER, RAYS_INST_EQ, RAYS_INST_DERIV, RAYS_INST_UE, RAYS_INST_MS)(int* n) {
  *n = (int)(sizeof(kEntries) / sizeof(kEntries[0]));
  return kEntries;
}

}  // namespace rays
