// TEST INFRASTRUCTURE ONLY: a one-lane-per-wave host emulation of the handful of HIP constructs the
// trace kernels use, so the *product kernel source* (rays_amd/csrc/rays_{rk4,sg}.hpp) can be
// compiled with g++ and run on the CPU under sanitizers (GPU ASan is not available on the pool)
// and compared with the oracle in the CPU test tier.  Never part of librays_hip.so.
#pragma once
#define RAYS_HOST_EMUL 1
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ thread_local   /* `extern __shared__ double lds[]` = this host thread's rays::lds */
#define __restrict__ __restrict

struct emul_dim3 { unsigned x = 1, y = 1, z = 1; };
extern thread_local emul_dim3 threadIdx, blockIdx, blockDim, gridDim;
// (thread_local: the emulated C ABI traces on several host threads at once, one per device slot)
// rays_emul_redo_steps: how many rays the RK4 kernel handed over to rk4_resume_ray (rays_rk4_body.inc:
// kStopResumeExact; emul_trace.cpp), so that a test can tell that a fixture exercises the hand-over
#include <atomic>
namespace rays { extern std::atomic<long long> rays_emul_redo_steps; }
#define RAYS_EMUL_DEFINE_GLOBALS \
  thread_local emul_dim3 threadIdx, blockIdx, blockDim, gridDim; \
  namespace rays { thread_local double lds[1 << 16]; std::atomic<long long> rays_emul_redo_steps{0}; }

#ifdef RAYS_EMUL_WAVE
#include "hip_wave_emul.h"   // 64 lanes per wave as fibers: __any / __ballot / __shfl are real cross-lane operations
#else
inline int __any(int x) { return x; }
inline unsigned long long __ballot(int x) { return x ? 1ull : 0ull; }
inline int __shfl(int x, int, int = 64) { return x; }  /* only lane 0 exists */
#endif
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
#define __builtin_amdgcn_readlane(v, r) ((r) == 0 ? (v) : 0)  /* only lane 0 exists */
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
using std::copysign; using std::fabs; using std::fmax; using std::fmin; using std::ilogb;
using std::pow; using std::scalbn; using std::sqrt; using std::exp;
#ifdef RAYS_EMUL_RUNTIME
#include "hip_runtime_api_emul.h"   // + the runtime API, for the emulated C ABI (emul_capi.cpp)
#endif
