// rays_launch.hpp -- registry of compiled kernel specialisations.
//
// The reference dispatches on strings inside the ray loop; here each (solver, equilibrium,
// species count, derivative model, nv) combination is a separately compiled kernel, instantiated
// in rays_inst_*.hip (one translation unit per group so they build in parallel) and looked up once
// per call by rays_capi.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <utility>

#include "rays_device.hpp"
#include "rays_trace.hpp"

namespace rays {

enum { SOLVER_RK4 = 0, SOLVER_SG = 1 };

struct KernelEntry {
  int solver, eq, ns, deriv, nv;  // eq = equilibrium model | kEqUnitExp (the kernels' EQ argument)
  int occ;                        // waves per SIMD the kernel is built for (RK4: 1 and, for the common shapes, 2)
  const char* name;
  // Launches on `stream` with a grid sized for full residency (persistent waves + lane refill).
  hipError_t (*launch)(const DevParams&, const TraceArgs&, hipStream_t stream, int* grid_blocks);
};

constexpr int kBlock = 256;

// Occupancy of a kernel on the current device, cached per (kernel, device).  (All trace kernels have
// the same function type, so the cache must be keyed by the function's address, not by the
// template instantiation.)
struct Occupancy {
  int blocks_per_cu, cus;
};
template <typename Kernel>
inline hipError_t kernel_occupancy(Kernel kernel, size_t lds_bytes, Occupancy* out) {
  static thread_local std::map<std::pair<const void*, int>, Occupancy> cache;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const auto key = std::make_pair(reinterpret_cast<const void*>(kernel), dev);
  auto it = cache.find(key);
  if (it == cache.end()) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes);
    if (e != hipSuccess) return e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds_bytes);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    it = cache.emplace(key, Occupancy{per_cu > 0 ? per_cu : 1, prop.multiProcessorCount}).first;
  }
  *out = it->second;
  return hipSuccess;
}

template <typename Kernel>
inline hipError_t launch_persistent(Kernel kernel, size_t lds_bytes, const DevParams& P,
                                    const TraceArgs& A, hipStream_t stream, int* grid_blocks) {
  Occupancy occ;
  hipError_t e = kernel_occupancy(kernel, lds_bytes, &occ);
  if (e != hipSuccess) return e;
  const int cached_blocks_per_cu = occ.blocks_per_cu, cached_cus = occ.cus;
  const long long need = ((long long)A.nray + kBlock - 1) / kBlock;
  long long resident = (long long)cached_blocks_per_cu * cached_cus;
  int blocks = (int)(need < resident ? need : resident);
  if (blocks < 1) blocks = 1;
  if (grid_blocks) *grid_blocks = blocks;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBlock), lds_bytes, stream, P, A);
  return hipGetLastError();
}


}  // namespace rays
