// rays_launch.hpp -- registry of compiled kernel specialisations.
//
// The reference dispatches on strings inside the ray loop; here each (solver, equilibrium,
// species count, derivative model, nv) combination is a separately compiled kernel, instantiated
// in rays_inst_*.hip (one translation unit per group so they build in parallel) and looked up once
// per call by rays_capi.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

#include "rays_device.hpp"
#include "rays_trace.hpp"

namespace rays {

enum { SOLVER_RK4 = 0, SOLVER_SG = 1 };

struct KernelEntry {
  int solver, eq, ns, deriv, nv;  // eq = equilibrium model | kEqUnitExp (the kernels' EQ argument)
  int occ;                        // waves per SIMD the kernel is built for (RK4: 1 and, for the common shapes, 2)
  int sg_far_per_lane;            // SG: doubles per lane of the upper-tier workspace (TraceArgs::sg_far); RK4: 0
  int lanes_per_ray;              // 1, or G for the lane-group SG kernel (rays_sg_group.hpp)
  const char* name;
  // Launches on `stream` with a grid sized for full residency (persistent waves + lane refill).
  hipError_t (*launch)(const DevParams&, const TraceArgs&, hipStream_t stream, int* grid_blocks);
  // exact cold RK4 shapes: rk4_resume_kernel, the continuation of the rays their TOLERANCE twin hands over
  // (rays_rk4_body.inc: kStopResumeExact); null elsewhere
  hipError_t (*resume)(const DevParams&, const TraceArgs&, hipStream_t stream);
};

constexpr int kBlock = 256;

// Compute units of a device, asked once per process (hipGetDeviceProperties is slow; this sits on the
// launch path of every trace).
inline int device_cu_count(int dev) {
  static std::mutex mu;
  static int cus[64] = {0};
  std::lock_guard<std::mutex> lk(mu);
  if (dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

// Occupancy of a kernel on the current device, cached per (kernel, device) for the whole process
// (rays_hip_trace runs each device's share on a fresh host thread, so a thread-local cache would be
// rebuilt on every call).  All trace kernels have the same function type, so the cache is keyed by
// the function's address, not by the template instantiation.
struct Occupancy {
  int blocks_per_cu, cus;
};
inline std::mutex& occupancy_mutex() {
  static std::mutex mu;
  return mu;
}
inline std::map<std::pair<const void*, int>, Occupancy>& occupancy_cache() {
  static std::map<std::pair<const void*, int>, Occupancy> cache;
  return cache;
}
template <typename Kernel>
inline hipError_t kernel_occupancy(Kernel kernel, size_t lds_bytes, Occupancy* out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const auto key = std::make_pair(reinterpret_cast<const void*>(kernel), dev);
  std::lock_guard<std::mutex> lk(occupancy_mutex());
  auto& cache = occupancy_cache();
  auto it = cache.find(key);
  if (it == cache.end()) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes);
    if (e != hipSuccess) return e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds_bytes);
    if (e != hipSuccess) return e;
    it = cache.emplace(key, Occupancy{per_cu > 0 ? per_cu : 1, device_cu_count(dev)}).first;
  }
  *out = it->second;
  return hipSuccess;
}

// rays_per_block: rays a block starts with (kBlock: one ray per lane; kBlock / G for the lane-group kernels)
template <typename Kernel>
inline hipError_t launch_persistent(Kernel kernel, size_t lds_bytes, const DevParams& P,
                                    const TraceArgs& A, hipStream_t stream, int* grid_blocks, int rays_per_block = kBlock) {
  Occupancy occ;
  hipError_t e = kernel_occupancy(kernel, lds_bytes, &occ);
  if (e != hipSuccess) return e;
  const int cached_blocks_per_cu = occ.blocks_per_cu, cached_cus = occ.cus;
  const long long need = ((long long)A.nray + rays_per_block - 1) / rays_per_block;
  long long resident = (long long)cached_blocks_per_cu * cached_cus;
  int blocks = (int)(need < resident ? need : resident);
  if (A.sg_far && (long long)blocks * kBlock > A.sg_far_lanes) blocks = (int)(A.sg_far_lanes / kBlock);
  if (blocks < 1) blocks = 1;
  if (grid_blocks) *grid_blocks = blocks;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBlock), lds_bytes, stream, P, A);
  return hipGetLastError();
}


}  // namespace rays
