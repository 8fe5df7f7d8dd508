// rays_deposition.hip -- kernels of the device-side deposition profiles (see rays_deposition.hpp)
#include <hip/hip_runtime.h>

#include "rays_deposition.hpp"

namespace rays {

namespace {
constexpr int kDepWave = 64;

// one wave per block, one ray per lane; the ray's bins live in LDS (bin-major, lane-interleaved:
// conflict-free) while the binner walks the ray, then go to work[bin][ray] with coalesced stores
__global__ void __launch_bounds__(kDepWave) deposit_rays_kernel(const DevParams P, const DepArgs D) {
  extern __shared__ double rows[];  // [n_bins][64]
  typedef __attribute__((address_space(3))) double* lds_ptr;
  const int lane = threadIdx.x;
  const int iray = blockIdx.x * kDepWave + lane;
  if (iray >= D.nray) return;
  lds_ptr row = (lds_ptr)(rows + lane);
  deposit_ray(P, D, iray, row, kDepWave);
  for (int b = 0; b < D.n_bins; b++) D.work[(long long)b * D.nray + iray] = row[b * kDepWave];
}

// profile(b) = carry(b) + work(b, 1) + work(b, 2) + ... in ray order (sum(work, 2), continued).
// One wave per bin.  Each pass the lanes load U x 64 consecutive rays (coalesced); the running sum
// then takes them one by one: lane l of chunk u is ray base + 64 u + l.
__global__ void __launch_bounds__(kDepWave)
profile_sum_kernel(int n_bins, int nray, const double* __restrict__ work, const double* __restrict__ carry,
                   double* __restrict__ profile) {
  constexpr int U = 8;
  const int b = blockIdx.x, lane = threadIdx.x;
  const double* w = work + (long long)b * nray;
  double s = carry ? carry[b] : 0.;
  for (int base = 0; base < nray; base += U * kDepWave) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = base + u * kDepWave + lane;
      v[u] = r < nray ? w[r] : 0.;  // + 0.0 is an exact no-op here (s is never -0.0)
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned lo = (unsigned)__double_as_longlong(v[u]);
      const unsigned hi = (unsigned)(__double_as_longlong(v[u]) >> 32);
#pragma unroll
      for (int l = 0; l < kDepWave; l++) {
        const unsigned a = __builtin_amdgcn_readlane(lo, l), c = __builtin_amdgcn_readlane(hi, l);
        s = s + __longlong_as_double((long long)(((unsigned long long)c << 32) | a));
      }
    }
  }
  if (lane == 0) profile[b] = s;
}
}  // namespace

hipError_t launch_deposition(const DevParams& P, const DepArgs& D, const double* carry, double* profile,
                             hipStream_t s) {
  const size_t lds = sizeof(double) * (size_t)D.n_bins * kDepWave;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static thread_local int attr_dev = -1;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev != attr_dev) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(deposit_rays_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_dev = dev;
  }
  if (D.nray > 0) {
    hipLaunchKernelGGL(deposit_rays_kernel, dim3((D.nray + kDepWave - 1) / kDepWave), dim3(kDepWave), lds, s, P, D);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(profile_sum_kernel, dim3(D.n_bins), dim3(kDepWave), 0, s, D.n_bins, D.nray, D.work, carry, profile);
  return hipGetLastError();
}

}  // namespace rays
