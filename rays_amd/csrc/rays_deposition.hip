// rays_deposition.hip -- kernels of the device-side deposition profiles (see rays_deposition.hpp)
#include <hip/hip_runtime.h>

#include "rays_deposition.hpp"

namespace rays {

__global__ void __launch_bounds__(64) deposit_rays_kernel(const DevParams P, const DepArgs D) {
  const int iray = blockIdx.x * blockDim.x + threadIdx.x;
  if (iray < D.nray) deposit_ray(P, D, iray, D.work + (long long)iray * D.n_bins);
}

// profile(b) = carry(b) + work(b, 1) + work(b, 2) + ... in ray order (sum(work, 2), continued)
__global__ void __launch_bounds__(64)
profile_sum_kernel(int n_bins, int nray, const double* __restrict__ work, const double* __restrict__ carry,
                   double* __restrict__ profile) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_bins) return;
  double s = carry ? carry[b] : 0.;
  for (int r = 0; r < nray; r++) s = s + work[(long long)r * n_bins + b];
  profile[b] = s;
}

hipError_t launch_deposition(const DevParams& P, const DepArgs& D, const double* carry, double* profile,
                             hipStream_t s) {
  hipLaunchKernelGGL(deposit_rays_kernel, dim3((D.nray + 63) / 64), dim3(64), 0, s, P, D);
  hipLaunchKernelGGL(profile_sum_kernel, dim3((D.n_bins + 63) / 64), dim3(64), 0, s, D.n_bins, D.nray, D.work, carry,
                     profile);
  return hipGetLastError();
}

}  // namespace rays
