// rays_pack.hip -- compact ("CSR") form of the trajectory arrays for the multi-GPU gather.
//
// The reference layout pads every ray to nstep_max+1 points (ray_results_m.f90:44-46); a Solovev
// fan fills ~20 % of it.  Before the RCCL exchange each rank packs its slab to
//   packed_vec[sum(npoints)][nv], packed_res[sum(npoints)]   (rays in order, points in order)
// and the root unpacks every peer's block straight into the padded global arrays.  One wave per
// ray, lanes stride over the ray's contiguous run: both sides of the copy are coalesced.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rays {

template <bool PACK>
__global__ void __launch_bounds__(256)
pack_kernel(int nray, int nv, long long npt, const int32_t* __restrict__ npoints,
            const long long* __restrict__ offsets,  // exclusive prefix sum of npoints
            double* __restrict__ ray_vec, double* __restrict__ residual,
            double* __restrict__ packed_vec, double* __restrict__ packed_res) {
  const int lane = threadIdx.x & 63;
  const int waves_per_block = blockDim.x >> 6;
  for (long long r = (long long)blockIdx.x * waves_per_block + (threadIdx.x >> 6); r < nray;
       r += (long long)gridDim.x * waves_per_block) {
    const long long n = npoints[r], off = offsets[r];
    double* pv = ray_vec + r * npt * nv;
    double* pr = residual + r * npt;
    double* qv = packed_vec + off * nv;
    double* qr = packed_res + off;
    for (long long e = lane; e < n * nv; e += 64) {
      if (PACK) qv[e] = pv[e]; else pv[e] = qv[e];
    }
    for (long long e = lane; e < n; e += 64) {
      if (PACK) qr[e] = pr[e]; else pr[e] = qr[e];
    }
  }
}

hipError_t launch_pack(bool pack, int nray, int nv, int nstep_max, const int32_t* npoints,
                       const long long* offsets, double* ray_vec, double* residual, double* packed_vec,
                       double* packed_res, hipStream_t stream) {
  if (nray <= 0) return hipSuccess;
  int blocks = (nray + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const long long npt = (long long)nstep_max + 1;
  if (pack)
    hipLaunchKernelGGL(pack_kernel<true>, dim3(blocks), dim3(256), 0, stream, nray, nv, npt, npoints, offsets,
                       ray_vec, residual, packed_vec, packed_res);
  else
    hipLaunchKernelGGL(pack_kernel<false>, dim3(blocks), dim3(256), 0, stream, nray, nv, npt, npoints, offsets,
                       ray_vec, residual, packed_vec, packed_res);
  return hipGetLastError();
}

}  // namespace rays
