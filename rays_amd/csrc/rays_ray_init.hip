// rays_ray_init.hip -- kernels of the device-side ray initialisation (see rays_ray_init.hpp).
// Three passes: (1) one thread per fan member -> candidate index vector + keep flag and the
// per-block survivor count; (2) exclusive scan of the block counts; (3) ordered scatter of the
// survivors, so rays are numbered exactly as the reference's nested loops number them.
#include <hip/hip_runtime.h>

#include "rays_ray_init.hpp"

namespace rays {

namespace {
constexpr int kInitBlock = 256;

template <int EQ, int NS>
__global__ void __launch_bounds__(kInitBlock)
fan_candidates_kernel(const DevParams P, const FanArgs F, int n_cand, double* __restrict__ cand,
                      int* __restrict__ keep, int* __restrict__ block_count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  int k = 0;
  if (t < n_cand) {
    const int per_launch = F.n_a * F.n_b;
    const int il = t / per_launch, rem = t - il * per_launch;
    const int ia = rem / F.n_b, ib = rem - ia * F.n_b;
    const double rvec[3] = {F.launch[3 * il], F.launch[3 * il + 1], F.launch[3 * il + 2]};
    double ri[3] = {0., 0., 0.};
    k = fan_member<EQ, NS>(P, F, rvec, ia, ib, ri) ? 1 : 0;
    cand[3ll * t] = ri[0];
    cand[3ll * t + 1] = ri[1];
    cand[3ll * t + 2] = ri[2];
    keep[t] = k;
  }
  __shared__ int wave_sum[kInitBlock / 64];
  const int n = __popcll(__ballot(k));
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x / 64] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kInitBlock / 64; w++) s += wave_sum[w];
    block_count[blockIdx.x] = s;
  }
}

// exclusive scan of the block counts (<= a few thousand entries: one thread); offs[nblocks] = total
__global__ void scan_blocks_kernel(const int* __restrict__ block_count, int nblocks, int* __restrict__ offs) {
  int s = 0;
  for (int b = 0; b < nblocks; b++) {
    offs[b] = s;
    s += block_count[b];
  }
  offs[nblocks] = s;
}

__global__ void __launch_bounds__(kInitBlock)
fan_scatter_kernel(const FanArgs F, int n_cand, const double* __restrict__ cand, const int* __restrict__ keep,
                   const int* __restrict__ offs, double* __restrict__ rvec0, double* __restrict__ rindex_vec0,
                   int* __restrict__ first_of_launch) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = t < n_cand ? keep[t] : 0;
  __shared__ int wave_sum[kInitBlock / 64];
  const unsigned long long m = __ballot(k);
  const int lane = threadIdx.x & 63, wave = threadIdx.x / 64;
  if (lane == 0) wave_sum[wave] = __popcll(m);
  __syncthreads();
  int base = offs[blockIdx.x];
  for (int w = 0; w < wave; w++) base += wave_sum[w];
  const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
  if (t < n_cand) {
    const int per_launch = F.n_a * F.n_b;
    const int il = t / per_launch;
    if (t == il * per_launch) first_of_launch[il] = pos;  // survivors before this launch position
    if (k) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        rvec0[3ll * pos + i] = F.launch[3 * il + i];
        rindex_vec0[3ll * pos + i] = cand[3ll * t + i];
      }
    }
  }
}

template <int EQ, int NS>
hipError_t run_candidates(const DevParams& P, const FanArgs& F, int n_cand, double* cand, int* keep,
                          int* block_count, hipStream_t s) {
  const int nb = (n_cand + kInitBlock - 1) / kInitBlock;
  hipLaunchKernelGGL((fan_candidates_kernel<EQ, NS>), dim3(nb), dim3(kInitBlock), 0, s, P, F, n_cand, cand, keep,
                     block_count);
  return hipGetLastError();
}
template <int EQ>
hipError_t run_candidates_ns(int ns, const DevParams& P, const FanArgs& F, int n_cand, double* cand, int* keep,
                             int* block_count, hipStream_t s) {
  switch (ns) {
    case 1: return run_candidates<EQ, 1>(P, F, n_cand, cand, keep, block_count, s);
    case 2: return run_candidates<EQ, 2>(P, F, n_cand, cand, keep, block_count, s);
    case 3: return run_candidates<EQ, 3>(P, F, n_cand, cand, keep, block_count, s);
    case 4: return run_candidates<EQ, 4>(P, F, n_cand, cand, keep, block_count, s);
    case 5: return run_candidates<EQ, 5>(P, F, n_cand, cand, keep, block_count, s);
    default: return run_candidates<EQ, 6>(P, F, n_cand, cand, keep, block_count, s);
  }
}
}  // namespace

// cand[n_cand][3], keep[n_cand], block_count[nblocks], offs[nblocks + 1], first_of_launch[n_launch]
// are caller-provided device work arrays.  Asynchronous on `s`; the survivor count is offs[nblocks].
hipError_t launch_ray_init(int eq_model, int ns, const DevParams& P, const FanArgs& F, int n_cand, double* cand,
                           int* keep, int* block_count, int* offs, int* first_of_launch, double* rvec0,
                           double* rindex_vec0, hipStream_t s) {
  hipError_t e;
  if (eq_model == RAYS_EQ_SLAB)
    e = run_candidates_ns<RAYS_EQ_SLAB>(ns, P, F, n_cand, cand, keep, block_count, s);
  else if (eq_model == RAYS_EQ_SOLOVEV)
    e = run_candidates_ns<RAYS_EQ_SOLOVEV>(ns, P, F, n_cand, cand, keep, block_count, s);
  else
    e = run_candidates_ns<RAYS_EQ_AXISYM>(ns, P, F, n_cand, cand, keep, block_count, s);
  if (e != hipSuccess) return e;
  const int nb = (n_cand + kInitBlock - 1) / kInitBlock;
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1), 0, s, block_count, nb, offs);
  hipLaunchKernelGGL(fan_scatter_kernel, dim3(nb), dim3(kInitBlock), 0, s, F, n_cand, cand, keep, offs, rvec0,
                     rindex_vec0, first_of_launch);
  return hipGetLastError();
}

int ray_init_block() { return kInitBlock; }

}  // namespace rays
