// What one DEPENDENT memory access costs a lone wave per SIMD on gfx950 (developer tool, not product; the numbers behind
// DESIGN.md 6 "fewer dependent table accesses"): scalar load from the kernarg / constant cache, LDS read, global load that
// hits L2 (a 2 MB table, random cells) -- each issued, waited for, and its result used for the next address.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ml tools/ubench/mem_latency.hip && /tmp/ml
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 2048
__global__ void __launch_bounds__(256) k_smem(const unsigned* __restrict__ tab, unsigned* out, long long* clk) {
  unsigned idx = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N; i++) {
    unsigned v;
    const unsigned* p = tab + (idx & 255u);  // uniform address -> s_load
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    idx = v + (unsigned)i;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = idx; clk[0] = t1 - t0; }
}
__global__ void __launch_bounds__(256) k_lds(const unsigned* __restrict__ tab, unsigned* out, long long* clk) {
  __shared__ unsigned s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = tab[i & 255];
  __syncthreads();
  unsigned idx = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N; i++) idx = (s[idx & 4095u] + idx + 64u);
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = idx;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[1] = t1 - t0;
}
__global__ void __launch_bounds__(256) k_l2(const unsigned* __restrict__ big, unsigned nwords, unsigned* out, long long* clk) {
  unsigned idx = (blockIdx.x * 256 + threadIdx.x) * 977u;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N; i++) idx = big[(idx % nwords)] + idx * 31u + 7u;
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = idx;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[2] = t1 - t0;
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  const unsigned nwords = 512 * 1024;  // 2 MB
  std::vector<unsigned> h(nwords);
  for (unsigned i = 0; i < nwords; i++) h[i] = i * 2654435761u >> 7;
  unsigned *tab, *big, *out; long long* clk;
  hipMalloc(&tab, 4096 * 4); hipMalloc(&big, nwords * 4); hipMalloc(&out, 4 * 256 * cus); hipMalloc(&clk, 64);
  hipMemcpy(tab, h.data(), 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(big, h.data(), nwords * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    k_smem<<<cus, 256>>>(tab, out, clk); k_lds<<<cus, 256>>>(tab, out, clk); k_l2<<<cus, 256>>>(big, nwords, out, clk);
    hipDeviceSynchronize();
  }
  long long c[3]; hipMemcpy(c, clk, 24, hipMemcpyDeviceToHost);
  // s_memtime counts at a constant 100 MHz on gfx9; convert with the wall clock of a known loop? report ticks and ns
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[3];
  hipEventRecord(e0); k_smem<<<cus, 256>>>(tab, out, clk); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
  hipEventRecord(e0); k_lds<<<cus, 256>>>(tab, out, clk); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
  hipEventRecord(e0); k_l2<<<cus, 256>>>(big, nwords, out, clk); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
  const char* names[3] = {"scalar load (constant cache hit) + wait", "LDS read + wait", "global load (2 MB table, L2) + wait"};
  for (int i = 0; i < 3; i++)
    printf("%-42s %7.1f s_memtime ticks, %7.1f ns per dependent access (kernel %.3f ms / %d)\n", names[i], (double)c[i] / N,
           ms[i] * 1e6 / N, ms[i], N);
  return 0;
}
