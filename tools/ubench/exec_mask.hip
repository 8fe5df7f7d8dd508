// Does a partially filled wave issue FP64 VALU work faster on gfx950?  (developer tool, not product;
// the answer decides whether long rays would finish sooner in waves of 16 lanes: DESIGN.md 4.5)
// hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o em exec_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 8192
__global__ void __launch_bounds__(256) k(double* out, const double* in, int lim, int first, int stride = 1) {
  double a = in[threadIdx.x], b = in[threadIdx.x + 256], c = in[threadIdx.x + 512], d = in[threadIdx.x + 768];
  const double m = in[1024 + (threadIdx.x & 1)], n = in[1030];
  const int lane = threadIdx.x & 63;
  if (lane >= first && lane < first + lim * stride && (lane - first) % stride == 0) {
    for (int i = 0; i < N; i++) {
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
int main() {
  double *out, *in;
  hipMalloc(&out, 8 * 256 * 512); hipMalloc(&in, 8 * 2048);
  std::vector<double> h(2048, 1.0000001);
  hipMemcpy(in, h.data(), 8 * 2048, hipMemcpyHostToDevice);
  const int lims[] = {64, 16, 12, 11, 10, 9, 8, 1}, firsts[] = {0, 20};
  for (int blocks : {256})
    for (int lim : lims)
      for (int first : firsts) {
        if (first + lim > 64 || (first && lim > 16)) continue;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<blocks, 256>>>(out, in, lim, first); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, 256>>>(out, in, lim, first); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD=%d lanes [%2d,%2d)  %.3f ms  ns per fma instruction = %.3f\n", blocks / 256, first, first + lim, ms, ms * 1e6 / (4.0 * N));
        fflush(stdout);
      }
  // scattered live lanes: n lanes, one every `stride`
  const int pat[][2] = {{8, 8}, {9, 7}, {10, 6}, {12, 5}, {16, 4}, {4, 16}, {2, 32}};
  for (auto& q : pat) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 256>>>(out, in, q[0], 0, q[1]); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<256, 256>>>(out, in, q[0], 0, q[1]); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD=1 %2d lanes, one every %2d  %.3f ms  ns per fma instruction = %.3f\n", q[0], q[1], ms, ms * 1e6 / (4.0 * N));
  }
  return 0;
}
