// Calibration of rocprofv3's WRITE_SIZE on the trace kernels' store pattern (developer tool, not product;
// DESIGN.md 4.5).  Every kernel writes the SAME number of bytes; only the pattern differs:
//   stream16   : 16 B per lane, consecutive lanes consecutive addresses (the pattern the guide calibrates)
//   rec56      : one lane = one ray; per step a 56-B record (3 x 16 B + 8 B) at ray*STRIDE + step*56 and an
//                8-B residual at ray*RSTRIDE + step*8 -- the RK4/SG kernels' record_point
//   rec56_slow : the same with ~9 us of FP64 work between steps (the real kernel's step time)
//   rec64      : 64-B records, 64-B aligned (4 x 16 B), no residual stream
// hipcc -O3 --offload-arch=gfx950 -o sp.bin store_pattern.hip ; rocprofv3 --pmc WRITE_SIZE --kernel-trace -- ./sp.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define NRAY 65536
#define NSTEP 192
#define NPT 1001
__global__ void __launch_bounds__(256) stream16(double2* out, double a) {
  const size_t n = (size_t)NRAY * NSTEP * 4;  // 64 B per ray-step
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = double2{a, a + i};
}
template <int SLOW>
__global__ void __launch_bounds__(256) rec56(double* ray_vec, double* residual, double a, const double* in) {
  const size_t ray = blockIdx.x * 256ull + threadIdx.x;
  double* rv = ray_vec + ray * NPT * 7;
  double* rs = residual + ray * NPT;
  double x = in[threadIdx.x & 7], m = in[8], c = in[9];
  for (int s = 0; s < NSTEP; s++) {
    if (SLOW)
      for (int i = 0; i < 3600; i++) x = __builtin_fma(x, m, c);
    double* p = rv + 7 * s;
    for (int i = 0; i < 7; i++) p[i] = a + x + i;
    rs[s] = x;
  }
}
__global__ void __launch_bounds__(256) rec64(double* ray_vec, double a) {
  const size_t ray = blockIdx.x * 256ull + threadIdx.x;
  double* rv = ray_vec + ray * NPT * 8;
  for (int s = 0; s < NSTEP; s++) {
    double* p = rv + 8 * s;
    for (int i = 0; i < 8; i++) p[i] = a + i;
  }
}
int main() {
  double *rv, *rs, *in;
  hipMalloc(&rv, sizeof(double) * NRAY * NPT * 8ull); hipMalloc(&rs, sizeof(double) * NRAY * NPT); hipMalloc(&in, 128);
  double h[16]; for (int i = 0; i < 16; i++) h[i] = 1.0000001; hipMemcpy(in, h, 128, hipMemcpyHostToDevice);
  const double mb = (double)NRAY * NSTEP * 64 / 1e6;
  printf("bytes written by every kernel: %.1f MB\n", mb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
#define TIME(name, launch) hipEventRecord(e0); launch; hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); printf("%-12s %8.3f ms  %7.1f GB/s\n", name, ms, mb / ms);
  for (int rep = 0; rep < 2; rep++) {
    TIME("stream16", (stream16<<<2048, 256>>>((double2*)rv, 1.0)));
    TIME("rec56", (rec56<0><<<NRAY / 256, 256>>>(rv, rs, 1.0, in)));
    TIME("rec56_slow", (rec56<1><<<NRAY / 256, 256>>>(rv, rs, 1.0, in)));
    TIME("rec64", (rec64<<<NRAY / 256, 256>>>(rv, 1.0)));
  }
  return 0;
}
