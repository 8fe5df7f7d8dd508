// Micro-benchmarks of single-wave-per-SIMD issue behaviour on gfx950 (developer tool, not product;
// results quoted in DESIGN.md 4.5).  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o ub issue_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, const double* in, long long* cyc) {
  double a = in[threadIdx.x], b = in[threadIdx.x + 256], c = in[threadIdx.x + 512], d = in[threadIdx.x + 768];
  double e = a + 1, f = b + 1, g = c + 1, h = d + 1;
  const double m = in[1024 + (threadIdx.x & 1)], n = in[1030];
  unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0; int v0 = threadIdx.x, v1 = 0, v2 = 0, v3 = 0; const int lane = threadIdx.x; unsigned long long sl = 0;
  __shared__ double lbuf[256]; lbuf[threadIdx.x] = a; __syncthreads();
  long long t0 = clock64();
  for (int i = 0; i < N; i++) {
    if (MODE == 0) {  // dependent fma chain (4 per iter)
      a = __builtin_fma(a, m, n); a = __builtin_fma(a, m, n); a = __builtin_fma(a, m, n); a = __builtin_fma(a, m, n);
    } else if (MODE == 1) {  // 2 independent chains
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n);
    } else if (MODE == 2) {  // 4 independent chains
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
    } else if (MODE == 3) {  // 8 independent
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      e = __builtin_fma(e, m, n); f = __builtin_fma(f, m, n); g = __builtin_fma(g, m, n); h = __builtin_fma(h, m, n);
    } else if (MODE == 4) {  // dependent mul chain
      a = a * m; a = a * m; a = a * m; a = a * m;
    } else if (MODE == 5) {  // dependent add chain
      a = a + m; a = a + m; a = a + m; a = a + m;
    } else if (MODE == 6) {  // rcp chain (4)
      a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a);
    } else if (MODE == 7) {  // independent rcp (4)
      a = __builtin_amdgcn_rcp(a); b = __builtin_amdgcn_rcp(b); c = __builtin_amdgcn_rcp(c); d = __builtin_amdgcn_rcp(d);
    } else if (MODE == 8) {  // div_fixup dependent
      a = __builtin_amdgcn_div_fixup(a, m, n); a = __builtin_amdgcn_div_fixup(a, m, n); a = __builtin_amdgcn_div_fixup(a, m, n); a = __builtin_amdgcn_div_fixup(a, m, n);
    } else if (MODE == 9) {  // 4 independent div_fixup
      a = __builtin_amdgcn_div_fixup(a, m, n); b = __builtin_amdgcn_div_fixup(b, m, n); c = __builtin_amdgcn_div_fixup(c, m, n); d = __builtin_amdgcn_div_fixup(d, m, n);
    } else if (MODE == 10) {  // sqrt (correctly rounded, ocml) x1 dependent
      a = sqrt(a) + m;
    } else if (MODE == 11) {  // full IEEE division dependent x1
      a = a / m + n;
    } else if (MODE == 12) {  // readlane + use  (4)
      int x = __builtin_amdgcn_readlane((int)i, i & 63);
      a = a + (double)x; 
    } else if (MODE == 13) {  // 4 fma + 4 s_add
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
    } else if (MODE == 14) {  // 4 fma + 4 v_mov_b32
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(i));
    } else if (MODE == 15) {  // 4 fma + 4 v_readlane
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 4\n v_readlane_b32 %2, %4, 5\n v_readlane_b32 %3, %4, 6" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(v0));
    } else if (MODE == 16) {  // 4 fma + 4 cndmask
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(i), "v"(lane) : "vcc");
    } else if (MODE == 17) {  // 4 fma + s_load + wait
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(sl) : "s"(in));
    } else if (MODE == 18) {  // 4 fma + 4 v_mov_b64
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4" : "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(a));
    } else if (MODE == 19) {  // 4 fma + 4 taken s_branch to next
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("s_branch 1f\n1:\n s_branch 2f\n2:\n s_branch 3f\n3:\n s_branch 4f\n4:");
    } else if (MODE == 20) {  // 4 fma + v_cmp + cbranch not taken
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("v_cmp_gt_f64 vcc, %0, %0\n s_cbranch_vccnz 1f\n1:" ::"v"(a) : "vcc");
    } else if (MODE == 21) {  // 4 fma + ds_read + wait
      a = __builtin_fma(a, m, n); b = __builtin_fma(b, m, n); c = __builtin_fma(c, m, n); d = __builtin_fma(d, m, n);
      asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(lane * 8));
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (double)(s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3) + (double)sl + lbuf[(threadIdx.x + 1) & 255];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int per_iter) {
  printf("start %s\n", name); fflush(stdout);
  double *out, *in; long long* cyc;
  hipMalloc(&out, 8 * 256 * 256 * 2); hipMalloc(&in, 8 * 2048); hipMalloc(&cyc, 8);
  std::vector<double> h(2048, 1.0000001);
  hipMemcpy(in, h.data(), 8 * 2048, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {  // 1 wave / SIMD, 2 waves / SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, in, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, in, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD=%d  %.3f ms  clock64/iter=%.2f  ns/iter=%.2f (%d ops/iter)\n", name, blocks / 256, ms, (double)c / N, ms * 1e6 / N, per_iter); fflush(stdout);
  }
}
int main() {
  run<2>("fma 4 chains x4", 4);
  run<13>("4 fma + 4 s_add", 8); run<14>("4 fma + 4 v_mov_b32", 8); run<15>("4 fma + 4 v_readlane", 8);
  run<16>("4 fma + 4 v_cndmask", 8); run<17>("4 fma + s_load+wait", 5); run<18>("4 fma + 4 v_mov_b64", 8); run<19>("4 fma + 4 s_branch", 8);
  run<20>("4 fma + v_cmp+cbranch", 6); run<21>("4 fma + ds_read+wait", 5);
  return 0;
}
