// TEST INFRASTRUCTURE ONLY: host emulation of the part of the HIP runtime API that rays_capi.hip uses, so that
// the C ABI's host logic -- device lists and cache slots, streams, the blocks of rays_hip_trace, the multi-device
// gather of rays_hip_trace_gather -- runs on the CPU under ASan + UBSan with SEVERAL emulated devices (the GPU pool
// hands out one GPU per call, so the multi-device branches never run on hardware there).  Included by
// hip_runtime.h when RAYS_EMUL_RUNTIME is defined.  Never part of librays_hip.so.
//
// What it models, because the real runtime enforces it and the library's correctness depends on it:
//   * a current device per host thread (hipSetDevice / hipGetDevice), RAYS_EMUL_DEVICES devices (default 4);
//   * every allocation and every stream belongs to the device that was current when it was created;
//   * work submitted to a stream of another device than the current one fails with hipErrorInvalidHandle
//     (HIP: hipErrorInvalidResourceHandle), as does a copy / memset that touches another device's memory through
//     it -- the failure mode of a cache slot that kept a stream of the device it served before;
//   * kernels run at once on the calling thread, block by block and lane by lane (one lane per wave, as in
//     emul_trace.cpp); streams are therefore always idle and events always complete.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

enum hipError_t {
  hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101,
  hipErrorInvalidHandle = 400, hipErrorNotReady = 600, hipErrorNotSupported = 801
};
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipHostMallocPortable = 1, hipEventDisableTiming = 2 };
struct emul_stream { int device; };
struct emul_event { int device; };
typedef emul_stream* hipStream_t;
typedef emul_event* hipEvent_t;
struct hipDeviceProp_t { int multiProcessorCount; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hip_emul {
struct Alloc { size_t bytes; int device; };
struct State {
  std::mutex mu;
  std::map<const char*, Alloc> allocs;   // device memory: base -> (size, owner)
  std::map<int, emul_stream> null_stream;
  long long launches = 0, wrong_device = 0;
};
inline State& state() { static State s; return s; }
inline int& current() { static thread_local int d = 0; return d; }
inline hipError_t& last() { static thread_local hipError_t e = hipSuccess; return e; }
inline int device_count() {
  const char* e = std::getenv("RAYS_EMUL_DEVICES");
  const int n = e ? std::atoi(e) : 4;
  return n < 0 ? 0 : n;
}
inline hipError_t set(hipError_t e) { if (e != hipSuccess) last() = e; return e; }
// the device that owns [p, p + bytes), or -1 if it is not (entirely) device memory
inline int owner(const void* p, size_t bytes) {
  State& s = state();
  std::lock_guard<std::mutex> lk(s.mu);
  auto it = s.allocs.upper_bound((const char*)p);
  if (it == s.allocs.begin()) return -1;
  --it;
  const char* q = (const char*)p;
  if (q < it->first || q + bytes > it->first + it->second.bytes) return -1;
  return it->second.device;
}
inline int stream_device(hipStream_t st) { return st ? st->device : current(); }
// work on `st` touching device memory [p, p + bytes): the stream must be the current device's, the memory too
inline hipError_t check(hipStream_t st, const void* p, size_t bytes, const char* what) {
  if (stream_device(st) != current()) {
    state().wrong_device++;
    std::fprintf(stderr, "[hip_emul] %s: stream of device %d used while device %d is current\n", what, stream_device(st), current());
    return hipErrorInvalidHandle;
  }
  if (p && bytes) {
    const int o = owner(p, bytes);
    if (o < 0) { std::fprintf(stderr, "[hip_emul] %s: %p + %zu is not device memory\n", what, p, bytes); return hipErrorInvalidValue; }
    if (o != current()) {
      state().wrong_device++;
      std::fprintf(stderr, "[hip_emul] %s: memory of device %d used while device %d is current\n", what, o, current());
      return hipErrorInvalidHandle;
    }
  }
  return hipSuccess;
}
}  // namespace hip_emul

inline const char* hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorInvalidHandle: return "invalid resource handle";
    case hipErrorNotSupported: return "operation not supported";
    default: return "error";
  }
}
inline hipError_t hipGetLastError() { const hipError_t e = hip_emul::last(); hip_emul::last() = hipSuccess; return e; }
inline hipError_t hipGetDeviceCount(int* n) { *n = hip_emul::device_count(); return hipSuccess; }
inline hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= hip_emul::device_count()) return hip_emul::set(hipErrorInvalidDevice);
  hip_emul::current() = d;
  return hipSuccess;
}
inline hipError_t hipGetDevice(int* d) { *d = hip_emul::current(); return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 2; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

template <class T>
inline hipError_t hipMalloc(T** out, size_t bytes) {
  void* p = std::malloc(bytes ? bytes : 1);
  if (!p) return hip_emul::set(hipErrorOutOfMemory);
  std::memset(p, 0xA5, bytes);  // device memory is not zero: reading what was never written shows
  hip_emul::State& s = hip_emul::state();
  std::lock_guard<std::mutex> lk(s.mu);
  s.allocs[(const char*)p] = hip_emul::Alloc{bytes ? bytes : 1, hip_emul::current()};
  *out = (T*)p;
  return hipSuccess;
}
inline hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  hip_emul::State& s = hip_emul::state();
  {
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.allocs.find((const char*)p);
    if (it == s.allocs.end()) return hip_emul::set(hipErrorInvalidValue);
    s.allocs.erase(it);
  }
  std::free(p);
  return hipSuccess;
}
inline hipError_t hipHostMalloc(void** out, size_t bytes, unsigned) {
  *out = std::malloc(bytes ? bytes : 1);
  return *out ? hipSuccess : hip_emul::set(hipErrorOutOfMemory);
}
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }

inline hipError_t hipStreamCreate(hipStream_t* st) { *st = new emul_stream{hip_emul::current()}; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t st) { delete st; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t st) { return hip_emul::set(hip_emul::check(st, nullptr, 0, "hipStreamSynchronize")); }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emul_event{hip_emul::current()}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t st) { return hip_emul::set(hip_emul::check(st, nullptr, 0, "hipEventRecord")); }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }

inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  hipError_t e = hipSuccess;
  if (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice) e = hip_emul::check(st, dst, bytes, "hipMemcpy (destination)");
  if (e == hipSuccess && (kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice)) e = hip_emul::check(st, src, bytes, "hipMemcpy (source)");
  if (e != hipSuccess) return hip_emul::set(e);
  if (bytes) std::memmove(dst, src, bytes);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) { return hipMemcpyAsync(dst, src, bytes, kind, nullptr); }
inline hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind) {
  for (size_t r = 0; r < height; r++) {
    const hipError_t e = hipMemcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, width, kind);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t st) {
  const hipError_t e = hip_emul::check(st, dst, bytes, "hipMemset");
  if (e != hipSuccess) return hip_emul::set(e);
  std::memset(dst, v, bytes);
  return hipSuccess;
}
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
template <class K>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 1; return hipSuccess; }

// hipLaunchKernelGGL: the kernel runs here and now, one lane after the other.  (Kernels compiled for the host
// are plain functions; a lane that loops until its wave's work is done -- the persistent trace kernels -- simply
// does all of it.)
template <class K, class... Args>
inline void hip_emul_launch(K kernel, dim3 grid, dim3 block, hipStream_t st, Args... args) {
  const hipError_t e = hip_emul::check(st, nullptr, 0, "kernel launch");
  if (e != hipSuccess) { hip_emul::set(e); return; }
  hip_emul::state().launches++;
  gridDim.x = grid.x; blockDim.x = block.x;
  for (unsigned b = 0; b < grid.x; b++)
    for (unsigned t = 0; t < block.x; t++) {
      blockIdx.x = b; threadIdx.x = t;
      kernel(args...);
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hip_emul_launch(kernel, grid, block, stream, __VA_ARGS__)
