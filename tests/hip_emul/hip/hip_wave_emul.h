// TEST INFRASTRUCTURE ONLY: a wave of 64 lanes on the host, for kernels whose lanes talk to each other (the
// lane-group Shampine-Gordon kernel, rays_sg_group.hpp: one ray per group of G lanes, __shfl between them).
// Included by hip_runtime.h when RAYS_EMUL_WAVE is defined.  Never part of librays_hip.so.
//
// Every lane is a fiber (own stack, a few lines of assembly to switch) running the kernel function with its own threadIdx; a cross-lane operation
// (__any, __ballot, __shfl) is a rendezvous: a lane deposits its operand and yields until every lane of the wave
// that has not finished the kernel has arrived, then all read the operands.  This models the operations as the
// hardware executes them when ALL lanes of the wave take part, which is how the kernels under test use them
// (cross-lane operations sit in wave-uniform control flow); a lane that skips a rendezvous its wave-mates wait at
// is reported as an error instead of dead-locking.  Lanes share the host thread's `rays::lds`, like a wave its LDS.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#if !defined(__x86_64__)
#error "the host wave emulator switches fibers with a few lines of x86-64 assembly"
#endif
// Fiber switch: callee-saved registers + stack pointer (glibc's swapcontext also saves the signal mask -- a system
// call per switch, and a rendezvous is 64 switches).
extern "C" void rays_wave_emul_switch(void** save_sp, void* const* load_sp);
asm(R"(
.text
.weak rays_wave_emul_switch
.type rays_wave_emul_switch,@function
rays_wave_emul_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size rays_wave_emul_switch,.-rays_wave_emul_switch
)");

namespace wave_emul {
constexpr int kLanes = 64;
constexpr size_t kStackBytes = 1u << 20;

struct Wave {
  void* sched = nullptr;   // saved stack pointers
  void* ctx[kLanes];
  std::vector<char> stacks;
  bool done[kLanes], waiting[kLanes];
  unsigned wait_epoch[kLanes];
  int ndone = 0, cur = 0;
  unsigned base_thread = 0;
  unsigned long long in[kLanes], out[kLanes];
  int arrived = 0;
  unsigned epoch = 0;
  std::function<void()> body;
};
inline Wave*& current() { static thread_local Wave* w = nullptr; return w; }

inline void complete(Wave& w) {  // every live lane has deposited its operand
  for (int i = 0; i < kLanes; i++) w.out[i] = w.done[i] ? 0ull : w.in[i];
  w.arrived = 0;
  w.epoch++;
}
// deposit `v`, wait for the wave, return everybody's operands (valid until this lane's next rendezvous)
inline const unsigned long long* exchange(unsigned long long v) {
  Wave& w = *current();
  const int me = w.cur;
  w.in[me] = v;
  w.arrived++;
  if (w.arrived == kLanes - w.ndone) {
    complete(w);
  } else {
    w.waiting[me] = true;
    w.wait_epoch[me] = w.epoch;
    rays_wave_emul_switch(&w.ctx[me], &w.sched);  // resumed by the scheduler once the epoch has moved on
  }
  return w.out;
}
inline void lane_entry() {
  Wave& w = *current();
  w.body();
  w.done[w.cur] = true;
  w.ndone++;
  rays_wave_emul_switch(&w.ctx[w.cur], &w.sched);  // never resumed
  std::abort();
}
// run one wave: lanes base_thread .. base_thread + 63 of the current block
inline void run_wave(unsigned base_thread, std::function<void()> body, emul_dim3& thread_idx) {
  Wave w;
  w.stacks.resize(kStackBytes * kLanes);
  w.base_thread = base_thread;
  w.body = std::move(body);
  Wave* outer = current();
  current() = &w;
  for (int i = 0; i < kLanes; i++) {
    w.done[i] = w.waiting[i] = false;
    // a fresh fiber: six zeroed callee-saved registers, then lane_entry as the address `ret` jumps to, on a stack
    // that is 16-byte aligned at that point (so lane_entry starts as if it had been called)
    char* top = w.stacks.data() + kStackBytes * (size_t)(i + 1);
    top = (char*)((unsigned long long)top & ~15ull) - 64;
    void** sp = (void**)top;
    sp[0] = (void*)lane_entry;
    sp[1] = nullptr;
    for (int r = 1; r <= 6; r++) sp[-r] = nullptr;
    w.ctx[i] = (void*)(sp - 6);
  }
  int idle_passes = 0;
  while (w.ndone < kLanes) {
    bool ran = false;
    for (int i = 0; i < kLanes; i++) {
      if (w.done[i]) continue;
      if (w.waiting[i]) {
        if (w.wait_epoch[i] == w.epoch) continue;  // its rendezvous is not complete yet
        w.waiting[i] = false;
      }
      w.cur = i;
      thread_idx.x = base_thread + (unsigned)i;
      rays_wave_emul_switch(&w.sched, &w.ctx[i]);
      ran = true;
      // a lane that finished while others wait may have been the one they were waiting for
      if (w.arrived > 0 && w.arrived == kLanes - w.ndone) complete(w);
    }
    if (!ran && ++idle_passes > 1) {
      std::fprintf(stderr, "[wave_emul] dead-lock: %d lanes wait at a cross-lane operation that %d live lanes never reach "
                           "(a cross-lane operation in divergent control flow)\n", w.arrived, kLanes - w.ndone - w.arrived);
      std::abort();
    }
    if (ran) idle_passes = 0;
  }
  current() = outer;
}
inline int lane_id() { return current()->cur; }
inline unsigned long long bits(double x) { unsigned long long u; std::memcpy(&u, &x, 8); return u; }
inline double from_bits(unsigned long long u) { double x; std::memcpy(&x, &u, 8); return x; }
}  // namespace wave_emul

inline int __any(int x) {
  const unsigned long long* o = wave_emul::exchange(x ? 1ull : 0ull);
  for (int i = 0; i < wave_emul::kLanes; i++) if (o[i]) return 1;
  return 0;
}
inline unsigned long long __ballot(int x) {
  const unsigned long long* o = wave_emul::exchange(x ? 1ull : 0ull);
  unsigned long long m = 0;
  for (int i = 0; i < wave_emul::kLanes; i++) if (o[i]) m |= 1ull << i;
  return m;
}
// value of lane `src` of this lane's `width`-lane segment (width a power of two)
inline double __shfl(double x, int src, int width = 64) {
  const int me = wave_emul::lane_id();
  const unsigned long long* o = wave_emul::exchange(wave_emul::bits(x));
  return wave_emul::from_bits(o[(me & ~(width - 1)) | (src & (width - 1))]);
}
inline int __shfl(int x, int src, int width = 64) {
  const int me = wave_emul::lane_id();
  const unsigned long long* o = wave_emul::exchange((unsigned long long)(unsigned)x);
  return (int)(unsigned)o[(me & ~(width - 1)) | (src & (width - 1))];
}
