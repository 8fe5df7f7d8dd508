// TEST INFRASTRUCTURE ONLY: the C ABI of librays_hip.so (rays_amd/csrc/rays_capi.hip, rays_gather.inc, rays_pack.hip)
// compiled for the host against the emulated HIP runtime (hip/hip_runtime_api_emul.h: several devices, streams and
// memory that belong to a device), so that its host logic -- device lists, cache slots, the blocks of
// rays_hip_trace, the multi-device gather of rays_hip_trace_gather over (a stand-in for) RCCL -- runs in the CPU
// tier and under ASan + UBSan.  Linked with a few kernel groups of rays_inst.hip compiled the same way
// (tests/hip_emul/Makefile.capi); every other group reports "no kernel specialisation".
#define RAYS_EMUL_RUNTIME 1
#include <hip/hip_runtime.h>
RAYS_EMUL_DEFINE_GLOBALS
#include "../../rays_amd/csrc/rays_capi.hip"
#include "../../rays_amd/csrc/rays_pack.hip"

namespace rays {
// not part of the emulated build: the launcher, deposition and probe kernels (covered lane by lane by emul_trace.cpp)
hipError_t launch_deposition(const DevParams&, const DepArgs&, const double*, double*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_ray_init(int, int, const DevParams&, const FanArgs&, int, double*, int*, int*, int*, int*, double*, double*,
                           hipStream_t) { return hipErrorNotSupported; }
int ray_init_block() { return 256; }
void probe_kernel(const DevParams, int, int, int, int, const double*, double*, double*, double*, double*, int*) {}

// kernel groups without an emulated build
#define RAYS_NO_GROUP(name) const KernelEntry* name(int* n) { *n = 0; return nullptr; }
#include "emul_capi_groups.inc"
}  // namespace rays

// test hooks of the emulated runtime
extern "C" void rays_emul_runtime_stats(long long* launches, long long* wrong_device, long long* live_allocations) {
  hip_emul::State& s = hip_emul::state();
  std::lock_guard<std::mutex> lk(s.mu);
  *launches = s.launches; *wrong_device = s.wrong_device; *live_allocations = (long long)s.allocs.size();
}
