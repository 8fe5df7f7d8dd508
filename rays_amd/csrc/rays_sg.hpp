// placeholder replaced below
#pragma once
#include "rays_trace.hpp"
namespace rays {
template <int EQ, int NS, int DERIV, int NV, int K>
__global__ void __launch_bounds__(256) sg_trace_kernel(const DevParams P, const TraceArgs A) {}
}
