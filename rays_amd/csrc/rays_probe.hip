// rays_probe.hip -- diagnostic kernel: evaluates the RHS pieces at given states (tests only use it
// through rays_hip_probe to compare device functions with the oracle value by value).
#include "rays_device.hpp"

namespace rays {

template <int EQ, int NS, int NV>
__device__ void probe_one(const DevParams& P, const double* vin, double* cold7, double* num7,
                          double* dvds, double* resid, int* codes) {
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = vin[i];
  const double rvec[3] = {v[0], v[1], v[2]}, kvec[3] = {v[3], v[4], v[5]};
  EqPoint<NS> eq;
  equilibrium<EQ, NS>(P, const_recip(P.omgrf, P.inv_omgrf), const_recip(P.omgrf2, P.inv_omgrf2), rvec, eq, true);
  codes[0] = eq.err;
  const double nvec[3] = {kvec[0] / P.k0, kvec[1] / P.k0, kvec[2] / P.k0};
  double dx[3], dk[3], dw;
  deriv_cold<NS>(P, eq, nvec, dx, dk, dw);
  for (int i = 0; i < 3; i++) {
    cold7[i] = dx[i];
    cold7[3 + i] = dk[i];
  }
  cold7[6] = dw;
  deriv_num<EQ, NS>(P, eq, rvec, kvec, dx, dk, dw);
  for (int i = 0; i < 3; i++) {
    num7[i] = dx[i];
    num7[3 + i] = dk[i];
  }
  num7[6] = dw;
  double f[NV], r;
  int cs_flag, code;
  bool cs_stop;
  rhs_eval<EQ, NS, RAYS_DERIV_COLD, NV>(P, v, true, r, cs_flag, cs_stop, code, f);
#pragma unroll
  for (int i = 0; i < NV; i++) dvds[i] = f[i];
  *resid = r;
  codes[1] = code;
  codes[2] = cs_flag;
  codes[3] = cs_stop ? 1 : 0;
}

__global__ void probe_kernel(const DevParams P, int eq, int ns, int nv, int n, const double* v,
                             double* cold7, double* num7, double* dvds, double* resid, int* codes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || nv != 7) return;
  const double* vi = v + (size_t)i * 7;
#define RAYS_PROBE(E, N) \
  if (eq == E && ns == N) probe_one<E, N, 7>(P, vi, cold7 + 7 * (size_t)i, num7 + 7 * (size_t)i, dvds + 7 * (size_t)i, resid + i, codes + 4 * (size_t)i);
  RAYS_PROBE(0, 2)
  RAYS_PROBE(0, 3)
  RAYS_PROBE(1, 2)
  RAYS_PROBE(1, 3)
  RAYS_PROBE(2, 2)
  RAYS_PROBE(2, 3)
#undef RAYS_PROBE
}

}  // namespace rays
