// rays_capi.hip -- the C ABI of librays_hip.so (include/rays_hip.h).
//
// Host side of the drop-in boundary for `call trace_rays` (RAYS_project/RAYS_lib/ray_tracing.f90).
// No oracle, no CPU fallback: every entry point either runs the HIP kernels or fails with an
// error message.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rays_hip.h"
#include "rays_launch.hpp"
#include "rays_ray_init.hpp"
#include "rays_deposition.hpp"

namespace rays {
#define RAYS_DECL_ENTRIES(s, e, d) \
  const KernelEntry* rays_entries_##s##_##e##_##d##_0_0(int* n); \
  const KernelEntry* rays_entries_##s##_##e##_##d##_1_0(int* n); \
  const KernelEntry* rays_entries_##s##_##e##_##d##_0_1(int* n); \
  const KernelEntry* rays_entries_##s##_##e##_##d##_1_1(int* n);
RAYS_DECL_ENTRIES(0, 0, 0)
RAYS_DECL_ENTRIES(0, 0, 1)
RAYS_DECL_ENTRIES(0, 1, 0)
RAYS_DECL_ENTRIES(0, 1, 1)
RAYS_DECL_ENTRIES(0, 2, 0)
RAYS_DECL_ENTRIES(0, 2, 1)
RAYS_DECL_ENTRIES(1, 0, 0)
RAYS_DECL_ENTRIES(1, 0, 1)
RAYS_DECL_ENTRIES(1, 1, 0)
RAYS_DECL_ENTRIES(1, 1, 1)
RAYS_DECL_ENTRIES(1, 2, 0)
RAYS_DECL_ENTRIES(1, 2, 1)
#define RAYS_DECL_TOL(e) \
  const KernelEntry* rays_entries_tol_0_##e##_0_0_0(int* n); \
  const KernelEntry* rays_entries_tol_0_##e##_0_1_0(int* n);
RAYS_DECL_TOL(0)
RAYS_DECL_TOL(1)
RAYS_DECL_TOL(2)
hipError_t launch_pack(bool pack, int nray, int nv, int nstep_max, const int32_t* npoints,
                       const long long* offsets, double* ray_vec, double* residual, double* packed_vec,
                       double* packed_res, hipStream_t stream);
hipError_t launch_deposition(const DevParams& P, const DepArgs& D, const double* carry, double* profile,
                             hipStream_t s);
struct FanArgs;
hipError_t launch_ray_init(int eq_model, int ns, const DevParams& P, const FanArgs& F, int n_cand, double* cand,
                           int* keep, int* block_count, int* offs, int* first_of_launch, double* rvec0,
                           double* rindex_vec0, hipStream_t s);
int ray_init_block();
__global__ void probe_kernel(const DevParams P, int eq, int ns, int nv, int n, const double* v,
                             double* cold7, double* num7, double* dvds, double* resid, int* codes);
}  // namespace rays

namespace {

thread_local std::string g_err;
std::mutex g_mu;
std::vector<int> g_devices;  // devices used by rays_hip_trace
bool g_devices_explicit = false;  // the list came from rays_hip_init_devices (slots as given, never multiplied)

int fail(const std::string& msg) {
  g_err = msg;
  return 1;
}
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return 2;
}
#define HIP_TRY(call)                              \
  do {                                             \
    hipError_t e_ = (call);                        \
    if (e_ != hipSuccess) return hip_fail(e_, #call); \
  } while (0)

struct FlagText {
  int code;
  const char* text;
};
const FlagText kFlags[] = {
    {RAYS_STOP_NONE, ""},
    {RAYS_STOP_SOUT_GT_SMAX, "sout > s_max"},
    {RAYS_STOP_NSTEP_MAX, " nstep > nstep_max"},
    {RAYS_STOP_X_OUT_OF_BOUNDS, "x out_of_bounds"},
    {RAYS_STOP_Y_OUT_OF_BOUNDS, "y out_of_bounds"},
    {RAYS_STOP_Z_OUT_OF_BOUNDS, "z out_of_bounds"},
    {RAYS_STOP_NEGATIVE_DENS, "negative_dens"},
    {RAYS_STOP_NEGATIVE_TEMP, "negative_temp"},
    {RAYS_STOP_R_OUT_OF_BOX, "R out_of_box"},
    {RAYS_STOP_Z_OUT_OF_BOX, "z out_of_box"},
    {RAYS_STOP_AXI_R_OUT_OF_BOX, "R_out_of_box"},
    {RAYS_STOP_AXI_Z_OUT_OF_BOX, "Z_out_of_box"},
    {RAYS_STOP_OUT_OF_PLASMA, "out_of_plasma"},
    {RAYS_STOP_SOLMAG_R_OUT_OF_BOUNDS, "R out_of_bounds"},
    {RAYS_STOP_SOLMAG_Z_OUT_OF_BOUNDS, "z out_of_bounds"},
    {RAYS_STOP_INFINITE_VG_RHS, "infinite Vg"},
    {RAYS_STOP_RAY_STALLED, "ray stalled"},
    {RAYS_STOP_DISP_RESIDUAL, "dispersion_residual"},
    {RAYS_STOP_INFINITE_VG_CHECK, "infinite_Vg"},
    {RAYS_STOP_TOTAL_ABSORPTION, "total_absorption"},
    {RAYS_STOP_ODE_TOTAL_ERROR, "ODE total error"},
    {RAYS_STOP_SG_MAXNUM, "step number .ge. maxnum"},
    {RAYS_STOP_SG_STIFF, "equations stiff"},
    {RAYS_STOP_SG_T_EQ_TOUT, "t == tout"},
    {RAYS_STOP_SG_NEG_ERR, "relerr or abserr < 0"},
    {RAYS_STOP_SG_EPS_LE_0, "eps <= 0"},
};

static_assert(rays::kBlock == rays::PointWindow<7>::kStride, "PointWindow rows are laid out for the launch block size");
#include "rays_dev_params.inc"

// rays_hip_set_numerics: RAYS_NUMERICS_* (include/rays_hip.h)
int initial_numerics() {
  const char* e = std::getenv("RAYS_HIP_NUMERICS");
  return e && (e[0] == 't' || e[0] == 'T' || e[0] == '1') ? RAYS_NUMERICS_TOLERANCE : RAYS_NUMERICS_EXACT;
}
std::atomic<int> g_numerics{initial_numerics()};

// nray: fan size (0 = unknown).  From two waves per SIMD worth of rays on, the two-waves-per-SIMD build
// of the kernel is preferred where one exists (rays_rk4.hpp).
const rays::KernelEntry* find_kernel(const rays_params_t& p, long long nray = 0, bool force_exact = false) {
  using namespace rays;
  typedef const KernelEntry* (*Getter)(int*);
  // [solver][equilibrium][derivative][unit exponents][multi_spec_damping]
#define RAYS_G(s, e, d) {{rays_entries_##s##_##e##_##d##_0_0, rays_entries_##s##_##e##_##d##_0_1}, \
                         {rays_entries_##s##_##e##_##d##_1_0, rays_entries_##s##_##e##_##d##_1_1}}
  static const Getter getters[2][3][2][2][2] = {
      {{RAYS_G(0, 0, 0), RAYS_G(0, 0, 1)}, {RAYS_G(0, 1, 0), RAYS_G(0, 1, 1)}, {RAYS_G(0, 2, 0), RAYS_G(0, 2, 1)}},
      {{RAYS_G(1, 0, 0), RAYS_G(1, 0, 1)}, {RAYS_G(1, 1, 0), RAYS_G(1, 1, 1)}, {RAYS_G(1, 2, 0), RAYS_G(1, 2, 1)}}};
#undef RAYS_G
  // tolerance flavour of the cold RK4 groups [equilibrium][unit exponents]
  static const Getter tol_getters[3][2] = {{rays_entries_tol_0_0_0_0_0, rays_entries_tol_0_0_0_1_0},
                                           {rays_entries_tol_0_1_0_0_0, rays_entries_tol_0_1_0_1_0},
                                           {rays_entries_tol_0_2_0_0_0, rays_entries_tol_0_2_0_1_0}};
  int n = 0;
  const bool tol = !force_exact && g_numerics.load() == RAYS_NUMERICS_TOLERANCE && p.ode_solver == RAYS_ODE_RK4 &&
                   p.ray_deriv == RAYS_DERIV_COLD && !p.multi_spec_damping;
  const KernelEntry* e = tol ? tol_getters[p.equilib_model][unit_exponents(p) ? 1 : 0](&n)
                             : getters[p.ode_solver][p.equilib_model][p.ray_deriv][unit_exponents(p) ? 1 : 0]
                                      [p.multi_spec_damping ? 1 : 0](&n);
  int ncu = 256;
  {
    int dev = 0;
    if (nray > 0 && hipGetDevice(&dev) == hipSuccess) ncu = device_cu_count(dev);
  }
  bool big = nray >= 2ll * ncu * 256;  // >= two waves per SIMD
  if (const char* f = std::getenv("RAYS_HIP_FORCE_WAVES_PER_SIMD"))  // developer measurement: "1" | "2"
    big = f[0] == '2';
  // SG with finite-difference dD (nv = 7) exists in two mappings: one ray per group of four lanes
  // (rays_sg_group.hpp, the default: 212 against 300 ms per pass on the 64k-ray Solovev fan) and one ray per lane
  // (rays_sg.hpp; RAYS_HIP_SG_GROUP=0, for A/B measurements).  Both are bit-identical to the reference.
  bool group = true;
  if (const char* f = std::getenv("RAYS_HIP_SG_GROUP")) group = f[0] != '0';
  const KernelEntry* found = nullptr;
  for (int i = 0; i < n; i++)
    if (e[i].ns == p.nspec + 1 && e[i].nv == p.nv) {
      if (e[i].lanes_per_ray > 1) {
        if (group) return &e[i];
        continue;
      }
      if (e[i].occ == 1 && !found) found = &e[i];
      if (e[i].occ == 2 && big) return &e[i];
    }
  return found;
}

// Z-function spline table (host copy + lazily uploaded per-device copies)
struct ZfunTable {
  std::vector<double> host;  // [nx][4]
  int nx = 0;
  double xmin = 0., xmax = 0.;
  unsigned long long version = 0;
};
ZfunTable g_zfun;
struct ZfunDevice {
  double* ptr = nullptr;
  unsigned long long version = 0;
};
std::vector<ZfunDevice> g_zfun_dev;

int get_zfun_device(const double** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_zfun.nx <= 0) return fail("damping_model = 'damp_fund_ECH' needs rays_hip_set_zfun_table() first");
  if ((int)g_zfun_dev.size() <= dev) g_zfun_dev.resize(dev + 1);
  ZfunDevice& z = g_zfun_dev[dev];
  if (z.version != g_zfun.version) {
    if (z.ptr) (void)hipFree(z.ptr);
    z.ptr = nullptr;
    HIP_TRY(hipMalloc(&z.ptr, sizeof(double) * g_zfun.host.size()));
    HIP_TRY(hipMemcpy(z.ptr, g_zfun.host.data(), sizeof(double) * g_zfun.host.size(), hipMemcpyHostToDevice));
    z.version = g_zfun.version;
  }
  *out = z.ptr;
  return 0;
}

// axisym_toroid spline tables: one packed host copy, lazily uploaded per device
struct AxisymHost {
  std::vector<double> blob;  // all arrays back to back
  size_t off[11] = {0};      // r_grid z_grid psi rb_grid rb_fspl ne_grid ne_fspl te_grid te_fspl ti_grid ti_fspl
  int nr = 0, nz = 0, n_rb = 0, n_ne = 0, n_te = 0, n_ti = 0;
  bool lin = false;          // tables of 'eqdsk_magnetics_lin_interp': psi = Psi(nr, nz) raw, rb_fspl = T(nr), rb_grid empty
  double dR = 0., dZ = 0.;
  unsigned long long version = 0;
};
AxisymHost g_axi;
std::vector<ZfunDevice> g_axi_dev;

int get_axisym_device(rays::DevParams* D) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  const bool analytic = D->a_mag_model == RAYS_AXI_MAG_SOLOVEV;  // 'solovev_magnetics': no psi / RBphi tables
  if (analytic && g_axi.blob.empty()) {
    D->a_nr = D->a_nz = D->a_n_rb = D->a_n_ne = D->a_n_te = D->a_n_ti = 0;
    D->a_r_grid = D->a_z_grid = D->a_psi_fspl = D->a_rb_grid = D->a_rb_fspl = nullptr;
    D->a_ne_grid = D->a_ne_fspl = D->a_te_grid = D->a_te_fspl = D->a_ti_grid = D->a_ti_fspl = nullptr;
    D->a_tab1d_doubles = 0;
    D->a_lds_tab = D->a_lds_rz = 0;
    set_spline_axes(*D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    return 0;
  }
  const bool lin = D->a_mag_model == RAYS_AXI_MAG_EQDSK_LIN;
  if (lin && !g_axi.lin)
    return fail("magnetics_model = 'eqdsk_magnetics_lin_interp' needs rays_hip_set_eqdsk_lin_tables() first");
  if (!analytic && !lin && (g_axi.lin || g_axi.nr <= 1 || g_axi.nz <= 1 || g_axi.n_rb <= 1))
    return fail("equilib_model = 'axisym_toroid' needs rays_hip_set_axisym_tables() first");
  if ((int)g_axi_dev.size() <= dev) g_axi_dev.resize(dev + 1);
  ZfunDevice& z = g_axi_dev[dev];
  if (z.version != g_axi.version) {
    if (z.ptr) (void)hipFree(z.ptr);
    z.ptr = nullptr;
    HIP_TRY(hipMalloc(&z.ptr, sizeof(double) * g_axi.blob.size()));
    HIP_TRY(hipMemcpy(z.ptr, g_axi.blob.data(), sizeof(double) * g_axi.blob.size(), hipMemcpyHostToDevice));
    z.version = g_axi.version;
  }
  const double* b = z.ptr;
  D->a_nr = g_axi.nr; D->a_nz = g_axi.nz; D->a_n_rb = g_axi.n_rb;
  D->a_n_ne = g_axi.n_ne; D->a_n_te = g_axi.n_te; D->a_n_ti = g_axi.n_ti;
  D->a_r_grid = b + g_axi.off[0]; D->a_z_grid = b + g_axi.off[1]; D->a_psi_fspl = b + g_axi.off[2];
  D->a_rb_grid = b + g_axi.off[3]; D->a_rb_fspl = b + g_axi.off[4];
  D->a_ne_grid = b + g_axi.off[5]; D->a_ne_fspl = b + g_axi.off[6];
  D->a_te_grid = b + g_axi.off[7]; D->a_te_fspl = b + g_axi.off[8];
  D->a_ti_grid = b + g_axi.off[9]; D->a_ti_fspl = b + g_axi.off[10];
  D->a_tab1d_doubles = (int)(g_axi.blob.size() - g_axi.off[3]);  // rb .. ti: contiguous at the end of the blob
  for (int k = 0; k < 8; k++) D->a_tab_off[k] = (int)(g_axi.off[3 + k] - g_axi.off[3]);
  D->a_lin_dR = g_axi.dR;
  D->a_lin_dZ = g_axi.dZ;
  D->a_lds_tab = 0;
  D->a_lds_rz = 0;
  {
    const double* h = g_axi.blob.data();  // the host image of the same blob
    const auto at = [&](int k, int n) { return n > 1 ? h + g_axi.off[k] : nullptr; };
    set_spline_axes(*D, at(0, g_axi.nr), at(1, g_axi.nz), at(3, g_axi.n_rb), at(5, g_axi.n_ne), at(7, g_axi.n_te),
                    at(9, g_axi.n_ti));
  }
  return 0;
}

// Per-device ring of refill counters.  A slot is handed to one launch at a time: the launch records
// the slot's event behind its kernel, and a later launch that comes round to the same slot waits for
// that event first (more than kCounterSlots launches in flight on one device would otherwise share a
// counter and silently skip rays).
constexpr int kCounterSlots = 256;
constexpr int kCounterStride = 32;  // 128 B apart
struct DeviceWorkspace {
  unsigned int* counters = nullptr;
  hipEvent_t done[kCounterSlots] = {};
  bool used[kCounterSlots] = {};
  int next = 0;
};
DeviceWorkspace g_ws[16];

int get_counter(unsigned int** out, int* slot_out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return fail("rays_hip: device ordinal >= 16");
  hipEvent_t wait_for = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceWorkspace& w = g_ws[dev];
    if (!w.counters) HIP_TRY(hipMalloc(&w.counters, sizeof(unsigned int) * kCounterSlots * kCounterStride));
    const int slot = w.next;
    w.next = (w.next + 1) % kCounterSlots;
    if (!w.done[slot]) HIP_TRY(hipEventCreateWithFlags(&w.done[slot], hipEventDisableTiming));
    if (w.used[slot]) wait_for = w.done[slot];
    w.used[slot] = true;
    *out = w.counters + (size_t)slot * kCounterStride;
    *slot_out = slot;
  }
  if (wait_for && hipEventQuery(wait_for) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(hipEventSynchronize(wait_for));  // the launch that last used this slot is still running
  }
  return 0;
}
// Workspace of the SG kernels' upper storage tiers (TraceArgs::sg_far), one per (device, stream): launches on a
// stream run one after another, so they may share it; grown on demand, released by rays_hip_finalize.
struct SgWorkspace {
  double* ptr = nullptr;
  size_t bytes = 0;
};
std::map<std::pair<int, hipStream_t>, SgWorkspace> g_sg_ws;
int get_sg_workspace(hipStream_t stream, size_t bytes, double** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  SgWorkspace& w = g_sg_ws[std::make_pair(dev, stream)];
  if (w.bytes < bytes) {
    if (w.ptr) {
      HIP_TRY(hipStreamSynchronize(stream));  // an earlier launch on this stream may still use the old block
      (void)hipFree(w.ptr);
    }
    w.ptr = nullptr;
    w.bytes = 0;
    HIP_TRY(hipMalloc(&w.ptr, bytes));
    w.bytes = bytes;
  }
  *out = w.ptr;
  return 0;
}

// State of the "long rays first" hand-out order of the RK4 kernels (TraceArgs::sched; rays_trace.hpp: take_rays), one
// block per (device, stream) like the SG workspace.
std::map<std::pair<int, hipStream_t>, SgWorkspace> g_sched_ws;
int get_sched_workspace(hipStream_t stream, size_t bytes, unsigned int** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  SgWorkspace& w = g_sched_ws[std::make_pair(dev, stream)];
  if (w.bytes < bytes) {
    if (w.ptr) {
      HIP_TRY(hipStreamSynchronize(stream));
      (void)hipFree(w.ptr);
    }
    w.ptr = nullptr;
    w.bytes = 0;
    HIP_TRY(hipMalloc(&w.ptr, bytes));
    w.bytes = bytes;
  }
  *out = reinterpret_cast<unsigned int*>(w.ptr);
  return 0;
}
// Neighbourhood size of that order: every second row of 64 rays is traced first (cfg 5b: 4.05 | 3.26 | 3.34 | 3.42 ms for
// index order | 2 | 4 | 8; the model of tools/refill_model.py agrees: the more pilots, the better the later half is
// ordered).  RAYS_HIP_RAY_ORDER=index hands the rays out in index order instead (for A/B measurements; the results
// are the same), RAYS_HIP_RAY_ORDER=pilot2 ... pilot8 sets other neighbourhood sizes (developer switch).
constexpr int kSchedStride = 2;
int sched_stride() {
  const char* f = std::getenv("RAYS_HIP_RAY_ORDER");
  if (!f || !f[0]) return kSchedStride;
  if (f[0] == 'i') return 0;
  const int n = (int)std::strlen(f);
  const int s = f[n - 1] - '0';
  return (s >= 2 && s <= 8) ? s : kSchedStride;
}

int counter_launched(int slot, hipStream_t stream) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  HIP_TRY(hipEventRecord(g_ws[dev].done[slot], stream));
  return 0;
}

}  // namespace

extern "C" {

int rays_hip_last_error(char* buf, int len) {
  if (buf && len > 0) {
    std::snprintf(buf, (size_t)len, "%s", g_err.c_str());
  }
  return (int)g_err.size();
}

const char* rays_hip_stop_flag_text(int stop_code) {
  for (const FlagText& f : kFlags)
    if (f.code == stop_code) return f.text;
  return "";
}

int rays_hip_set_zfun_table(const double* fspl_re, int nx, double x_min, double x_max) {
  if (!fspl_re || nx < 2 || !(x_max > x_min)) return fail("rays_hip_set_zfun_table: bad table");
  std::lock_guard<std::mutex> lk(g_mu);
  g_zfun.host.assign(fspl_re, fspl_re + 4 * (size_t)nx);
  g_zfun.nx = nx;
  g_zfun.xmin = x_min;
  g_zfun.xmax = x_max;
  g_zfun.version++;
  return 0;
}

namespace {
int set_axisym_tables_impl(const rays_axisym_tables_t* t, bool lin, double dR, double dZ);
}
int rays_hip_set_axisym_tables(const rays_axisym_tables_t* t) { return set_axisym_tables_impl(t, false, 0., 0.); }
int rays_hip_set_eqdsk_lin_tables(const rays_axisym_tables_t* t, double dR, double dZ) {
  if (!t || t->nr < 2 || t->nz < 2 || t->n_rb != t->nr || !t->r_grid || !t->z_grid || !t->psi_fspl || !t->rb_fspl ||
      !(dR > 0.) || !(dZ > 0.))
    return fail("rays_hip_set_eqdsk_lin_tables: bad tables");
  return set_axisym_tables_impl(t, true, dR, dZ);
}
namespace {
int set_axisym_tables_impl(const rays_axisym_tables_t* t, bool lin, double dR, double dZ) {
  // (magnetics_model = 'solovev_magnetics' with splined profiles: nr = nz = n_rb = 0, profile tables only)
  const bool profiles_only = t && t->nr == 0 && t->nz == 0 && t->n_rb == 0 && (t->n_ne > 0 || t->n_te > 0 || t->n_ti > 0);
  if (!t || (!profiles_only && !lin && (t->nr < 2 || t->nz < 2 || t->n_rb < 2 || !t->r_grid || !t->z_grid || !t->psi_fspl ||
                                        !t->rb_grid || !t->rb_fspl)))
    return fail("rays_hip_set_axisym_tables: bad tables");
  if ((t->n_ne > 0 && (!t->ne_grid || !t->ne_fspl)) || (t->n_te > 0 && (!t->te_grid || !t->te_fspl)) ||
      (t->n_ti > 0 && (!t->ti_grid || !t->ti_fspl)))
    return fail("rays_hip_set_axisym_tables: profile table pointers missing");
  std::lock_guard<std::mutex> lk(g_mu);
  AxisymHost& h = g_axi;
  h.blob.clear();
  const double* src[11] = {t->r_grid, t->z_grid, t->psi_fspl, t->rb_grid, t->rb_fspl, t->ne_grid, t->ne_fspl,
                           t->te_grid, t->te_fspl, t->ti_grid, t->ti_fspl};
  // ('eqdsk_magnetics_lin_interp': raw Psi(nr, nz) and T(nr) in the psi / rb_fspl slots, no rb_grid)
  const size_t len[11] = {(size_t)t->nr, (size_t)t->nz, (size_t)(lin ? 1 : 16) * t->nr * t->nz, lin ? (size_t)0 : (size_t)t->n_rb,
                          (size_t)(lin ? 1 : 4) * t->n_rb, (size_t)(t->n_ne > 0 ? t->n_ne : 0), (size_t)4 * (t->n_ne > 0 ? t->n_ne : 0),
                          (size_t)(t->n_te > 0 ? t->n_te : 0), (size_t)4 * (t->n_te > 0 ? t->n_te : 0),
                          (size_t)(t->n_ti > 0 ? t->n_ti : 0), (size_t)4 * (t->n_ti > 0 ? t->n_ti : 0)};
  for (int k = 0; k < 11; k++) {
    h.off[k] = h.blob.size();
    if (len[k]) h.blob.insert(h.blob.end(), src[k], src[k] + len[k]);
    while (h.blob.size() % 16) h.blob.push_back(0.);  // keep every table 128-B aligned
  }
  h.nr = t->nr; h.nz = t->nz; h.n_rb = t->n_rb;
  h.n_ne = t->n_ne > 0 ? t->n_ne : 0; h.n_te = t->n_te > 0 ? t->n_te : 0; h.n_ti = t->n_ti > 0 ? t->n_ti : 0;
  h.lin = lin; h.dR = dR; h.dZ = dZ;
  h.version++;
  return 0;
}
}  // namespace

int rays_hip_sizeof_params(void) { return (int)sizeof(rays_params_t); }

int rays_hip_set_numerics(int mode) {
  if (mode != RAYS_NUMERICS_EXACT && mode != RAYS_NUMERICS_TOLERANCE) {
    g_err = "rays_hip_set_numerics: unknown mode";
    return -1;
  }
  return g_numerics.exchange(mode);
}
int rays_hip_get_numerics(void) { return g_numerics.load(); }

int rays_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rays_hip_init(int ngpu) {
  int n = rays_hip_device_count();
  if (n <= 0) {
    g_err = "rays_hip_init: no HIP device visible";
    return -1;
  }
  if (ngpu <= 0 || ngpu > n) ngpu = n;
  if (ngpu > 16) ngpu = 16;
  std::lock_guard<std::mutex> lk(g_mu);
  g_devices.clear();
  for (int i = 0; i < ngpu; i++) g_devices.push_back(i);
  g_devices_explicit = false;
  return ngpu;
}

int rays_hip_init_devices(int n, const int* device_ids) {
  const int visible = rays_hip_device_count();
  if (visible <= 0) {
    g_err = "rays_hip_init_devices: no HIP device visible";
    return -1;
  }
  if (n <= 0 || n > 16 || !device_ids) {
    g_err = "rays_hip_init_devices: 1..16 device slots";
    return -1;
  }
  for (int i = 0; i < n; i++)
    if (device_ids[i] < 0 || device_ids[i] >= visible) {
      g_err = "rays_hip_init_devices: device ordinal out of range";
      return -1;
    }
  std::lock_guard<std::mutex> lk(g_mu);
  g_devices.assign(device_ids, device_ids + n);
  g_devices_explicit = true;
  return n;
}

static void release_cached_device_blocks();
static void rccl_close_all();
static void release_step_scratch();
static void drop_kept_result();
int rays_hip_finalize(void) {
  rccl_close_all();
  drop_kept_result();
  std::lock_guard<std::mutex> lk(g_mu);
  for (int d = 0; d < 16; d++)
    if (g_ws[d].counters) {
      (void)hipSetDevice(d);
      (void)hipDeviceSynchronize();
      (void)hipFree(g_ws[d].counters);
      g_ws[d].counters = nullptr;
      for (int i = 0; i < kCounterSlots; i++) {
        if (g_ws[d].done[i]) (void)hipEventDestroy(g_ws[d].done[i]);
        g_ws[d].done[i] = nullptr;
        g_ws[d].used[i] = false;
      }
      g_ws[d].next = 0;
    }
  for (auto& kv : g_sg_ws)
    if (kv.second.ptr) {
      (void)hipSetDevice(kv.first.first);
      (void)hipDeviceSynchronize();
      (void)hipFree(kv.second.ptr);
    }
  g_sg_ws.clear();
  for (auto& kv : g_sched_ws)
    if (kv.second.ptr) {
      (void)hipSetDevice(kv.first.first);
      (void)hipDeviceSynchronize();
      (void)hipFree(kv.second.ptr);
    }
  g_sched_ws.clear();
  release_step_scratch();
  release_cached_device_blocks();
  g_devices.clear();
  return 0;
}

int rays_hip_check_params(const rays_params_t* p) {
  if (!p) return fail("rays_hip: null parameter block");
  if (p->abi_version != RAYS_ABI_VERSION) return fail("rays_hip: rays_params_t ABI version mismatch");
  if (p->nspec < 0 || p->nspec > RAYS_NSPEC0) return fail("rays_hip: nspec out of range 0..5");
  if (p->ode_solver != RAYS_ODE_RK4 && p->ode_solver != RAYS_ODE_SG)
    return fail("ode_solver, invalid ode solver");  // ode_m.f90:246-249
  if (p->ray_deriv != RAYS_DERIV_COLD && p->ray_deriv != RAYS_DERIV_NUM)
    return fail("EQN_RAY: invalid value, ray_deriv_name");  // eqn_ray.f90:120-122
  if (p->ray_param != RAYS_PARAM_ARCL && p->ray_param != RAYS_PARAM_TIME)
    return fail("EQN_RAY: invalid ray parameter");  // eqn_ray.f90:183-185
  if (p->equilib_model != RAYS_EQ_SLAB && p->equilib_model != RAYS_EQ_SOLOVEV && p->equilib_model != RAYS_EQ_AXISYM)
    return fail("equilibrium_m: invalid equilibrium model (device path: slab | solovev | axisym_toroid)");
  if (p->damping_model != RAYS_DAMP_NONE && p->damping_model != RAYS_DAMP_FUND_ECH)
    return fail("damping: Unimplemented damping model");  // damping_m.f90:103-106
  if (p->multi_spec_damping && !p->damping_model)  // eqn_ray.f90:196-213: the species rows sit inside the damping branch
    return fail("rays_hip: multi_spec_damping without a damping model leaves its rows of the ODE vector undefined");
  if (p->nv != 7 + (p->damping_model ? 1 : 0) + (p->multi_spec_damping ? 1 + p->nspec : 0) +
                   (p->integrate_eq_gradients ? 5 : 0))
    return fail("rays_hip: nv must be 7 (+1 with damping, +1+nspec with multi_spec_damping, +5 with "
                "integrate_eq_gradients) (ode_m.f90:160-173)");
  if (p->nstep_max < 0) return fail("rays_hip: nstep_max < 0");
  if (p->equilib_model == RAYS_EQ_SOLOVEV) {
    if (p->solovev.dens_prof_model != RAYS_SOLOVEV_N_CONSTANT && p->solovev.dens_prof_model != RAYS_SOLOVEV_N_PARABOLIC)
      return fail("solovev_eq invalid dens_prof_model");  // solovev_eq_m.f90:227-229
    for (int is = 0; is <= p->nspec; is++)
      if (p->solovev.t_prof_model[is] != RAYS_SOLOVEV_T_ZERO && p->solovev.t_prof_model[is] != RAYS_SOLOVEV_T_PARABOLIC)
        return fail("SOLOVEV: t_prof_model must be 'zero' or 'parabolic' ('constant' leaves ts undefined in the reference)");
  } else if (p->equilib_model == RAYS_EQ_AXISYM) {
    const rays_axisym_params_t& a = p->axisym;
    if (a.magnetics_model != RAYS_AXI_MAG_EQDSK_SPLINE && a.magnetics_model != RAYS_AXI_MAG_SOLOVEV &&
        a.magnetics_model != RAYS_AXI_MAG_EQDSK_LIN)
      return fail("axisym_toroid: magnetics_model must be 'eqdsk_magnetics_spline_interp', 'eqdsk_magnetics_lin_interp' "
                  "or 'solovev_magnetics'");
    if (a.magnetics_model == RAYS_AXI_MAG_SOLOVEV &&
        (p->solovev.outer_bound < p->solovev.rmaj || p->solovev.outer_bound >= std::sqrt(2.) * p->solovev.rmaj))
      return fail("Inner boundary complex, outer_bound >=  sqrt2*rmaj");  // solovev_magnetics_m.f90:99-103
    if (a.density_prof_model < 0 || a.density_prof_model > RAYS_AXI_N_SPLINE)
      return fail("axisym_toroid_eq: Unknown density_prof_model");
    for (int is = 0; is <= p->nspec; is++)
      if (a.t_prof_model[is] < 0 || a.t_prof_model[is] > RAYS_AXI_T_SPLINE)
        return fail("axisym_toroid_eq: Unknown temperature_prof_model");
    if (!(a.psiB != 0.)) return fail("axisym_toroid: psiB (PSIBOUND - PSIAXIS) is zero");
  } else {
    const rays_slab_params_t& s = p->slab;
    if (s.bx_prof_model != RAYS_SLAB_BX_ZERO) return fail("SLAB: invalid bx_prof_model");
    if (s.by_prof_model < 0 || s.by_prof_model > RAYS_SLAB_BY_LINEAR_SHEAR) return fail("SLAB: invalid by_prof_model");
    if (s.bz_prof_model < 0 || s.bz_prof_model > RAYS_SLAB_BZ_LINEAR_2) return fail("SLAB: invalid bz_prof_model");
    if (s.dens_prof_model < 0 || s.dens_prof_model > RAYS_SLAB_N_GAUSSIAN) return fail("SLAB: invalid dens_prof_model");
    for (int is = 0; is <= p->nspec; is++)
      if (s.t_prof_model[is] < 0 || s.t_prof_model[is] > RAYS_SLAB_T_PARABOLIC) return fail("SLAB: invalid t_prof_model");
  }
  if (p->ode_solver == RAYS_ODE_SG && (p->rel_err0 < (double)1.e-10f || p->abs_err0 < (double)1.e-10f))
    return fail("initialize_SG_ode: rel_err0, abs_err0 too small");  // SG_ode_m.f90:63-66
  if (!find_kernel(*p)) {
    char msg[256];
    std::snprintf(msg, sizeof msg, "rays_hip: no kernel built for this configuration (%s, equilibrium %d, %d species, %s dD, "
                  "nv = %d%s): the default library holds nspec = 1 plus the fixtures' shapes -- rebuild with "
                  "`make -C rays_amd/csrc FULL=1` for every species count", p->ode_solver == RAYS_ODE_RK4 ? "RK4" : "SG",
                  p->equilib_model, p->nspec + 1, p->ray_deriv == RAYS_DERIV_COLD ? "cold" : "numerical", p->nv,
                  p->multi_spec_damping ? ", multi_spec_damping" : "");
    return fail(msg);
  }
  return 0;
}

const char* rays_hip_kernel_name(const rays_params_t* p) {
  if (!p || rays_hip_check_params(p)) return "";
  return find_kernel(*p)->name;
}

const char* rays_hip_kernel_name_for(const rays_params_t* p, int nray) {
  if (!p || rays_hip_check_params(p)) return "";
  return find_kernel(*p, nray)->name;
}

// Common launcher of the trace kernels: `extra` carries the optional per-ray starting conditions and the
// per-run steps of a fused scan (rays_trace.hpp: TraceArgs).
namespace {
struct TraceExtras {
  const double* v0 = nullptr;
  const double* s0 = nullptr;
  const double* ds_run = nullptr;
  int rays_per_run = 0;
};
int launch_trace(const rays_params_t* p, int nray, const double* d_rvec0, const double* d_rindex_vec0,
                 double* d_ray_vec, double* d_residual, int32_t* d_npoints, int32_t* d_stop_code,
                 double* d_end_ray_vec, double* d_end_residuals, double* d_max_residuals, hipStream_t stream,
                 int flags, const TraceExtras& extra) {
  const size_t npt = (size_t)p->nstep_max + 1;
  if (!(flags & RAYS_TRACE_NO_ZERO_FILL)) {  // ray_results_m.f90:154-164
    HIP_TRY(hipMemsetAsync(d_ray_vec, 0, sizeof(double) * npt * (size_t)p->nv * (size_t)nray, stream));
    HIP_TRY(hipMemsetAsync(d_residual, 0, sizeof(double) * npt * (size_t)nray, stream));
  }
  unsigned int* counter = nullptr;
  int counter_slot = 0;
  int rc = get_counter(&counter, &counter_slot);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned int), stream));
  rays::TraceArgs A;
  A.nray = nray;
  A.rvec0 = d_rvec0;
  A.rindex_vec0 = d_rindex_vec0;
  A.ray_vec = d_ray_vec;
  A.residual = d_residual;
  A.npoints = d_npoints;
  A.stop_code = d_stop_code;
  A.end_ray_vec = d_end_ray_vec;
  A.end_residuals = d_end_residuals;
  A.max_residuals = d_max_residuals;
  A.next_ray = counter;
  A.v0 = extra.v0;
  A.s0 = extra.s0;
  A.ds_run = extra.ds_run;
  A.rays_per_run = extra.rays_per_run;
  A.sg_far = nullptr;
  A.sg_far_lanes = 0;
  A.sched = nullptr;
  A.sched_stride = 0;
  const rays::KernelEntry* kernel = find_kernel(*p, nray);
  const int stride = kernel->solver == RAYS_ODE_RK4 ? sched_stride() : 0;
  if (stride > 1) {
    // more rays than one wave per SIMD holds (the kernel decides with the lanes it is launched with)
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if ((long long)nray > (long long)rays::device_cu_count(dev) * rays::kBlock) {
      const size_t words = 4 + rays::sched_pilots((unsigned)nray, stride);
      unsigned int* ws = nullptr;
      rc = get_sched_workspace(stream, sizeof(unsigned int) * words, &ws);
      if (rc) return rc;
      HIP_TRY(hipMemsetAsync(ws, 0, sizeof(unsigned int) * words, stream));
      A.sched = ws;
      A.sched_stride = stride;
    }
  }
  if (kernel->sg_far_per_lane > 0) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    // the one-ray-per-lane SG kernels fill the CU's LDS with one workgroup, so at most one 256-lane block per CU is
    // resident; the lane-group kernels (G lanes per ray) hold up to four
    const int G = kernel->lanes_per_ray;
    const long long resident = (long long)rays::device_cu_count(dev) * rays::kBlock * (G > 1 ? 4 : 1);
    const long long rays_per_block = rays::kBlock / G;
    const long long want = ((long long)nray + rays_per_block - 1) / rays_per_block * rays::kBlock;
    A.sg_far_lanes = want < resident ? want : resident;
    double* ws = nullptr;
    rc = get_sg_workspace(stream, sizeof(double) * (size_t)kernel->sg_far_per_lane * (size_t)A.sg_far_lanes, &ws);
    if (rc) return rc;
    A.sg_far = ws;
  }
  rays::DevParams D = make_dev_params(*p);
  if (p->damping_model == RAYS_DAMP_FUND_ECH) {
    const double* zf = nullptr;
    rc = get_zfun_device(&zf);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    D.zf_fspl = zf;
    D.zf_nx = g_zfun.nx;
    D.zf_xmin = g_zfun.xmin;
    D.zf_xmax = g_zfun.xmax;
  }
  if (p->equilib_model == RAYS_EQ_AXISYM) {
    rc = get_axisym_device(&D);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    const bool need_ne = p->axisym.density_prof_model == RAYS_AXI_N_SPLINE && g_axi.n_ne < 2;
    bool need_t = false;
    for (int is = 0; is <= p->nspec; is++)
      if (p->axisym.t_prof_model[is] == RAYS_AXI_T_SPLINE && (g_axi.n_te < 2 || g_axi.n_ti < 2)) need_t = true;
    if (need_ne || need_t) return fail("axisym_toroid: spline profile model selected but its table was not set");
  }
  // A tolerance-flavour kernel hands its ill-conditioned steps over to the reference's arithmetic (rays_rk4_body.inc:
  // kStopResumeExact): the exact twin's resume kernel follows it on the stream.  The hand-over travels in the per-ray
  // summaries, so they exist for such a launch whether or not the caller asked for them.
  const rays::KernelEntry* twin = nullptr;
  if (kernel->eq & rays::kEqTol) {
    twin = find_kernel(*p, 0, true);
    if (!twin || !twin->resume) return fail("rays_hip: the tolerance kernel's exact twin is not in this build");
    if (!A.end_ray_vec || !A.max_residuals) {
      double* ws = nullptr;
      rc = get_sg_workspace(stream, sizeof(double) * ((size_t)p->nv + 1) * (size_t)nray, &ws);  // (no SG kernel runs with it at the same time: one stream)
      if (rc) return rc;
      if (!A.end_ray_vec) A.end_ray_vec = ws;
      if (!A.max_residuals) A.max_residuals = ws + (size_t)p->nv * (size_t)nray;
    }
  }
  int grid = 0;
  hipError_t e = kernel->launch(D, A, stream, &grid);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  if (twin) {
    e = twin->resume(D, A, stream);
    if (e != hipSuccess) return hip_fail(e, "resume kernel launch");
  }
  return counter_launched(counter_slot, stream);
}
}  // namespace

int rays_hip_trace_device(const rays_params_t* p, int nray, const double* d_rvec0,
                          const double* d_rindex_vec0, double* d_ray_vec, double* d_residual,
                          int32_t* d_npoints, int32_t* d_stop_code, double* d_end_ray_vec,
                          double* d_end_residuals, double* d_max_residuals, void* hip_stream,
                          int flags) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (nray < 0) return fail("rays_hip_trace_device: nray < 0");
  if (nray == 0) return 0;
  if (!d_rvec0 || !d_rindex_vec0 || !d_ray_vec || !d_residual || !d_npoints || !d_stop_code)
    return fail("rays_hip_trace_device: null device pointer");
  return launch_trace(p, nray, d_rvec0, d_rindex_vec0, d_ray_vec, d_residual, d_npoints, d_stop_code,
                      d_end_ray_vec, d_end_residuals, d_max_residuals, (hipStream_t)hip_stream, flags,
                      TraceExtras());
}

// ray_scan fused into one launch (ray_scan.f90:33-49, scanner_m.f90:174-205: scan_parameter = 'ds').
int rays_hip_scan_device(const rays_params_t* p, int n_runs, const double* d_ds_values, int nray,
                         const double* d_rvec0, const double* d_rindex_vec0, double* d_ray_vec,
                         double* d_residual, int32_t* d_npoints, int32_t* d_stop_code, double* d_end_ray_vec,
                         double* d_end_residuals, double* d_max_residuals, void* hip_stream, int flags) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (n_runs < 0 || nray < 0) return fail("rays_hip_scan_device: n_runs, nray < 0");
  if (n_runs == 0 || nray == 0) return 0;
  if ((long long)n_runs * nray > 0x7fffffffll) return fail("rays_hip_scan_device: n_runs * nray exceeds 2^31 - 1");
  if (!d_ds_values || !d_rvec0 || !d_rindex_vec0 || !d_ray_vec || !d_residual || !d_npoints || !d_stop_code)
    return fail("rays_hip_scan_device: null device pointer");
  TraceExtras x;
  x.ds_run = d_ds_values;
  x.rays_per_run = nray;
  return launch_trace(p, n_runs * nray, d_rvec0, d_rindex_vec0, d_ray_vec, d_residual, d_npoints, d_stop_code,
                      d_end_ray_vec, d_end_residuals, d_max_residuals, (hipStream_t)hip_stream, flags, x);
}

// Batched `call ode_solver(eqn_ray, nv, v, s, sout, ray_stop)` (ode_m.f90:218-254) + the check_save that
// trace_rays applies to its result (ray_tracing.f90:212-243): one output step from n arbitrary states.
// Runs the trace kernels with nstep_max = 1 from the caller's v0 / s0 and picks point 2 of each ray.
namespace rays {
__global__ void ode_step_collect_kernel(int n, int nv, const double* __restrict__ ray_vec,
                                        const double* __restrict__ residual, const int32_t* __restrict__ npoints,
                                        const int32_t* __restrict__ stop, const double* __restrict__ end_ray_vec,
                                        double* __restrict__ v1, double* __restrict__ resid, int32_t* __restrict__ code) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const bool stepped = npoints[r] == 2;  // the step was taken and passed check_save
  // not recorded: the state `ode_solver` left behind -- the advanced v when check_save refused the step, v0 when the
  // solver itself stopped (ray_tracing.f90:214-234, RK4_ode_m.f90:83-89) -- is the trace kernels' end_ray_vec
  // (zeros when v0 already failed the initial check_save: that ray never started, ray_tracing.f90:100-112)
  for (int c = 0; c < nv; c++)
    v1[(long long)r * nv + c] = stepped ? ray_vec[((long long)r * 2 + 1) * nv + c] : end_ray_vec[(long long)r * nv + c];
  if (resid) resid[r] = stepped ? residual[(long long)r * 2 + 1] : 0.;
  // a ray that took its one step ends on ' nstep > nstep_max' (or 'sout > s_max'), which is not a stop of this step
  code[r] = stepped ? RAYS_STOP_NONE : stop[r];
}
}  // namespace rays

namespace {
// scratch of rays_hip_ode_step_device, one block per (device, stream), grown on demand and kept (a host that calls
// the entry per time step would otherwise pay four hipMalloc / hipFree per call); released by rays_hip_finalize
std::map<std::pair<int, hipStream_t>, SgWorkspace> g_step_ws;
int get_step_scratch(hipStream_t stream, size_t bytes, char** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  SgWorkspace& w = g_step_ws[std::make_pair(dev, stream)];
  if (w.bytes < bytes) {
    if (w.ptr) {
      HIP_TRY(hipStreamSynchronize(stream));
      (void)hipFree(w.ptr);
    }
    w.ptr = nullptr;
    w.bytes = 0;
    HIP_TRY(hipMalloc(&w.ptr, bytes));
    w.bytes = bytes;
  }
  *out = reinterpret_cast<char*>(w.ptr);
  return 0;
}
}  // namespace

int rays_hip_ode_step_device(const rays_params_t* p, int n, const double* d_v0, const double* d_s0,
                             double* d_v1, double* d_resid, int32_t* d_stop_code, void* hip_stream) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (n < 0) return fail("rays_hip_ode_step_device: n < 0");
  if (n == 0) return 0;
  if (!d_v0 || !d_v1 || !d_stop_code) return fail("rays_hip_ode_step_device: null device pointer");
  hipStream_t stream = (hipStream_t)hip_stream;
  rays_params_t q = *p;
  q.nstep_max = 1;
  q.s_max = 1.7976931348623157e308;
  const size_t nv = (size_t)p->nv, N = (size_t)n;
  // one block: ray_vec[n][2][nv] | residual[n][2] | end_ray_vec[n][nv] | npoints[n] | stop_code[n]
  const size_t off_res = sizeof(double) * 2 * nv * N, off_ev = off_res + sizeof(double) * 2 * N,
               off_np = off_ev + sizeof(double) * nv * N, off_sc = off_np + sizeof(int32_t) * N,
               total = off_sc + sizeof(int32_t) * N;
  char* base = nullptr;
  rc = get_step_scratch(stream, total, &base);
  if (rc) return rc;
  double *d_rv = reinterpret_cast<double*>(base), *d_res = reinterpret_cast<double*>(base + off_res),
         *d_ev = reinterpret_cast<double*>(base + off_ev);
  int32_t *d_np = reinterpret_cast<int32_t*>(base + off_np), *d_sc = reinterpret_cast<int32_t*>(base + off_sc);
  TraceExtras x;
  x.v0 = d_v0;
  x.s0 = d_s0;
  // rvec0 / rindex_vec0 are not read when v0 is given; any valid pointer will do
  rc = launch_trace(&q, n, d_v0, d_v0, d_rv, d_res, d_np, d_sc, d_ev, nullptr, nullptr, stream,
                    RAYS_TRACE_NO_ZERO_FILL, x);
  if (rc) return rc;
  hipLaunchKernelGGL(rays::ode_step_collect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, (int)nv, d_rv,
                     d_res, d_np, d_sc, d_ev, d_v1, d_resid, d_stop_code);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rays_hip_ode_step_device");
  return 0;  // asynchronous on `stream` like rays_hip_trace_device (the scratch block outlives the call)
}

static void release_step_scratch() {  // caller holds g_mu
  for (auto& kv : g_step_ws)
    if (kv.second.ptr) {
      (void)hipSetDevice(kv.first.first);
      (void)hipDeviceSynchronize();
      (void)hipFree(kv.second.ptr);
    }
  g_step_ws.clear();
}

// Device buffers of rays_hip_trace are kept between calls (a host that traces repeatedly -- ray_scan, a
// time loop -- otherwise pays ~6 ms per call for hipMalloc/hipFree of the 64k fan's 5 GB): a released
// block goes to its device's free list and serves the next request of a similar size.  Everything is
// returned to the driver by rays_hip_finalize, or at once when an allocation fails.
// (One cache per SLOT of the device list rays_hip_init[_devices] selected -- a device may appear in
// several slots, each with its own host thread, stream and buffers.)
struct DeviceBlockCache {
  struct Block { void* p; size_t cap; };
  int device = -1;
  std::mutex mu;
  std::vector<Block> idle;
  std::map<void*, size_t> live;
  hipStream_t stream = nullptr;  // the entry's stream on this device (creating one costs ~8 ms per call)
  void drop_idle() {  // caller holds mu and has the device current
    for (auto& b : idle) (void)hipFree(b.p);
    idle.clear();
  }
};
static DeviceBlockCache g_blocks[17];  // slots 0..15: the device list; 16: the gathered result (rays_gather.inc)
static hipError_t cached_malloc(int slot, void** out, size_t bytes) {
  if (slot < 0 || slot >= 17) return hipMalloc(out, bytes);
  DeviceBlockCache& c = g_blocks[slot];
  std::lock_guard<std::mutex> lk(c.mu);
  size_t best = c.idle.size();
  for (size_t i = 0; i < c.idle.size(); i++)
    if (c.idle[i].cap >= bytes && c.idle[i].cap <= bytes + bytes / 4 + (1u << 20) &&
        (best == c.idle.size() || c.idle[i].cap < c.idle[best].cap))
      best = i;
  if (best < c.idle.size()) {
    *out = c.idle[best].p;
    c.live[*out] = c.idle[best].cap;
    c.idle.erase(c.idle.begin() + (long)best);
    return hipSuccess;
  }
  hipError_t e = hipMalloc(out, bytes ? bytes : 1);
  if (e != hipSuccess) {  // give the idle blocks back and try once more
    (void)hipGetLastError();
    c.drop_idle();
    e = hipMalloc(out, bytes ? bytes : 1);
  }
  if (e == hipSuccess) c.live[*out] = bytes ? bytes : 1;
  return e;
}
static hipError_t cached_stream(int slot, hipStream_t* out, bool* owned) {
  *owned = slot < 0 || slot >= 17;
  if (*owned) return hipStreamCreate(out);
  DeviceBlockCache& c = g_blocks[slot];
  std::lock_guard<std::mutex> lk(c.mu);
  if (!c.stream) {
    hipError_t e = hipStreamCreate(&c.stream);
    if (e != hipSuccess) return e;
  }
  *out = c.stream;
  return hipSuccess;
}
static void cached_free(int slot, void* ptr) {
  if (!ptr) return;
  if (slot < 0 || slot >= 17) { (void)hipFree(ptr); return; }
  DeviceBlockCache& c = g_blocks[slot];
  std::lock_guard<std::mutex> lk(c.mu);
  auto it = c.live.find(ptr);
  if (it == c.live.end()) { (void)hipFree(ptr); return; }
  c.idle.push_back({ptr, it->second});
  c.live.erase(it);
}

// A cache slot serves one device at a time (its stream and idle blocks live there).  Every user of a slot -- the
// blocks of rays_hip_trace and of rays_hip_trace_gather alike -- claims it for the device it is about to use: a slot
// that last served another device (e.g. four slots on device 0 for a large fan, then one slot per device for a
// gather) first gives that device's blocks and stream back.  Caller has `dev` current; it is current on return.
static void claim_slot_for_device(int slot, int dev) {
  if (slot < 0 || slot >= 17) return;
  DeviceBlockCache& c = g_blocks[slot];
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.device >= 0 && c.device != dev) {
    (void)hipSetDevice(c.device);
    c.drop_idle();
    if (c.stream) (void)hipStreamDestroy(c.stream);
    c.stream = nullptr;
    (void)hipSetDevice(dev);
  }
  c.device = dev;
}

static void release_cached_device_blocks() {
  for (int d = 0; d < 17; d++) {
    DeviceBlockCache& c = g_blocks[d];
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.idle.empty() && !c.stream) continue;
    if (c.device >= 0) (void)hipSetDevice(c.device);
    c.drop_idle();
    if (c.stream) (void)hipStreamDestroy(c.stream);
    c.stream = nullptr;
  }
}

// Pinned staging for the packed device-to-host copy of rays_hip_trace: two buffers per device,
// allocated once (pinning is slow) and kept.  `points` = trajectory points one buffer holds.
struct StagingBuffers {
  double* vec[2] = {nullptr, nullptr};
  double* res[2] = {nullptr, nullptr};
  long long points = 0;
  size_t nv = 0;
};
// Fixed storage: every device's host thread keeps a pointer into it for the whole copy phase, so
// the elements must never move (one slot per device ordinal, like g_blocks).
constexpr int kMaxDevices = 16;
static StagingBuffers g_staging[kMaxDevices];
static StagingBuffers* staging_for_slot(int dev, size_t nv, long long min_points) {
  if (dev < 0 || dev >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  StagingBuffers& sb = g_staging[dev];
  if (sb.points == 0 || sb.nv < nv || sb.points < min_points) {
    for (int b = 0; b < 2; b++) {
      if (sb.vec[b]) (void)hipHostFree(sb.vec[b]);
      if (sb.res[b]) (void)hipHostFree(sb.res[b]);
      sb.vec[b] = sb.res[b] = nullptr;
    }
    // 1 M points per buffer (8 (nv + 1) MB, e.g. 64 MB for nv = 7), and never less than one whole ray
    const long long pts = std::max(1ll << 20, min_points);
    for (int b = 0; b < 2; b++) {
      if (hipHostMalloc((void**)&sb.vec[b], sizeof(double) * nv * (size_t)pts, hipHostMallocPortable) != hipSuccess ||
          hipHostMalloc((void**)&sb.res[b], sizeof(double) * (size_t)pts, hipHostMallocPortable) != hipSuccess) {
        sb.points = 0;
        return nullptr;
      }
    }
    sb.points = pts;
    sb.nv = nv;
  }
  return &sb;
}

// ---- the device-resident image of the last rays_hip_trace call (rays_hip_keep_last_result) -------------------------
// The reference's drivers trace and then post-process in one process (RAYS_P.f90:19-44: trace_rays, then the
// deposition profiles of the same ray_results_m arrays).  With the switch on, a block of rays_hip_trace leaves its
// padded ray_vec slab and its npoints on the device it traced on instead of giving them back to the block cache, and
// rays_hip_deposition_last bins them in place: the trajectories cross PCIe once (to the caller's arrays), never back.
struct KeptBlock {
  int slot = -1, device = -1, r0 = 0, r1 = 0;
  double* d_ray_vec = nullptr;
  int32_t* d_npoints = nullptr;
};
struct KeptResult {
  bool keep = false;
  int nray = 0, nv = 0, nstep_max = 0;
  std::vector<KeptBlock> blocks;  // in ray order once complete
} g_kept;
static void drop_kept_result() {  // caller holds no lock
  std::vector<KeptBlock> old;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    old.swap(g_kept.blocks);
    g_kept.nray = 0;
  }
  for (const KeptBlock& b : old) {
    (void)hipSetDevice(b.device);
    cached_free(b.slot, b.d_ray_vec);
    cached_free(b.slot, b.d_npoints);
  }
}

// One device's share of rays_hip_trace: rays [r0, r1) -> contiguous slabs of the host arrays.
static int trace_block_on_device(int slot, int dev, const rays_params_t* p, int r0, int r1, const double* rvec0,
                                 const double* rindex_vec0, double* ray_vec, double* residual,
                                 int32_t* npoints, int32_t* stop_code, double* end_ray_vec,
                                 double* end_residuals, double* max_residuals, std::string* err) {
  auto bail = [&](int rc) {
    *err = g_err;
    return rc;
  };
#define DEV_TRY(call)                                      \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) return bail(hip_fail(e_, #call)); \
  } while (0)
  const int n = r1 - r0;
  if (n <= 0) return 0;
  const size_t npt = (size_t)p->nstep_max + 1, nv = (size_t)p->nv;
  const bool timing = std::getenv("RAYS_HIP_TIMING") != nullptr;  // phase times of this entry on stderr
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[rays_hip_trace dev %d] %-28s %8.2f ms\n", dev, what,
                 std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  DEV_TRY(hipSetDevice(dev));
  claim_slot_for_device(slot, dev);
  hipStream_t st;
  bool own_stream = false;
  DEV_TRY(cached_stream(slot, &st, &own_stream));
  lap("stream");
  double *d_r = nullptr, *d_n = nullptr, *d_rv = nullptr, *d_res = nullptr, *d_ev = nullptr, *d_er = nullptr,
         *d_mr = nullptr;
  int32_t *d_np = nullptr, *d_sc = nullptr;
  int rc = 0;
  do {
#define DEV_CHK(call)                        \
  {                                          \
    hipError_t e_ = (call);                  \
    if (e_ != hipSuccess) {                  \
      rc = bail(hip_fail(e_, #call));        \
      break;                                 \
    }                                        \
  }
    DEV_CHK(cached_malloc(slot, (void**)&d_r, sizeof(double) * 3 * n));
    DEV_CHK(cached_malloc(slot, (void**)&d_n, sizeof(double) * 3 * n));
    DEV_CHK(cached_malloc(slot, (void**)&d_rv, sizeof(double) * npt * nv * n));
    DEV_CHK(cached_malloc(slot, (void**)&d_res, sizeof(double) * npt * n));
    DEV_CHK(cached_malloc(slot, (void**)&d_np, sizeof(int32_t) * n));
    DEV_CHK(cached_malloc(slot, (void**)&d_sc, sizeof(int32_t) * n));
    DEV_CHK(cached_malloc(slot, (void**)&d_ev, sizeof(double) * nv * n));
    DEV_CHK(cached_malloc(slot, (void**)&d_er, sizeof(double) * n));
    DEV_CHK(cached_malloc(slot, (void**)&d_mr, sizeof(double) * n));
    lap("device allocations");
    DEV_CHK(hipMemcpyAsync(d_r, rvec0 + 3 * (size_t)r0, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
    DEV_CHK(hipMemcpyAsync(d_n, rindex_vec0 + 3 * (size_t)r0, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
    // no zero-fill of the device arrays: only recorded points are read back
    rc = rays_hip_trace_device(p, n, d_r, d_n, d_rv, d_res, d_np, d_sc, d_ev, d_er, d_mr, st, RAYS_TRACE_NO_ZERO_FILL);
    if (rc) {
      bail(rc);
      break;
    }
    // ---- trajectories: only the recorded points cross PCIe ------------------------------------
    // The padded arrays are ~80 % zeros (a ray uses npoints of nstep_max+1 slots; 4.7 GB for the 64k
    // fan, 0.82 GB of it data).  Pack on the device, copy the packed block through two pinned
    // staging buffers, and scatter it into the caller's arrays with host threads while the next
    // chunk is in flight.  Entries past npoints are not written: like the reference's trace_rays,
    // which relies on initialize_ray_results_m having zero-filled the arrays (ray_results_m.f90:
    // 154-164), this entry leaves them as the caller passed them.
    DEV_CHK(hipMemcpyAsync(npoints + r0, d_np, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    DEV_CHK(hipStreamSynchronize(st));
    lap("inputs + trace kernel");
    {
      std::vector<long long> offs((size_t)n + 1);
      offs[0] = 0;
      for (int i = 0; i < n; i++) offs[(size_t)i + 1] = offs[i] + (npoints[r0 + i] > 0 ? npoints[r0 + i] : 0);
      const long long total = offs[n];
      long long* d_off = nullptr;
      double *d_pv = nullptr, *d_pr = nullptr;
      DEV_CHK(cached_malloc(slot, (void**)&d_off, sizeof(long long) * ((size_t)n + 1)));
      bool ok = true;
      do {
        if (total == 0) break;
        if (cached_malloc(slot, (void**)&d_pv, sizeof(double) * nv * (size_t)total) != hipSuccess ||
            cached_malloc(slot, (void**)&d_pr, sizeof(double) * (size_t)total) != hipSuccess) { ok = false; break; }
        if (hipMemcpyAsync(d_off, offs.data(), sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice, st) != hipSuccess) { ok = false; break; }
        if (rays::launch_pack(true, n, (int)nv, p->nstep_max, d_np, d_off, d_rv, d_res, d_pv, d_pr, st) != hipSuccess) { ok = false; break; }
        StagingBuffers* sb = staging_for_slot(slot, nv, (long long)npt);
        if (!sb) { ok = false; break; }
        // chunks of rays whose packed size fits one staging buffer
        int c0 = 0, buf = 0;
        hipEvent_t ev[2];
        if (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) { ok = false; break; }
        struct Chunk { int a, b, buf; };
        Chunk pending{0, 0, -1};
        auto scatter = [&](const Chunk& c) {   // host side of one chunk: packed staging -> padded arrays
          const long long base = offs[c.a];
          const double* sv = sb->vec[c.buf];
          const double* sr = sb->res[c.buf];
          const int nt = 16;
          std::vector<std::thread> th;
          for (int t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
              for (int i = c.a + t; i < c.b; i += nt) {
                const long long np_i = offs[(size_t)i + 1] - offs[i];
                if (np_i <= 0) continue;
                std::memcpy(ray_vec + npt * nv * (size_t)(r0 + i), sv + (offs[i] - base) * (long long)nv,
                            sizeof(double) * nv * (size_t)np_i);
                std::memcpy(residual + npt * (size_t)(r0 + i), sr + (offs[i] - base), sizeof(double) * (size_t)np_i);
              }
            });
          for (auto& x : th) x.join();
        };
        while (c0 < n && ok) {
          int c1 = c0;
          while (c1 < n && offs[(size_t)c1 + 1] - offs[c0] <= sb->points) c1++;
          if (c1 == c0) { ok = false; break; }  // a ray has at most nstep_max+1 <= sb->points points (staging_for_slot)
          const long long pts = offs[c1] - offs[c0];
          if (pts > 0) {
            if (hipMemcpyAsync(sb->vec[buf], d_pv + offs[c0] * (long long)nv, sizeof(double) * nv * (size_t)pts,
                               hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(sb->res[buf], d_pr + offs[c0], sizeof(double) * (size_t)pts, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipEventRecord(ev[buf], st) != hipSuccess) { ok = false; break; }
          }
          if (pending.buf >= 0) scatter(pending);   // overlaps the copy just queued
          if (pts > 0) {
            if (hipEventSynchronize(ev[buf]) != hipSuccess) { ok = false; break; }
            pending = Chunk{c0, c1, buf};
            buf ^= 1;
          } else {
            pending.buf = -1;
          }
          c0 = c1;
        }
        if (ok && pending.buf >= 0) scatter(pending);
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
      } while (0);
      cached_free(slot, d_off); cached_free(slot, d_pv); cached_free(slot, d_pr);
      lap("pack + copy + host scatter");
      if (!ok) {
        rc = bail(fail("rays_hip_trace: packed device-to-host copy failed (out of memory?)"));
        break;
      }
    }
    DEV_CHK(hipMemcpyAsync(stop_code + r0, d_sc, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    if (end_ray_vec) DEV_CHK(hipMemcpyAsync(end_ray_vec + nv * (size_t)r0, d_ev, sizeof(double) * nv * n, hipMemcpyDeviceToHost, st));
    if (end_residuals) DEV_CHK(hipMemcpyAsync(end_residuals + r0, d_er, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    if (max_residuals) DEV_CHK(hipMemcpyAsync(max_residuals + r0, d_mr, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    DEV_CHK(hipStreamSynchronize(st));
  } while (0);
  lap("summaries");
  {
    bool kept = false;
    if (rc == 0) {
      std::lock_guard<std::mutex> lk(g_mu);
      if (g_kept.keep) {
        KeptBlock b;
        b.slot = slot; b.device = dev; b.r0 = r0; b.r1 = r1; b.d_ray_vec = d_rv; b.d_npoints = d_np;
        g_kept.blocks.push_back(b);
        kept = true;
      }
    }
    if (kept) d_rv = nullptr, d_np = nullptr;  // (cached_free ignores null)
  }
  cached_free(slot, d_r); cached_free(slot, d_n); cached_free(slot, d_rv); cached_free(slot, d_res); cached_free(slot, d_np);
  cached_free(slot, d_sc); cached_free(slot, d_ev); cached_free(slot, d_er); cached_free(slot, d_mr);
  if (own_stream) (void)hipStreamDestroy(st);
  lap("device frees");
  return rc;
#undef DEV_CHK
#undef DEV_TRY
}

int rays_hip_trace(const rays_params_t* p, int nray, const double* rvec0, const double* rindex_vec0,
                   double* ray_vec, double* residual, int32_t* npoints, int32_t* stop_code,
                   double* end_ray_vec, double* end_residuals, double* max_residuals,
                   double* elapsed_s) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (nray < 0) return fail("rays_hip_trace: nray < 0");
  if (nray > 0 && (!rvec0 || !rindex_vec0 || !ray_vec || !residual || !npoints || !stop_code))
    return fail("rays_hip_trace: null array argument");
  std::vector<int> devs;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    devs = g_devices;
  }
  if (devs.empty()) {
    if (rays_hip_init(0) < 0) return 3;
    std::lock_guard<std::mutex> lk(g_mu);
    devs = g_devices;
  }
  {
    // Large fans: several slots per device, so that one slot's packed device-to-host copy (what bounds this
    // entry: 0.8 GB at ~50 GB/s for the 64k fan) runs while the other slots still trace.  Measured on the 64k
    // fan: 22.0 ms with one slot, 20.9 / 18.8 / 21.0 ms with 2 / 4 / 8.  RAYS_HIP_SLOTS_PER_DEVICE overrides;
    // a list given through rays_hip_init_devices is taken as it is.
    bool explicit_list;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      explicit_list = g_devices_explicit;
    }
    int k = (long long)nray >= 32768ll * (long long)devs.size() ? 4 : 1;
    if (const char* e = std::getenv("RAYS_HIP_SLOTS_PER_DEVICE")) k = std::atoi(e);
    while (k > 1 && (size_t)k * devs.size() > 16) k--;
    if (!explicit_list && k > 1) {
      std::vector<int> slots;
      for (int d : devs)
        for (int i = 0; i < k; i++) slots.push_back(d);
      devs.swap(slots);
    }
  }
  drop_kept_result();  // the image of an earlier call (if any) goes back to the block cache
  const auto t0 = std::chrono::steady_clock::now();
  const int G = (int)devs.size();
  // contiguous blocks, like the reference's OpenMP schedule(static) (ray_tracing.f90:62)
  const int per = (nray + G - 1) / G;
  std::vector<int> rcs(G, 0);
  std::vector<std::string> errs(G);
  std::vector<std::thread> th;
  for (int g = 0; g < G; g++) {
    const int r0 = std::min(nray, g * per), r1 = std::min(nray, (g + 1) * per);
    th.emplace_back([&, g, r0, r1] {
      rcs[g] = trace_block_on_device(g, devs[g], p, r0, r1, rvec0, rindex_vec0, ray_vec, residual, npoints,
                                     stop_code, end_ray_vec, end_residuals, max_residuals, &errs[g]);
    });
  }
  for (auto& t : th) t.join();
  for (int g = 0; g < G; g++)
    if (rcs[g]) {
      g_err = errs[g];
      drop_kept_result();
      return rcs[g];
    }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_kept.keep) {
      std::sort(g_kept.blocks.begin(), g_kept.blocks.end(), [](const KeptBlock& a, const KeptBlock& b) { return a.r0 < b.r0; });
      g_kept.nray = nray;
      g_kept.nv = p->nv;
      g_kept.nstep_max = p->nstep_max;
    }
  }
  if (elapsed_s)
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

int rays_hip_keep_last_result(int on) {
  if (std::getenv("RAYS_HIP_NO_KEEP_LAST_RESULT")) on = 0;  // measurement switch: A/B against the host-array path
  bool was;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    was = g_kept.keep;
    g_kept.keep = on != 0;
  }
  if (!on) drop_kept_result();
  return was ? 1 : 0;
}

int rays_hip_pack_device(int nray, int nv, int nstep_max, const int32_t* d_npoints,
                         const int64_t* d_offsets, const double* d_ray_vec, const double* d_residual,
                         double* d_packed_vec, double* d_packed_res, void* hip_stream) {
  hipError_t e = rays::launch_pack(true, nray, nv, nstep_max, d_npoints, (const long long*)d_offsets,
                                   const_cast<double*>(d_ray_vec), const_cast<double*>(d_residual),
                                   d_packed_vec, d_packed_res, (hipStream_t)hip_stream);
  return e == hipSuccess ? 0 : hip_fail(e, "rays_hip_pack_device");
}

int rays_hip_unpack_device(int nray, int nv, int nstep_max, const int32_t* d_npoints,
                           const int64_t* d_offsets, const double* d_packed_vec,
                           const double* d_packed_res, double* d_ray_vec, double* d_residual,
                           void* hip_stream) {
  hipError_t e = rays::launch_pack(false, nray, nv, nstep_max, d_npoints, (const long long*)d_offsets,
                                   d_ray_vec, d_residual, const_cast<double*>(d_packed_vec),
                                   const_cast<double*>(d_packed_res), (hipStream_t)hip_stream);
  return e == hipSuccess ? 0 : hip_fail(e, "rays_hip_unpack_device");
}

// Diagnostic entry (tests): evaluate the RHS pieces at n states on the current device.
// v[n][nv] host; outputs host: cold7[n][7], num7[n][7], dvds[n][nv], resid[n], codes[n][4]
// (codes: equilibrium err, eqn_ray stop code, check_save flag, check_save stop_ode).
// rho(psiN) spline table: host copy + lazily uploaded per-device copies (grid[n] then fspl[n][4])
namespace {
struct RhoTable {
  std::vector<double> host;
  int n = 0;
  unsigned long long version = 0;
};
RhoTable g_rho;
std::vector<ZfunDevice> g_rho_dev;
int get_rho_device(const double** out, int* n) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rho.n < 2) return fail("Ptotal_rho needs rays_hip_set_rho_table() first");
  if ((int)g_rho_dev.size() <= dev) g_rho_dev.resize(dev + 1);
  ZfunDevice& z = g_rho_dev[dev];
  if (z.version != g_rho.version) {
    if (z.ptr) (void)hipFree(z.ptr);
    z.ptr = nullptr;
    HIP_TRY(hipMalloc(&z.ptr, sizeof(double) * g_rho.host.size()));
    HIP_TRY(hipMemcpy(z.ptr, g_rho.host.data(), sizeof(double) * g_rho.host.size(), hipMemcpyHostToDevice));
    z.version = g_rho.version;
  }
  *out = z.ptr;
  *n = g_rho.n;
  return 0;
}
}  // namespace

int rays_hip_set_rho_table(const double* grid, const double* fspl, int n) {
  if (!grid || !fspl || n < 2) return fail("rays_hip_set_rho_table: bad table");
  std::lock_guard<std::mutex> lk(g_mu);
  g_rho.host.assign(grid, grid + n);
  g_rho.host.insert(g_rho.host.end(), fspl, fspl + 4 * (size_t)n);
  g_rho.n = n;
  g_rho.version++;
  return 0;
}

int rays_hip_deposition_device(const rays_params_t* p, int which, int n_bins, int nray, const double* d_ray_vec,
                               const int32_t* d_npoints, const double* d_initial_ray_power, double* d_work,
                               const double* d_profile_in, double* d_profile_out, void* hip_stream) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (p->equilib_model != RAYS_EQ_AXISYM && p->equilib_model != RAYS_EQ_SLAB)  // deposition_profiles_m.f90:129-222
    return fail("initialize_deposition_profiles: unimplemented equilib_model");
  if (p->nv < 8 || p->damping_model == RAYS_DAMP_NONE)
    return fail("rays_hip_deposition: needs a run with damping (ray_vec(8) = absorbed power fraction)");
  if (p->equilib_model == RAYS_EQ_AXISYM && p->axisym.magnetics_model != RAYS_AXI_MAG_EQDSK_SPLINE && which == RAYS_DEP_PTOTAL_RHO)
    return fail("axisym_toroid_rho: rho is only implemented for eqdsk_magnetics_spline_interp");  // axisym_toroid_eq_m.f90:398-430
  if (p->equilib_model == RAYS_EQ_SLAB ? which != RAYS_DEP_PTOTAL_X
                                       : (which != RAYS_DEP_PTOTAL_PSI && which != RAYS_DEP_PTOTAL_RHO))
    return fail("initialize_deposition_profiles: unimplemented profile for this equilib_model");  // :162-169, 204-212
  if (n_bins < 1 || nray < 0) return fail("rays_hip_deposition: bad n_bins / nray");
  if (!d_ray_vec || !d_npoints || !d_initial_ray_power || !d_work || !d_profile_out)
    return fail("rays_hip_deposition: null device pointer");
  rays::DevParams D = make_dev_params(*p);
  if (p->equilib_model == RAYS_EQ_AXISYM) {
    rc = get_axisym_device(&D);
    if (rc) return rc;
  }
  rays::DepArgs A;
  A.which = which;
  A.n_bins = n_bins; A.nray = nray; A.nv = p->nv; A.npt = p->nstep_max + 1;
  A.grid_min = 0.0; A.grid_max = 1.0;  // :176-177
  if (which == RAYS_DEP_PTOTAL_X) { A.grid_min = p->slab.xmin; A.grid_max = p->slab.xmax; }  // :136-137
  A.ray_vec = d_ray_vec; A.npoints = d_npoints; A.power = d_initial_ray_power; A.work = d_work;
  A.rho_grid = nullptr; A.rho_fspl = nullptr; A.n_rho = 0;
  if (which == RAYS_DEP_PTOTAL_RHO) {
    const double* t = nullptr;
    int n = 0;
    rc = get_rho_device(&t, &n);
    if (rc) return rc;
    A.rho_grid = t; A.rho_fspl = t + n; A.n_rho = n;
  }
  hipError_t e = rays::launch_deposition(D, A, d_profile_in, d_profile_out, (hipStream_t)hip_stream);
  if (e != hipSuccess) return hip_fail(e, "deposition kernels");
  return 0;
}

// Host-pointer form of the deposition profiles: what a Fortran post-processor holding the ray_results_m arrays
// calls instead of calculate_deposition_profiles (deposition_profiles_m.f90:228-260).  Only points 1..maxval(npoints)
// of every ray cross PCIe (a strided copy); work comes back in the reference's work(n_bins, nray) layout.
int rays_hip_deposition(const rays_params_t* p, int which, int n_bins, int nray, const double* ray_vec,
                        const int32_t* npoints, const double* initial_ray_power, double* work, double* profile) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (n_bins < 1 || nray < 0) return fail("rays_hip_deposition: bad n_bins / nray");
  if (nray == 0) {
    if (profile) std::memset(profile, 0, sizeof(double) * (size_t)n_bins);
    return 0;
  }
  if (!ray_vec || !npoints || !initial_ray_power || !profile) return fail("rays_hip_deposition: null array argument");
  const size_t nv = (size_t)p->nv, npt = (size_t)p->nstep_max + 1;
  int maxnp = 1;
  for (int i = 0; i < nray; i++) maxnp = std::max(maxnp, (int)npoints[i]);
  if ((size_t)maxnp > npt) return fail("rays_hip_deposition: npoints exceeds nstep_max + 1");
  const bool timing = std::getenv("RAYS_HIP_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  rays_params_t q = *p;
  q.nstep_max = maxnp - 1;  // the device copy holds maxnp points per ray
  double *d_rv = nullptr, *d_pw = nullptr, *d_work = nullptr, *d_prof = nullptr;
  int32_t* d_np = nullptr;
  auto release = [&]() { (void)hipFree(d_rv); (void)hipFree(d_pw); (void)hipFree(d_work); (void)hipFree(d_prof); (void)hipFree(d_np); };
#define DEP_TRY(call)                                                \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) { release(); return hip_fail(e_, #call); } \
  } while (0)
  DEP_TRY(hipMalloc(&d_rv, sizeof(double) * nv * (size_t)maxnp * (size_t)nray));
  DEP_TRY(hipMalloc(&d_pw, sizeof(double) * (size_t)nray));
  DEP_TRY(hipMalloc(&d_work, sizeof(double) * (size_t)n_bins * (size_t)nray));
  DEP_TRY(hipMalloc(&d_prof, sizeof(double) * (size_t)n_bins));
  DEP_TRY(hipMalloc(&d_np, sizeof(int32_t) * (size_t)nray));
  DEP_TRY(hipMemcpy2D(d_rv, sizeof(double) * nv * (size_t)maxnp, ray_vec, sizeof(double) * nv * npt,
                      sizeof(double) * nv * (size_t)maxnp, (size_t)nray, hipMemcpyHostToDevice));
  DEP_TRY(hipMemcpy(d_pw, initial_ray_power, sizeof(double) * (size_t)nray, hipMemcpyHostToDevice));
  DEP_TRY(hipMemcpy(d_np, npoints, sizeof(int32_t) * (size_t)nray, hipMemcpyHostToDevice));
  rc = rays_hip_deposition_device(&q, which, n_bins, nray, d_rv, d_np, d_pw, d_work, nullptr, d_prof, nullptr);
  if (rc) { release(); return rc; }
  DEP_TRY(hipDeviceSynchronize());
  DEP_TRY(hipMemcpy(profile, d_prof, sizeof(double) * (size_t)n_bins, hipMemcpyDeviceToHost));
  if (work) {  // device: [n_bins][nray]  ->  reference work(n_bins, nray) = C [nray][n_bins]
    std::vector<double> w((size_t)n_bins * (size_t)nray);
    DEP_TRY(hipMemcpy(w.data(), d_work, sizeof(double) * w.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < nray; r++)
      for (int b = 0; b < n_bins; b++) work[(size_t)r * n_bins + b] = w[(size_t)b * nray + r];
  }
#undef DEP_TRY
  release();
  if (timing)
    std::fprintf(stderr, "[rays_hip_deposition] %d rays, %.1f MB of trajectories uploaded: %.2f ms\n", nray,
                 1e-6 * sizeof(double) * nv * (double)maxnp * (double)nray,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}

// The deposition profiles of the rays the last rays_hip_trace call traced, binned where they lie (see KeptResult).
// Blocks are binned in ray order, each continuing the running sums of the one before it (rays_hip_deposition_device's
// d_profile_in): the profile is the reference's ray-ordered sum bit for bit, whatever the number of blocks / devices.
int rays_hip_deposition_last(const rays_params_t* p, int which, int n_bins, int nray, const double* initial_ray_power,
                             double* work, double* profile) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (n_bins < 1 || nray < 0 || !initial_ray_power || !profile) return fail("rays_hip_deposition_last: bad argument");
  std::vector<KeptBlock> blocks;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_kept.keep || g_kept.blocks.empty() || g_kept.nray != nray || g_kept.nv != p->nv || g_kept.nstep_max != p->nstep_max) {
      g_err = "rays_hip_deposition_last: no device-resident result of a rays_hip_trace call with this shape "
              "(rays_hip_keep_last_result(1) before the trace; same nray, nv, nstep_max)";
      return RAYS_HIP_NO_KEPT_RESULT;
    }
    blocks = g_kept.blocks;
  }
  const bool timing = std::getenv("RAYS_HIP_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<double> carry((size_t)n_bins, 0.0), wbuf;
  bool have_carry = false;
  for (const KeptBlock& b : blocks) {
    const int n = b.r1 - b.r0;
    if (n <= 0) continue;
    HIP_TRY(hipSetDevice(b.device));
    double *d_pw = nullptr, *d_work = nullptr, *d_in = nullptr, *d_out = nullptr;
    auto release = [&]() { (void)hipFree(d_pw); (void)hipFree(d_work); (void)hipFree(d_in); (void)hipFree(d_out); };
#define DEPL_TRY(call)                                               \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) { release(); return hip_fail(e_, #call); } \
  } while (0)
    DEPL_TRY(hipMalloc(&d_pw, sizeof(double) * (size_t)n));
    DEPL_TRY(hipMalloc(&d_work, sizeof(double) * (size_t)n_bins * (size_t)n));
    DEPL_TRY(hipMalloc(&d_in, sizeof(double) * (size_t)n_bins));
    DEPL_TRY(hipMalloc(&d_out, sizeof(double) * (size_t)n_bins));
    DEPL_TRY(hipMemcpy(d_pw, initial_ray_power + b.r0, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    if (have_carry) DEPL_TRY(hipMemcpy(d_in, carry.data(), sizeof(double) * (size_t)n_bins, hipMemcpyHostToDevice));
    rc = rays_hip_deposition_device(p, which, n_bins, n, b.d_ray_vec, b.d_npoints, d_pw, d_work, have_carry ? d_in : nullptr,
                                    d_out, nullptr);
    if (rc) { release(); return rc; }
    DEPL_TRY(hipDeviceSynchronize());
    DEPL_TRY(hipMemcpy(carry.data(), d_out, sizeof(double) * (size_t)n_bins, hipMemcpyDeviceToHost));
    have_carry = true;
    if (work) {  // device: [n_bins][n]  ->  reference work(n_bins, nray) = C [nray][n_bins]
      wbuf.resize((size_t)n_bins * (size_t)n);
      DEPL_TRY(hipMemcpy(wbuf.data(), d_work, sizeof(double) * wbuf.size(), hipMemcpyDeviceToHost));
      for (int r = 0; r < n; r++)
        for (int bb = 0; bb < n_bins; bb++) work[(size_t)(b.r0 + r) * n_bins + bb] = wbuf[(size_t)bb * n + r];
    }
#undef DEPL_TRY
    release();
  }
  std::memcpy(profile, carry.data(), sizeof(double) * (size_t)n_bins);
  if (timing)
    std::fprintf(stderr, "[rays_hip_deposition_last] %d rays in %zu device-resident block(s), no trajectory upload: %.2f ms\n",
                 nray, blocks.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}

int rays_hip_sizeof_fan(void) { return (int)sizeof(rays_fan_t); }

#include "rays_fan_setup.inc"

static int ray_init_run(const rays_params_t* p, const rays_fan_t* fan, int nray_max, double* d_rvec0,
                        double* d_rindex_vec0, int32_t* nray, hipStream_t stream,
                        std::vector<int>* first_of_launch_host, int* per_r_launch) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (!nray || !d_rvec0 || !d_rindex_vec0) return fail("rays_hip_ray_init: null pointer");
  rays::FanArgs F;
  std::vector<double> launch;
  const char* why = "";
  if (fan_setup(p, fan, nray_max, &F, &launch, per_r_launch, &why)) return fail(why);
  const int n_cand = F.n_launch * F.n_a * F.n_b;
  const int nb = (n_cand + rays::ray_init_block() - 1) / rays::ray_init_block();
  rays::DevParams D = make_dev_params(*p);
  if (p->equilib_model == RAYS_EQ_AXISYM) {
    rc = get_axisym_device(&D);
    if (rc) return rc;
  }
  double *d_launch = nullptr, *d_cand = nullptr;
  int *d_keep = nullptr, *d_bc = nullptr, *d_offs = nullptr, *d_first = nullptr;
  auto release = [&]() {
    (void)hipFree(d_launch); (void)hipFree(d_cand); (void)hipFree(d_keep); (void)hipFree(d_bc);
    (void)hipFree(d_offs); (void)hipFree(d_first);
  };
#define INIT_TRY(call)                                   \
  do {                                                   \
    hipError_t e_ = (call);                              \
    if (e_ != hipSuccess) { release(); return hip_fail(e_, #call); } \
  } while (0)
  INIT_TRY(hipMalloc(&d_launch, sizeof(double) * launch.size()));
  INIT_TRY(hipMalloc(&d_cand, sizeof(double) * 3 * (size_t)n_cand));
  INIT_TRY(hipMalloc(&d_keep, sizeof(int) * (size_t)n_cand));
  INIT_TRY(hipMalloc(&d_bc, sizeof(int) * (size_t)nb));
  INIT_TRY(hipMalloc(&d_offs, sizeof(int) * (size_t)(nb + 1)));
  INIT_TRY(hipMalloc(&d_first, sizeof(int) * (size_t)F.n_launch));
  INIT_TRY(hipMemcpyAsync(d_launch, launch.data(), sizeof(double) * launch.size(), hipMemcpyHostToDevice, stream));
  F.launch = d_launch;
  INIT_TRY(rays::launch_ray_init(p->equilib_model, p->nspec + 1, D, F, n_cand, d_cand, d_keep, d_bc, d_offs,
                                 d_first, d_rvec0, d_rindex_vec0, stream));
  int total = 0;
  INIT_TRY(hipMemcpyAsync(&total, d_offs + nb, sizeof(int), hipMemcpyDeviceToHost, stream));
  if (first_of_launch_host) {
    first_of_launch_host->resize(F.n_launch);
    INIT_TRY(hipMemcpyAsync(first_of_launch_host->data(), d_first, sizeof(int) * F.n_launch, hipMemcpyDeviceToHost, stream));
  }
  INIT_TRY(hipStreamSynchronize(stream));
#undef INIT_TRY
  release();
  *nray = total;
  if (total == 0) return fail("No successful ray initializations");  // simple_slab_ray_init_m.f90:172
  return 0;
}

int rays_hip_ray_init_device(const rays_params_t* p, const rays_fan_t* fan, int nray_max, double* d_rvec0,
                             double* d_rindex_vec0, int32_t* nray, void* hip_stream) {
  int per_r = 0;
  return ray_init_run(p, fan, nray_max, d_rvec0, d_rindex_vec0, nray, (hipStream_t)hip_stream, nullptr, &per_r);
}

int rays_hip_ray_init(const rays_params_t* p, const rays_fan_t* fan, int nray_max, double* rvec0,
                      double* rindex_vec0, double* ray_pwr_wt, int32_t* nray) {
  if (!rvec0 || !rindex_vec0 || !nray) return fail("rays_hip_ray_init: null pointer");
  if (nray_max <= 0) return fail("rays_hip_ray_init: nray_max <= 0");
  {  // configuration errors first (they need no device), as the reference's launchers `stop 1`
    int rc0 = rays_hip_check_params(p);
    if (rc0) return rc0;
    rays::FanArgs F;
    std::vector<double> launch;
    int per_r0 = 0;
    const char* why = "";
    if (fan_setup(p, fan, nray_max, &F, &launch, &per_r0, &why)) return fail(why);
  }
  double *d_r = nullptr, *d_n = nullptr;
  HIP_TRY(hipMalloc(&d_r, sizeof(double) * 3 * (size_t)nray_max));
  if (hipMalloc(&d_n, sizeof(double) * 3 * (size_t)nray_max) != hipSuccess) {
    (void)hipFree(d_r);
    return fail("rays_hip_ray_init: out of device memory");
  }
  std::vector<int> first;
  int per_r = 0;
  int rc = ray_init_run(p, fan, nray_max, d_r, d_n, nray, nullptr, &first, &per_r);
  if (rc == 0) {
    const size_t n = (size_t)*nray;
    hipError_t e = hipMemcpy(rvec0, d_r, sizeof(double) * 3 * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(rindex_vec0, d_n, sizeof(double) * 3 * n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy (ray init results)");
    if (rc == 0 && ray_pwr_wt) {
      if (fan->model == RAYS_RAY_INIT_SIMPLE_SLAB) {  // :179,182: 1/nray, divided by nray once more
        for (size_t i = 0; i < n; i++) ray_pwr_wt[i] = 1.0 / (double)n / (double)n;
      } else if (fan->model == RAYS_RAY_INIT_AXISYM_R_Z_NPHI_NTHETA) {
        for (size_t i = 0; i < n; i++) ray_pwr_wt[i] = 1.0 / (double)n;
      } else {
        // solovev_ray_init_nphi_ntheta_m.f90:196: only ray_pwr_wt(count) = 1. after each r-launch
        // loop; the other entries are left unset by the reference (zero here)
        for (size_t i = 0; i < n; i++) ray_pwr_wt[i] = 0.;
        const int nl = (int)first.size();
        for (int ir = 0; per_r > 0 && ir < nl / per_r; ir++) {
          const int end = (ir + 1) * per_r < nl ? first[(ir + 1) * per_r] : (int)n;  // count after this r loop
          if (end >= 1) ray_pwr_wt[end - 1] = 1.;
        }
      }
    }
  }
  (void)hipFree(d_r);
  (void)hipFree(d_n);
  return rc;
}

int rays_hip_probe(const rays_params_t* p, int n, const double* v, double* cold7, double* num7,
                   double* dvds, double* resid, int32_t* codes) {
  int rc = rays_hip_check_params(p);
  if (rc) return rc;
  if (n <= 0) return 0;
  const size_t nv = (size_t)p->nv;
  double *d_v = nullptr, *d_c = nullptr, *d_n = nullptr, *d_f = nullptr, *d_r = nullptr;
  int* d_k = nullptr;
  auto release = [&]() {
    (void)hipFree(d_v); (void)hipFree(d_c); (void)hipFree(d_n); (void)hipFree(d_f); (void)hipFree(d_r); (void)hipFree(d_k);
  };
#define PROBE_TRY(call)                                                  \
  do {                                                                   \
    hipError_t e_ = (call);                                              \
    if (e_ != hipSuccess) { release(); return hip_fail(e_, #call); }     \
  } while (0)
  PROBE_TRY(hipMalloc(&d_v, sizeof(double) * nv * n));
  PROBE_TRY(hipMalloc(&d_c, sizeof(double) * 7 * n));
  PROBE_TRY(hipMalloc(&d_n, sizeof(double) * 7 * n));
  PROBE_TRY(hipMalloc(&d_f, sizeof(double) * nv * n));
  PROBE_TRY(hipMalloc(&d_r, sizeof(double) * n));
  PROBE_TRY(hipMalloc(&d_k, sizeof(int) * 4 * n));
  PROBE_TRY(hipMemcpy(d_v, v, sizeof(double) * nv * n, hipMemcpyHostToDevice));
  rays::DevParams D = make_dev_params(*p);
  if (p->equilib_model == RAYS_EQ_AXISYM) {
    rc = get_axisym_device(&D);
    if (rc) { release(); return rc; }
  }
  hipLaunchKernelGGL(rays::probe_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, D, p->equilib_model,
                     p->nspec + 1, p->nv, n, d_v, d_c, d_n, d_f, d_r, d_k);
  PROBE_TRY(hipGetLastError());
  PROBE_TRY(hipMemcpy(cold7, d_c, sizeof(double) * 7 * n, hipMemcpyDeviceToHost));
  PROBE_TRY(hipMemcpy(num7, d_n, sizeof(double) * 7 * n, hipMemcpyDeviceToHost));
  PROBE_TRY(hipMemcpy(dvds, d_f, sizeof(double) * nv * n, hipMemcpyDeviceToHost));
  PROBE_TRY(hipMemcpy(resid, d_r, sizeof(double) * n, hipMemcpyDeviceToHost));
  PROBE_TRY(hipMemcpy(codes, d_k, sizeof(int) * 4 * n, hipMemcpyDeviceToHost));
#undef PROBE_TRY
  release();
  return 0;
}

}  // extern "C"

#include "rays_gather.inc"
static void rccl_close_all() { rccl_close(); }
