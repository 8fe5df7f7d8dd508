// TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so in the CPU test tier (loaded by rays_hip_trace_gather through
// RAYS_HIP_RCCL_LIB), so that the multi-device gather -- ncclCommInitAll, one grouped batch of ncclSend / ncclRecv,
// per-peer offsets, unpack -- runs under ASan + UBSan next to the emulated HIP runtime (hip/hip_runtime_api_emul.h).
// Semantics kept from RCCL (rccl.h:236, 700, 722): one communicator per device of the list, rank = position in the
// list; point-to-point calls must sit inside ncclGroupStart / ncclGroupEnd; a send from rank r to peer p pairs with the
// OLDEST unpaired receive of rank p from peer r in the group, count and datatype must agree; every send and every
// receive must find its partner before the group ends.  The copy happens at ncclGroupEnd.  A communicator may only be
// used with a stream of its own device (the emulated hipStream_t starts with its device ordinal).
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
enum { kSuccess = 0, kInvalidArgument = 4, kInvalidUsage = 5 };
struct Comm { int rank, nranks, device; long long world; bool alive; };
struct Op { bool send; void* buf; size_t count; int dtype, peer; Comm* comm; bool paired; };
std::mutex g_mu;
long long g_worlds = 0;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
long long g_groups = 0, g_pairs = 0, g_bytes = 0;

size_t dtype_bytes(int t) {
  switch (t) {
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: case 9: return 2;
    default: return 0;
  }
}
int post(bool send, void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
  Comm* c = (Comm*)comm;
  if (!c || !c->alive || peer < 0 || peer >= c->nranks || peer == c->rank || dtype_bytes(dtype) == 0 || (count && !buf))
    return kInvalidArgument;
  if (t_depth == 0) return kInvalidUsage;  // a lone blocking send / receive would dead-lock a single host thread
  if (stream && *(const int*)stream != c->device) {
    std::fprintf(stderr, "[fake_rccl] rank %d (device %d) used with a stream of device %d\n", c->rank, c->device, *(const int*)stream);
    return kInvalidUsage;
  }
  t_ops.push_back(Op{send, buf, count, dtype, peer, c, false});
  return kSuccess;
}
}  // namespace

extern "C" {
int ncclCommInitAll(void** comms, int ndev, const int* devlist) {
  if (!comms || ndev < 1) return kInvalidArgument;
  for (int i = 0; i < ndev; i++)
    for (int j = i + 1; devlist && j < ndev; j++)
      if (devlist[i] == devlist[j]) return kInvalidUsage;  // one rank per device
  std::lock_guard<std::mutex> lk(g_mu);
  const long long w = ++g_worlds;
  for (int i = 0; i < ndev; i++) comms[i] = new Comm{i, ndev, devlist ? devlist[i] : i, w, true};
  return kSuccess;
}
int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c || !c->alive) return kInvalidArgument;
  c->alive = false;
  delete c;
  return kSuccess;
}
int ncclGroupStart() { t_depth++; return kSuccess; }
int ncclGroupEnd() {
  if (t_depth <= 0) return kInvalidUsage;
  if (--t_depth > 0) return kSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  int rc = kSuccess;
  for (Op& s : ops) {
    if (!s.send) continue;
    for (Op& r : ops) {
      if (r.send || r.paired || r.comm->world != s.comm->world || r.comm->rank != s.peer || r.peer != s.comm->rank) continue;
      if (r.count != s.count || r.dtype != s.dtype) {
        std::fprintf(stderr, "[fake_rccl] send %d -> %d: %zu x type %d meets a receive of %zu x type %d\n", s.comm->rank, s.peer,
                     s.count, s.dtype, r.count, r.dtype);
        rc = kInvalidUsage;
      } else if (s.count) {
        std::memcpy(r.buf, s.buf, s.count * dtype_bytes(s.dtype));
      }
      r.paired = s.paired = true;
      std::lock_guard<std::mutex> lk(g_mu);
      g_pairs++;
      g_bytes += (long long)(s.count * dtype_bytes(s.dtype));
      break;
    }
  }
  for (const Op& o : ops)
    if (!o.paired) {
      std::fprintf(stderr, "[fake_rccl] unpaired %s of rank %d, peer %d (%zu elements)\n", o.send ? "send" : "receive", o.comm->rank,
                   o.peer, o.count);
      rc = kInvalidUsage;  // the real library would hang here
    }
  std::lock_guard<std::mutex> lk(g_mu);
  g_groups++;
  return rc;
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
  return post(true, const_cast<void*>(buf), count, dtype, peer, comm, stream);
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
  return post(false, buf, count, dtype, peer, comm, stream);
}
const char* ncclGetErrorString(int rc) {
  return rc == kSuccess ? "no error" : rc == kInvalidArgument ? "invalid argument" : rc == kInvalidUsage ? "invalid usage" : "error";
}
// test hook: what went through the stand-in so far
void fake_rccl_stats(long long* groups, long long* pairs, long long* bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  *groups = g_groups; *pairs = g_pairs; *bytes = g_bytes;
}
}
