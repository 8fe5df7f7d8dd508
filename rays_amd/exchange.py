"""Multi-GPU side of the hot path: ray sharding and the final trajectory gather.

Rays are independent, so the only exchange is the gather of each rank's trajectories to rank 0
(the reference writes straight into shared arrays from its OpenMP threads, ray_tracing.f90:62-64).

  * partition: contiguous blocks, rank r owns rays [r*per, (r+1)*per) -- the reference's
    `schedule(static)`; a block is a contiguous slab of ray_vec(nv, nstep_max+1, nray).
  * payload: the padded slab is ~80 % zeros (a ray uses npoints of nstep_max+1 slots), so each
    rank packs its slab to [sum(npoints)][nv] (+ residual) first; rank 0 receives every peer's
    packed block with grouped send/recv (one RCCL group: each peer uses its own xGMI link to the
    root, so the transfers are link-parallel, not ring-bound) and unpacks it into the peer's slab
    of the padded global arrays.
  * no other collective exists on this path.

When the consumer of the trajectories is the deposition-profile post-processor the
trajectories need not leave their GPU at all: `ProfileChain` replaces the gather by a chain of
n_bins-sized partial sums (rank r continues rank r-1's running sum over its own rays, in ray
order), which reproduces the single-process profile bit for bit and moves a few KB per rank.

pack / unpack are injected: on GPUs they are the HIP kernels of librays_hip.so
(rays_hip_pack_device / rays_hip_unpack_device); the world_size-2 gloo test passes CPU stand-ins
to exercise the protocol without a GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple


def shard_bounds(nray: int, world: int, rank: int) -> Tuple[int, int]:
    per = (nray + world - 1) // world
    return min(nray, rank * per), min(nray, (rank + 1) * per)


def hip_pack(nray, nv, nstep_max, npoints, offsets, ray_vec, residual, packed_vec, packed_res, stream):
    from . import hip

    hip.pack_device(nray, nv, nstep_max, npoints.data_ptr(), offsets.data_ptr(), ray_vec.data_ptr(),
                    residual.data_ptr(), packed_vec.data_ptr(), packed_res.data_ptr(), stream)


def hip_unpack(nray, nv, nstep_max, npoints, offsets, packed_vec, packed_res, ray_vec, residual, stream):
    from . import hip

    hip.unpack_device(nray, nv, nstep_max, npoints.data_ptr(), offsets.data_ptr(),
                      packed_vec.data_ptr(), packed_res.data_ptr(), ray_vec.data_ptr(),
                      residual.data_ptr(), stream)


class TrajectoryGather:
    """Gathers (ray_vec, residual, npoints, stop_code) of all ranks into rank 0's global arrays."""

    def __init__(self, nray_total: int, nv: int, nstep_max: int, device, pack: Callable = hip_pack,
                 unpack: Callable = hip_unpack, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.group = group
        self.nray_total, self.nv, self.nstep_max = nray_total, nv, nstep_max
        self.device = device
        self.pack, self.unpack = pack, unpack
        self.bounds = [shard_bounds(nray_total, self.world, r) for r in range(self.world)]
        self.nrays = [hi - lo for lo, hi in self.bounds]
        n_local = self.nrays[self.rank]
        f64, i64, i32 = torch.float64, torch.int64, torch.int32
        self.offsets = torch.zeros(n_local, dtype=i64, device=device)
        self.packed_vec = self.packed_res = None
        if self.rank == 0:
            npt = nstep_max + 1
            self.ray_vec = torch.zeros((nray_total, npt, nv), dtype=f64, device=device)
            self.residual = torch.zeros((nray_total, npt), dtype=f64, device=device)
            self.npoints = torch.zeros(nray_total, dtype=i32, device=device)
            self.stop_code = torch.zeros(nray_total, dtype=i32, device=device)
            self._r_off = [torch.zeros(n, dtype=i64, device=device) for n in self.nrays]
        self._counts = None
        self._pending = None
        self._buf = 0

    def _stream(self):
        t = self.torch
        return t.cuda.current_stream().cuda_stream if self.device.type == "cuda" else 0

    def prepare(self, npoints_local):
        """Exchange per-rank point counts and size the buffers (once per fan; not per step).
        Send/receive staging is double-buffered so the exchange of pass i can overlap the trace of
        pass i+1 (gather_async)."""
        t, dist = self.torch, self.dist
        mine = t.tensor([int(npoints_local.to(t.int64).sum().item())], dtype=t.int64, device=self.device)
        counts = [t.zeros(1, dtype=t.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(counts, mine, group=self.group)
        self._counts = [int(c.item()) for c in counts]
        cap = max(self._counts[self.rank], 1)
        n_local = self.nrays[self.rank]
        f64, i32 = t.float64, t.int32
        self._send = [dict(vec=t.empty((cap, self.nv), dtype=f64, device=self.device),
                           res=t.empty(cap, dtype=f64, device=self.device),
                           npoints=t.zeros(n_local, dtype=i32, device=self.device),
                           stop=t.zeros(n_local, dtype=i32, device=self.device)) for _ in range(2)]
        self.packed_vec, self.packed_res = self._send[0]["vec"], self._send[0]["res"]
        if self.rank == 0:
            self._recv = []
            for _ in range(2):
                slot = {}
                for r in range(1, self.world):
                    c = max(self._counts[r], 1)
                    slot[r] = dict(vec=t.empty((c, self.nv), dtype=f64, device=self.device),
                                   res=t.empty(c, dtype=f64, device=self.device),
                                   npoints=t.zeros(self.nrays[r], dtype=i32, device=self.device),
                                   stop=t.zeros(self.nrays[r], dtype=i32, device=self.device))
                self._recv.append(slot)
        self._pending = None
        self._buf = 0

    # ---- asynchronous form: start the exchange of this pass, complete the previous one -------
    def gather_async(self, ray_vec, residual, npoints, stop_code):
        """Pack this pass's slab and start its exchange; complete the previous pass's exchange.
        The caller launches the next trace right after: RCCL's stream carries the transfers while
        the compute stream runs the next pass.  Call finish() after the last pass."""
        t, dist = self.torch, self.dist
        if self._counts is None:
            self.prepare(npoints)
        stream = self._stream()
        b = self._buf
        n_local = self.nrays[self.rank]
        snd = self._send[b]
        t.cumsum(npoints, 0, dtype=t.int64, out=self.offsets)
        self.offsets.sub_(npoints)
        self.pack(n_local, self.nv, self.nstep_max, npoints, self.offsets, ray_vec, residual,
                  snd["vec"], snd["res"], stream)
        snd["npoints"].copy_(npoints)
        snd["stop"].copy_(stop_code)
        self._complete()  # previous pass (its buffers are the other slot)
        works = []
        if self.rank == 0:
            lo, hi = self.bounds[0]
            self.ray_vec[lo:hi].copy_(ray_vec)
            self.residual[lo:hi].copy_(residual)
            self.npoints[lo:hi].copy_(npoints)
            self.stop_code[lo:hi].copy_(stop_code)
            ops = []
            for r in range(1, self.world):
                if self.nrays[r] == 0:
                    continue
                rc = self._recv[b][r]
                ops += [dist.P2POp(dist.irecv, rc["npoints"], r, self.group),
                        dist.P2POp(dist.irecv, rc["stop"], r, self.group),
                        dist.P2POp(dist.irecv, rc["vec"], r, self.group),
                        dist.P2POp(dist.irecv, rc["res"], r, self.group)]
            works = dist.batch_isend_irecv(ops) if ops else []
        elif n_local > 0:
            ops = [dist.P2POp(dist.isend, snd["npoints"], 0, self.group),
                   dist.P2POp(dist.isend, snd["stop"], 0, self.group),
                   dist.P2POp(dist.isend, snd["vec"], 0, self.group),
                   dist.P2POp(dist.isend, snd["res"], 0, self.group)]
            works = dist.batch_isend_irecv(ops)
        self._pending = (b, works)
        self._buf = 1 - b

    def _complete(self):
        if self._pending is None:
            return
        t = self.torch
        b, works = self._pending
        self._pending = None
        for w in works:
            w.wait()
        if self.rank == 0:
            stream = self._stream()
            for r in range(1, self.world):
                b0, b1 = self.bounds[r]
                if b1 == b0:
                    continue
                rc = self._recv[b][r]
                self.npoints[b0:b1].copy_(rc["npoints"])
                self.stop_code[b0:b1].copy_(rc["stop"])
                t.cumsum(rc["npoints"], 0, dtype=t.int64, out=self._r_off[r])
                self._r_off[r].sub_(rc["npoints"])
                # the peer's slab may hold longer rays from an earlier pass only if the fan changed;
                # passes over the same fan rewrite the same entries, so no re-zeroing is needed
                self.unpack(b1 - b0, self.nv, self.nstep_max, rc["npoints"], self._r_off[r], rc["vec"],
                            rc["res"], self.ray_vec[b0:b1], self.residual[b0:b1], stream)

    def finish(self):
        self._complete()

    def gather(self, ray_vec, residual, npoints, stop_code):
        """Synchronous form: returns with rank 0's global arrays complete (stream-ordered)."""
        self.gather_async(ray_vec, residual, npoints, stop_code)
        self.finish()


class SummaryGather:
    """The exchange of a run whose trajectories STAY on the GPU that traced them (a consumer with device-side
    post-processing): rank 0 receives only the per-ray summaries of ray_results_m -- npoints, the stop
    code, end_ray_vec(nv), end_residuals, max_residuals: 8 (nv + 2) + 8 bytes per ray (~ 80-100 B) instead of the ray's
    packed trajectory (64 B per recorded step).  One grouped batch of receives on rank 0, one send per peer; the rows
    travel as one float64 block per rank (npoints and stop codes are exact in a double)."""

    def __init__(self, nray_total: int, nv: int, device, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.nv, self.device = nv, device
        self.bounds = [shard_bounds(nray_total, self.world, r) for r in range(self.world)]
        self.nrays = [hi - lo for lo, hi in self.bounds]
        self.width = nv + 4
        f64 = torch.float64
        self._send = torch.zeros((self.nrays[self.rank], self.width), dtype=f64, device=device)
        if self.rank == 0:
            self.table = torch.zeros((nray_total, self.width), dtype=f64, device=device)

    def bytes_per_ray(self) -> int:
        return 8 * self.width

    def gather(self, npoints, stop_code, end_ray_vec, end_residuals, max_residuals):
        """Returns rank 0's [nray_total][nv + 4] table (columns: npoints, stop code, end_ray_vec, end_residuals,
        max_residuals), None elsewhere."""
        t, dist = self.torch, self.dist
        s = self._send
        s[:, 0].copy_(npoints)
        s[:, 1].copy_(stop_code)
        s[:, 2:2 + self.nv].copy_(end_ray_vec)
        s[:, 2 + self.nv].copy_(end_residuals)
        s[:, 3 + self.nv].copy_(max_residuals)
        if self.rank == 0:
            lo, hi = self.bounds[0]
            self.table[lo:hi].copy_(s)
            ops = [dist.P2POp(dist.irecv, self.table[self.bounds[r][0]:self.bounds[r][1]], r, self.group)
                   for r in range(1, self.world) if self.nrays[r] > 0]
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            return self.table
        if self.nrays[self.rank] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, s, 0, self.group)]):
                w.wait()
        return None


class ProfileChain:
    """Ray-ordered reduction of deposition profiles across ranks holding consecutive ray blocks.

    The reference sums work(n_bins, nray) over rays sequentially (deposition_profiles_m.f90:246);
    floating-point addition is not associative, so an all-reduce would change the bits.  Rank r
    receives the running sums of ranks 0..r-1, lets `accumulate(carry_or_None, out)` add its own
    rays in order (rays_hip_deposition_device with d_profile_in = carry), and passes the result on;
    the last rank holds the total and sends it to rank 0.  Payload: n_bins doubles per hop."""

    def __init__(self, n_bins: int, device, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.carry = torch.zeros(n_bins, dtype=torch.float64, device=device)
        self.out = torch.zeros(n_bins, dtype=torch.float64, device=device)

    def reduce_unordered(self, accumulate: Callable):
        """Fast form: every rank sums its own rays (in order) and the partial profiles are combined
        by one RCCL reduce to rank 0.  Differs from the single-process profile only by the
        association of the per-rank partial sums (~1e-16 relative).  The ordered chain below costs
        one dependent add per ray and bin across ALL ranks, i.e. it serialises the ranks."""
        accumulate(None, self.out)
        if self.world > 1:
            self.dist.reduce(self.out, dst=0, op=self.dist.ReduceOp.SUM, group=self.group)
        return self.out if self.rank == 0 else None

    def reduce(self, accumulate: Callable):
        """Bit-exact form.  Returns the total profile on rank 0 (None elsewhere)."""
        d = self.dist
        if self.rank > 0:
            d.recv(self.carry, src=self.rank - 1, group=self.group)
            accumulate(self.carry, self.out)
        else:
            accumulate(None, self.out)
        if self.world == 1:
            return self.out
        if self.rank < self.world - 1:
            d.send(self.out, dst=self.rank + 1, group=self.group)
        else:
            d.send(self.out, dst=0, group=self.group)
        if self.rank == 0:
            d.recv(self.out, src=self.world - 1, group=self.group)
            return self.out
        return None
