"""CPU tier: the N>1 path -- ray sharding + packed trajectory gather -- with world_size 2 over
gloo.  The compute on each rank is the CPU oracle and pack/unpack are torch stand-ins, so only the
exchange protocol (rays_amd/exchange.py) is under test here; on GPUs the same class runs with the
HIP pack/unpack kernels over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.common import load_golden


def _pack(nray, nv, nstep_max, npoints, offsets, ray_vec, residual, packed_vec, packed_res, stream):
    for r in range(nray):
        n, o = int(npoints[r]), int(offsets[r])
        packed_vec[o:o + n] = ray_vec[r, :n]
        packed_res[o:o + n] = residual[r, :n]


def _unpack(nray, nv, nstep_max, npoints, offsets, packed_vec, packed_res, ray_vec, residual, stream):
    for r in range(nray):
        n, o = int(npoints[r]), int(offsets[r])
        ray_vec[r, :n] = packed_vec[o:o + n]
        residual[r, :n] = packed_res[o:o + n]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rays_amd.exchange import TrajectoryGather, shard_bounds
        from tests import oracle_lib

        g, nml, p = load_golden("cfg2_solovev1024_rk4")
        r0, n0 = g["rvec0"][:9], g["rindex_vec0"][:9]  # 9 rays -> uneven blocks 5 + 4
        lo, hi = shard_bounds(len(r0), world, rank)
        out = oracle_lib.trace(p, r0[lo:hi], n0[lo:hi], nthreads=1)
        tg = TrajectoryGather(len(r0), p.nv, p.nstep_max, torch.device("cpu"), pack=_pack, unpack=_unpack)
        args = (torch.from_numpy(out["ray_vec"]), torch.from_numpy(out["residual"]),
                torch.from_numpy(out["npoints"]), torch.from_numpy(out["stop_code"]))
        tg.gather(*args)                 # synchronous form
        for _ in range(3):               # pipelined form: buffers alternate, last pass drained by finish()
            tg.gather_async(*args)
        tg.finish()
        if rank == 0:
            full = oracle_lib.trace(p, r0, n0, nthreads=1)
            ok = (np.array_equal(tg.ray_vec.numpy(), full["ray_vec"])
                  and np.array_equal(tg.residual.numpy(), full["residual"])
                  and np.array_equal(tg.npoints.numpy(), full["npoints"])
                  and np.array_equal(tg.stop_code.numpy(), full["stop_code"]))
            q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _profile_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rays_amd.exchange import ProfileChain, shard_bounds
        from tests import emul_lib
        from tests.common import padded_full_trajectories

        g, nml, p = load_golden("gold_axisym64_eqdsk_damp_rk4")
        rv, nb = padded_full_trajectories(g, p), int(g["dep_n_bins"])
        lo, hi = shard_bounds(rv.shape[0], world, rank)
        work, _ = emul_lib.deposition(p, 0, nb, rv[lo:hi], g["npoints_full"][lo:hi], g["dep_power"][lo:hi],
                                      g["dep_rho_grid"], g["dep_rho_fspl"])

        def accumulate(carry, out):   # CPU stand-in of rays_hip_deposition_device's ordered sum
            s = np.zeros(nb) if carry is None else carry.numpy().copy()
            for r in range(work.shape[0]):
                s = s + work[r]
            out.copy_(torch.from_numpy(s))

        chain = ProfileChain(nb, torch.device("cpu"))
        tot = chain.reduce(accumulate)
        exact = bool(np.array_equal(tot.numpy(), g["dep_profile"][0])) if rank == 0 else None
        fast = chain.reduce_unordered(accumulate)   # per-rank sums + reduce: same to rounding
        if rank == 0:
            close = bool(np.allclose(fast.numpy(), g["dep_profile"][0], rtol=1e-12, atol=1e-300))
            q.put(exact and close)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_profile_chain_gloo():
    """Deposition profile over two ranks: chained partial sums == the reference's
    single-process profile, bit for bit."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_profile_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=300)
    assert all(pr.exitcode == 0 for pr in procs)
    assert q.get(timeout=5) is True


def test_shard_bounds():
    from rays_amd.exchange import shard_bounds
    for n, w in ((9, 2), (65536, 8), (3, 8), (0, 2)):
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def test_two_rank_gather_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=300)
        assert pr.exitcode == 0
    assert q.get(timeout=10) is True


def _summary_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rays_amd.exchange import SummaryGather, shard_bounds
        from tests import oracle_lib

        g, nml, p = load_golden("cfg2_solovev1024_rk4")
        r0, n0 = g["rvec0"][:9], g["rindex_vec0"][:9]  # uneven blocks 5 + 4
        lo, hi = shard_bounds(len(r0), world, rank)
        out = oracle_lib.trace(p, r0[lo:hi], n0[lo:hi], nthreads=1)
        sg = SummaryGather(len(r0), p.nv, torch.device("cpu"))
        tab = None
        for _ in range(2):
            tab = sg.gather(*(torch.from_numpy(out[k]) for k in ("npoints", "stop_code", "end_ray_vec", "end_residuals",
                                                                 "max_residuals")))
        if rank == 0:
            full = oracle_lib.trace(p, r0, n0, nthreads=1)
            t = tab.numpy()
            ok = (np.array_equal(t[:, 0], full["npoints"]) and np.array_equal(t[:, 1], full["stop_code"])
                  and np.array_equal(t[:, 2:2 + p.nv], full["end_ray_vec"], equal_nan=True)
                  and np.array_equal(t[:, 2 + p.nv], full["end_residuals"])
                  and np.array_equal(t[:, 3 + p.nv], full["max_residuals"]) and sg.bytes_per_ray() == 8 * (p.nv + 4))
            if not ok:
                print("summary gather mismatch:", [bool(np.array_equal(t[:, 0], full["npoints"])), bool(np.array_equal(t[:, 1], full["stop_code"])),
                      bool(np.array_equal(t[:, 2:2 + p.nv], full["end_ray_vec"])), bool(np.array_equal(t[:, 2 + p.nv], full["end_residuals"])),
                      bool(np.array_equal(t[:, 3 + p.nv], full["max_residuals"]))], flush=True)
            q.put(bool(ok))
        else:
            assert tab is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_summary_gather_gloo():
    """The exchange of a run whose trajectories stay on their GPUs: only ray_results_m's per-ray summaries reach rank 0."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_summary_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=300)
        assert pr.exitcode == 0
    assert q.get(timeout=10) is True
