#!/usr/bin/env python3
"""bench.py -- headline benchmark of the RAYS hot path on MI355X.

Metric (BASELINE.json): recorded ray-steps/sec, whole job, on the 64k-ray Solovev fan
(configs/cfg3b_solovev64k_rk4.in: 256 x 256 n_theta x n_phi fan, RK4_ODE, cold-plasma dD).

A "step" of the bench = ONE pass of the hot path over the fan: the trace kernel over all rays of
this rank (inputs resident in HBM), plus -- for N > 1 -- the RCCL gather of the trajectories to
rank 0.  Output arrays are zero-filled once before the timed region, exactly as the reference does
in initialize_ray_results_m (ray_results_m.f90:154-164), not inside trace_rays.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config configs/....in]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) starts its N ranks
itself -- child processes, spawned before the parent touches a GPU -- and relays rank 0's line.

N > 1 is weak scaling: the fan grows to 256*N x 256 rays over the same launch-angle range and is
block-partitioned, 65536 contiguous rays per rank (the reference's `schedule(static)`).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: vector FP64 (every instruction an FMA)
N_SIMD = 256 * 4         # 256 CUs x 4 SIMDs
FP64_CLK_PER_WAVE_INST = 4  # a wave64 FP64 VALU instruction occupies its SIMD's 16 lanes for 4 clocks


def visible_gpu_count():
    """GPUs this process would see, WITHOUT initialising HIP in it (the parent of a self-launched run never touches
    a GPU): the visibility variables if set, else the KFD topology (nodes with SIMDs are GPUs)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(top):
            for line in open(os.path.join(top, node, "properties")):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
    except OSError:
        return None   # no KFD here: let the ranks find out
    return n


def self_launch(args):
    """`--gpus N` with no launcher: start the N ranks as child processes (the parent has not touched a GPU
    and never will), relay rank 0's stdout, exit with the worst return code.  A rank that dies takes its
    siblings with it (they would otherwise sit in a rendezvous or a barrier until its timeout)."""
    import socket

    ndev = visible_gpu_count()
    if ndev is not None and ndev < args.gpus and not os.environ.get("RAYS_BENCH_SHARE_GPU"):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible "
                         "(RAYS_BENCH_SHARE_GPU=1 rehearses N ranks on fewer GPUs)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
        if any(rc not in (None, 0) for rc in rcs):     # one rank failed: end the others now
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    pr.terminate()
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = pr.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        rcs[i] = pr.wait()
            break
        time.sleep(0.2)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    raise SystemExit(max(abs(rc) for rc in rcs))


def kernel_source_hash():
    """sha256 over the sources librays_hip.so is built from (rays_amd/csrc + the C ABI header): profiles/counters.json
    records it when its counters are collected, and a bench run whose sources hash differently reports no counter-
    derived figures (the counters would describe another kernel)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rays_amd", "csrc")
    files = sorted(f for f in os.listdir(d) if f.endswith((".hpp", ".inc", ".hip")) or f == "Makefile")
    for f in files + [os.path.join("..", "..", "include", "rays_hip.h")]:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def numerics_evidence(cfg_name, flavour):
    """What tests/test_gpu_numerics_full_fans.py MEASURED for this workload and flavour on the full fan (every recorded
    step restarted from the oracle's point; every traced point against the oracle's), replayed from
    profiles/numerics_evidence.json -- or from gpurun_out/numerics_evidence.json, where that test leaves it on the box it
    ran on -- when it was collected from the kernel sources this run is built from."""
    keys = ("steps_restarted", "n_above_1e-10", "n_above_1e-11", "max_per_step", "median_per_step", "points_compared",
            "points_not_identical", "max_pointwise", "frac_points_above_1e-10", "rays_surveyed", "rays_with_other_counts", "kernel")
    stale = None
    for rel in (("profiles", "numerics_evidence.json"), ("gpurun_out", "numerics_evidence.json")):
        path = os.path.join(ROOT, *rel)
        if not os.path.exists(path):
            continue
        try:
            e = json.load(open(path)).get(f"{cfg_name}::{flavour}")
        except Exception as ex:
            print(f"[bench] {'/'.join(rel)} unreadable: {ex}", file=sys.stderr)
            continue
        if not e:
            continue
        if e.get("source_hash") != kernel_source_hash():
            stale = ("/".join(rel), e.get("source_hash"))
            continue
        out = {k: e.get(k) for k in keys}
        out["source"] = ("replayed from " + "/".join(rel) + " (tests/test_gpu_numerics_full_fans.py on MI355X, kernel "
                         "sources hash " + str(e.get("source_hash")) + "); not measured in this run")
        return out
    if stale:
        print(f"[bench] {stale[0]}: {cfg_name}::{flavour} was measured on other kernel sources ({stale[1]} != "
              f"{kernel_source_hash()}): numerics_evidence omitted -- re-run tests/test_gpu_numerics_full_fans.py on a GPU box "
              "and copy gpurun_out/numerics_evidence.json to profiles/", file=sys.stderr)
    return None


def numerics_text(flavour, is_tol_kernel, ev):
    if not (flavour == "tolerance" and is_tol_kernel):
        return flavour + ": bit-identical to the reference CPU path" if flavour == "exact" else \
            flavour + " requested; this configuration has no tolerance flavour: bit-identical to the reference CPU path"
    base = ("tolerance: NOT bit-identical (FMA contraction, re-association, once-refined reciprocals and roots); ray counts / "
            "step indices / stop flags exactly the reference's")
    if ev is None:
        return base + "; per-step and pointwise deviation of this build not measured (see value_exact for the bit-exact flavour)"
    where = " (one-GPU fan of this workload)" if ev.get("fan") else ""
    return (base + f" on all {ev['rays_surveyed']} rays surveyed{where}; restarted from every one of the oracle's {ev['steps_restarted']} "
            f"recorded points, {ev['n_above_1e-10']} steps land further than 1e-10 (relative, norm-wise on r and k) from the "
            f"reference's next point, max {ev['max_per_step']:.2e}; accumulated along the rays the traced fan deviates "
            f"pointwise by up to {ev['max_pointwise']:.1e} ({ev['frac_points_above_1e-10']:.1e} of the points above 1e-10) "
            "-- numerics_evidence; value_exact is the bit-identical flavour")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def metric_name(p, rays_per_gpu):
    eq = {0: "slab", 1: "Solovev", 2: "eqdsk"}.get(int(p.equilib_model), "?")
    n = f"{rays_per_gpu // 1024}k" if rays_per_gpu % 1024 == 0 else str(rays_per_gpu)
    return f"ray-steps/sec (whole node), {n}-ray {eq} fan"


def simd_idle_fraction(npoints, resident_lanes=N_SIMD * 64):
    """Share of SIMD-time without a wave, from the recorded step counts, when the fan fits the resident lanes
    (one wave per SIMD, nothing to refill): wave w holds rays 64w..64w+63 and lives as long as its longest ray."""
    n = len(npoints)
    if n == 0 or n > resident_lanes:
        return None
    steps = np.maximum(npoints.astype(np.int64) - 1, 0)
    pad = (-n) % 64
    per_wave = np.pad(steps, (0, pad)).reshape(-1, 64).max(axis=1)
    longest = per_wave.max()
    return None if longest == 0 else float(1.0 - per_wave.sum() / (N_SIMD * longest))


def build_fan(cfg_path, world, fan_scale=1, nstep_max=None):
    from rays_amd.namelist import read_namelist
    from rays_amd.params import params_from_namelist
    from rays_amd.ray_init import initialize_ray_init

    nml = read_namelist(cfg_path)
    world = world * fan_scale
    if nstep_max is not None:
        nml["ode_list"]["nstep_max"] = int(nstep_max)
    if world > 1:  # weak scaling: world x more launch angles in n_theta over the same range
        g = next(nml[k] for k in ("solovev_ray_init_nphi_ktheta_list",
                                   "axisym_toroid_ray_init_r_z_nphi_ntheta_list",
                                   "simple_slab_ray_init_list") if k in nml)
        if "n_rindex_theta" in g:
            g["n_rindex_theta"] = int(g["n_rindex_theta"]) * world
            g["delta_rindex_theta"] = float(g["delta_rindex_theta"]) / world
        else:
            g["n_ky_launch"] = int(g["n_ky_launch"]) * world
            g["delta_rindex_y0"] = float(g.get("delta_rindex_y0", 0.0)) / world
        nml["ray_init_list"]["nray_max"] = int(nml["ray_init_list"]["nray_max"]) * world
    from rays_amd.trace import load_axisym_tables

    tab = load_axisym_tables(cfg_path, nml)
    p = params_from_namelist(nml, tab)
    if tab is not None:
        from rays_amd import hip

        hip.set_axisym_tables(tab)
    r0, n0, w = initialize_ray_init(p, nml, tab)
    build_fan.ray_pwr_wt, build_fan.tables = w, tab   # for --exchange deposition
    return nml, p, r0, n0


def cpu_baseline(cfg_path, budget_s=20.0):
    """Reference CPU path on this host's cores, on a bounded sample of the same fan.

    kind "reference": oracle/_ref/rays_ref_dump (the reference RAYS_project hot path compiled from
    its own sources) run on the fan subsampled in launch angle, timing trace_rays only; the sample
    grows from 4x4 (4096 rays) towards the whole fan while it fits ~budget_s of wall time.  Falls back
    to the C restatement (kind "port") if the binary is not there."""
    from rays_amd.namelist import read_namelist

    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "rays_ref_dump")
    text0 = open(cfg_path).read()
    nml = read_namelist(cfg_path)
    import re

    def subsample(sub):
        def scale(txt, key_n, key_d):
            n = int(re.search(key_n + r"\s*=\s*(\d+)", txt).group(1))
            d = float(re.search(key_d + r"\s*=\s*([-\d.eE+]+)", txt).group(1))
            txt = re.sub(key_n + r"\s*=\s*\d+", f"{key_n} = {max(1, n // sub)}", txt)
            txt = re.sub(key_d + r"\s*=\s*[-\d.eE+]+", f"{key_d} = {d * sub!r}", txt)
            return txt

        if "n_rindex_theta" in text0:
            txt = scale(scale(text0, "n_rindex_theta", "delta_rindex_theta"), "n_rindex_phi", "delta_rindex_phi")
            what = "the whole fan" if sub == 1 else f"same fan subsampled {sub}x{sub} in launch angle"
        else:
            txt = scale(scale(text0, "n_ky_launch", "delta_rindex_y0"), "n_kz_launch", "delta_rindex_z0")
            what = "the whole fan" if sub == 1 else f"same slab fan subsampled {sub}x{sub} in launch index"
        return txt, what + ", all steps"

    def run_ref(text, threads=None):
        """one run of the reference binary; threads: /openmp_list/ num_threads (None = all)"""
        if threads is not None:
            text = re.sub(r"&openmp_list.*?/", "", text, flags=re.S | re.I) + f"\n &openmp_list\n num_threads = {threads}\n/\n"
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, "rays.in"), "w").write(text)
            for f in os.listdir(os.path.join(ROOT, "configs")):
                if f.endswith(".geqdsk"):
                    shutil.copy(os.path.join(ROOT, "configs", f), d)
            env = dict(os.environ, RAYS_DUMP_FILE="none")
            out = subprocess.run([ref], cwd=d, env=env, capture_output=True, text=True, timeout=600).stdout
        return dict(l.split("=")[0].split()[-1:] + [l.split("=")[1].strip()]
                    for l in out.splitlines() if l.startswith("RAYS_REF"))

    # the sample grows (4x4 -> 2x2 -> whole fan) while the next size is predicted to fit the budget
    text, sample = subsample(4)
    if os.path.exists(ref):
        best = None
        for sub in (4, 2, 1):
            text, sample = subsample(sub)
            try:
                vals = run_ref(text)
                wall = float(vals["trace_wall_s"])
                best = dict(value=float(vals["steps_per_s"]), unit="ray-steps/s",
                            cores=int(vals["threads"]), kind="reference", cpu_model=cpu_model(),
                            sample=sample + f" ({vals['nray']} rays); {vals['total_steps']} steps in {wall:.2f} s, "
                            "reference RAYS_project trace_rays (amdflang -O2 -fopenmp), OpenMP over rays")
            except Exception as e:  # keep what we have, or fall through to the port
                print(f"[bench] reference CPU baseline failed: {e}", file=sys.stderr)
                break
            if 4.0 * wall > budget_s:
                break
        if best is not None:
            try:  # one thread: a 16x16-subsampled fan (256 rays), a few seconds
                t1, s1 = subsample(16)
                v1 = run_ref(t1, threads=1)
                best["threads1"] = dict(value=float(v1["steps_per_s"]), unit="ray-steps/s", cores=1,
                                        sample=s1 + f" ({v1['nray']} rays); {v1['total_steps']} steps in "
                                        f"{float(v1['trace_wall_s']):.2f} s")
            except Exception as e:
                print(f"[bench] one-thread CPU baseline failed: {e}", file=sys.stderr)
            return best
        text, sample = subsample(4)
    from tests import oracle_lib
    from rays_amd.params import params_from_namelist
    from rays_amd.namelist import parse_namelist
    from rays_amd.ray_init import initialize_ray_init

    nml2 = parse_namelist(text)
    p = params_from_namelist(nml2)
    r0, n0, _ = initialize_ray_init(p, nml2)
    t0 = time.perf_counter()
    o = oracle_lib.trace(p, r0, n0, nthreads=cores)
    dt = time.perf_counter() - t0
    steps = int(np.maximum(o["npoints"].astype(np.int64) - 1, 0).sum())
    return dict(value=steps / dt, unit="ray-steps/s", cores=cores, kind="port", cpu_model=cpu_model(),
                sample=sample + f"; {steps} steps in {dt:.2f} s, C restatement (oracle/), OpenMP over rays")


def host_entry_time(p, r0, n0, reps=3):
    """What the reference's own call site sees (RAYS.f90:13 `call trace_rays` -> fortran/trace_rays_hip.f90 ->
    rays_hip_trace): host arrays in and out, blocking, the caller's arrays resident (allocated and zero-filled once by
    initialize_ray_results_m, as in the Fortran host).  PCIe-inclusive; never `value`."""
    from rays_amd import hip

    out = hip.trace_host(p, r0, n0, ngpu=1)        # first call: first touch of the caller's pages, buffer cache cold
    steps = int(np.maximum(out["npoints"].astype(np.int64) - 1, 0).sum())
    best, kern = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = hip.trace_host(p, r0, n0, ngpu=1, out=out)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return dict(ms_per_call=1e3 * best, value=steps / best, kernel=hip.kernel_name(p, len(r0)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "cfg3b_solovev64k_rk4.in"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-entry", action="store_true", help="skip the host_entry leg (rays_hip_trace on host arrays)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-streams leg (value_two_streams)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the trajectory gather (diagnostic)")
    ap.add_argument("--exchange", choices=("gather", "deposition"), default="gather",
                    help="what leaves the GPUs each pass: the packed trajectories, gathered to rank 0 "
                         "(default, the BASELINE metric), or -- eqdsk + damping configs only -- the "
                         "Ptotal(psiN) deposition profile, binned on the GPU and reduced over ranks in "
                         "ray order (SURVEY 8(f) f2: n_bins doubles per rank instead of the trajectories)")
    ap.add_argument("--exact-profile", action="store_true",
                    help="--exchange deposition, N>1: ray-ordered chain over ranks (bit-identical to one process)")
    ap.add_argument("--verify-gather", action="store_true",
                    help="N>1 diagnostic: rank 0 re-traces the whole fan and checks the gathered arrays")
    ap.add_argument("--fan-scale", type=int, default=1,
                    help="diagnostic: rays per GPU = 65536 x this (finer n_theta); not the headline config")
    ap.add_argument("--nstep-max", type=int, default=None, help="diagnostic: override nstep_max")
    ap.add_argument("--numerics", choices=("tolerance", "exact"), default="tolerance",
                    help="rays_hip_set_numerics: 'tolerance' = not bit-identical; every step within 1e-10 relative of the "
                         "reference's (measured 4e-15 on the headline fan: numerics_evidence), ray counts / step indices / stop "
                         "flags exactly the reference's; cold RK4 kernels | 'exact' = bit-identical to the reference CPU path.  "
                         "The line names what ran (config.kernel, config.numerics) and, on one GPU, carries the other flavour's "
                         "rate as value_exact")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)   # does not return

    import torch
    import torch.distributed as dist

    from rays_amd import hip
    from rays_amd.trace import DeviceTrace

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not os.environ.get("RAYS_BENCH_SHARE_GPU"):
        raise SystemExit(f"local rank {local_rank} has no GPU ({ndev} visible)")
    dev_index = local_rank % ndev  # RAYS_BENCH_SHARE_GPU=1: rehearsal of N ranks on fewer GPUs
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RAYS_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        dist.init_process_group(backend, device_id=dev if backend == "nccl" else None)

    hip.set_numerics(args.numerics)
    nml, p, r0, n0 = build_fan(args.config, world, args.fan_scale, args.nstep_max)
    nray_total = len(r0)
    from rays_amd.exchange import shard_bounds

    lo, hi = shard_bounds(nray_total, world, rank)
    tr = DeviceTrace(p, r0[lo:hi], n0[lo:hi], device=dev)
    nv, npt = p.nv, p.nstep_max + 1

    # first pass (also the zero-fill, as initialize_ray_results_m does) + workload size
    tr.launch(zero_fill=True)
    torch.cuda.synchronize()
    npts_local = tr.npoints.to(torch.int64)
    steps_local = int(torch.clamp(npts_local - 1, min=0).sum().item())
    points_local = int(npts_local.sum().item())

    # ---- multi-GPU exchange: packed trajectories, grouped send/recv to rank 0 (rays_amd/exchange.py)
    gather = None
    if world > 1 and not args.no_gather:
        from rays_amd.exchange import TrajectoryGather

        tg = TrajectoryGather(nray_total, nv, p.nstep_max, dev)
        tg.prepare(tr.npoints)

        def gather():
            # starts this pass's exchange and completes the previous one: the RCCL transfers of
            # pass i overlap the trace of pass i+1; barrier() below drains the last one
            tg.gather_async(tr.ray_vec, tr.residual, tr.npoints, tr.stop_code)

        # one untimed exchange of the first pass: RCCL builds its point-to-point connections on first
        # use, which must not land in the timed region when the caller asks for --warmup 0
        gather()
        tg.finish()

    deposit = None
    if args.exchange == "deposition":
        from rays_amd.exchange import ProfileChain

        gather = None
        n_bins = 100  # deposition_profiles_m.f90:53 default_n_bins
        power = torch.as_tensor(np.ascontiguousarray(build_fan.ray_pwr_wt[lo:hi]), dtype=torch.float64).to(dev)
        work = torch.zeros((n_bins, hi - lo), dtype=torch.float64, device=dev)
        chain = ProfileChain(n_bins, dev) if world > 1 else None
        prof = torch.zeros(n_bins, dtype=torch.float64, device=dev)

        def accumulate(carry, out):
            hip.deposition_device(p, "Ptotal_x" if p.equilib_model == 0 else "Ptotal_psi", n_bins, hi - lo, tr.ray_vec.data_ptr(), tr.npoints.data_ptr(),
                                  power.data_ptr(), work.data_ptr(), None if carry is None else carry.data_ptr(),
                                  out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)

        def deposit():
            if chain is None:
                accumulate(None, prof)
            elif args.exact_profile:
                chain.reduce(accumulate)            # bit-identical to one process; serialises the ranks
            else:
                chain.reduce_unordered(accumulate)  # per-rank ordered sums + one RCCL reduce

    def step(ev=None, with_exchange=True):
        if ev is not None:
            ev[0].record()
        tr.launch(zero_fill=False)
        if ev is not None:
            ev[1].record()
        if with_exchange and gather is not None:
            gather()
        if with_exchange and deposit is not None:
            deposit()

    def barrier():
        if gather is not None:
            tg.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, **kw):
        """n steps between two barrier + synchronize brackets; MAX over ranks of the wall time"""
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        barrier()
        t0 = time.perf_counter()
        for k in range(n):
            step(events[k], **kw)
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), float(np.mean([a.elapsed_time(b) for a, b in events]))

    for _ in range(args.warmup):
        step()
    # ---- THE timed region of the contract: exactly K steps (trace + this pass's exchange) -------------------
    elapsed, kernel_ms = timed(args.steps)

    # ---- diagnosis of an N > 1 line: where does the time go? (outside the timed region above) ---------------
    ones = torch.ones(1, dtype=torch.int64, device=dev)
    tot = torch.tensor([steps_local], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)   # RCCL: how many ranks really took part
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    n_ranks_seen = int(ones.item())
    total_steps = int(tot.item())
    if n_ranks_seen != args.gpus:   # never report an N-GPU line that fewer ranks produced
        raise SystemExit(f"bench.py --gpus {args.gpus}: the collective saw {n_ranks_seen} rank(s)")
    split = {}
    if world > 1 and gather is not None:
        el_trace, _ = timed(args.steps, with_exchange=False)        # the same K passes without the exchange
        reps = 3
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):                                       # exchange alone: pack + RCCL send/recv + unpack,
            tg.gather(tr.ray_vec, tr.residual, tr.npoints, tr.stop_code)   # completed before the next starts
            torch.cuda.synchronize()
        barrier()
        g = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
        gather_s = float(g.item())
        pts = torch.tensor([points_local], dtype=torch.int64, device=dev)
        dist.all_reduce(pts, op=dist.ReduceOp.MAX)
        peer_bytes = 8.0 * (nv + 1) * int(pts.item()) + 8.0 * (hi - lo)   # packed points + npoints/stop codes
        split = dict(value_trace_only=total_steps / (el_trace / args.steps),
                     trace_only_ms_per_step=1e3 * el_trace / args.steps,
                     gather_ms=1e3 * gather_s, gather_bytes_per_peer=peer_bytes,
                     gather_GBps_per_peer=peer_bytes / gather_s / 1e9,
                     gather_GBps_into_root=peer_bytes * (world - 1) / gather_s / 1e9,
                     scaling_bound="root ingest: every pass each of the N-1 peers sends its packed trajectories (gather_bytes_per_peer) "
                                   "to rank 0 over its own xGMI link (~77 GB/s per direction), so ms_per_step >= max(trace, "
                                   "gather_bytes_per_peer / link rate) for every N >= 2 and value levels off near "
                                   "N x recorded_steps / that time (DESIGN.md 5); value_trace_only is what the kernels scale to",
                     modelled_gather_floor_ms=1e3 * peer_bytes / 77e9,
                     gather_note="value = trace + gather as the metric is defined (the exchange of pass i runs behind the "
                          "trace of pass i+1, so ms_per_step ~ max(trace, gather)); value_trace_only = the same K "
                          "passes with the exchange switched off; gather_ms = one exchange on its own")

    # ---- N > 1: the same K passes when every rank's result STAYS on its GPU (a consumer with device-side
    # post-processing) and only ray_results_m's per-ray summaries reach rank 0 (outside the timed region) ----
    if world > 1 and args.exchange == "gather":
        from rays_amd.exchange import SummaryGather

        sg = SummaryGather(nray_total, nv, dev)
        summ = lambda: sg.gather(tr.npoints, tr.stop_code, tr.end_ray_vec, tr.end_residuals, tr.max_residuals)
        summ()
        saved_gather, gather = gather, None
        deposit_saved, deposit = deposit, summ
        el_s, _ = timed(args.steps)
        gather, deposit = saved_gather, deposit_saved
        split.update(value_sharded_result=total_steps / (el_s / args.steps), sharded_result_ms_per_step=1e3 * el_s / args.steps,
                     sharded_result_bytes_per_ray=sg.bytes_per_ray(),
                     sharded_result_note="the same K passes; each rank's trajectories stay in its HBM, rank 0 receives npoints, "
                                         "stop code, end_ray_vec, end_residuals, max_residuals of every ray (one grouped RCCL "
                                         "send/recv per pass): the mode a device-side consumer of the trajectories runs "
                                         "(deposition profiles, --exchange deposition); `value` above is trace + full gather as "
                                         "the metric is defined and is bound by the root's ingest")

    # the timed passes must have reproduced the first pass (deterministic kernels)
    assert int(torch.clamp(tr.npoints.to(torch.int64) - 1, min=0).sum().item()) == steps_local

    # ---- the other numerics flavour of the same kernel, same K passes (one GPU; outside the timed region) ----
    other = None
    if world == 1 and args.exchange == "gather":
        this_kernel = hip.kernel_name(p, hi - lo)
        hip.set_numerics("exact" if args.numerics == "tolerance" else "tolerance")
        if hip.kernel_name(p, hi - lo) != this_kernel:   # (configurations without a tolerance flavour run one kernel)
            npts_before = tr.npoints.clone()
            step()
            el_o, k_o = timed(args.steps)
            same_counts = bool(torch.equal(npts_before, tr.npoints))
            tag = "exact" if args.numerics == "tolerance" else "tolerance"
            other = {f"value_{tag}": total_steps / (el_o / args.steps), f"ms_per_step_{tag}": 1e3 * el_o / args.steps,
                     f"kernel_{tag}": hip.kernel_name(p, hi - lo), "flavours_agree_on_npoints": same_counts}
            if not same_counts:   # a tolerance kernel whose ray counts differ from the exact one's has no claim to the metric
                raise SystemExit("bench.py: the tolerance and the exact flavour disagree on npoints for this fan -- no line")
        hip.set_numerics(args.numerics)
        tr.launch(zero_fill=False)     # leave the arrays as the timed flavour wrote them
        torch.cuda.synchronize()

    # ---- two independent passes in flight (one GPU; outside the timed region) ------------------------------------------
    # One 64k fan is one ray per resident lane and the pass lasts as long as its longest ray: 55 % of the SIMD-time is
    # idle (simd_idle_frac).  A host with INDEPENDENT fans to trace (a parameter scan, a time loop over launch conditions)
    # can fill it: the same K passes issued alternately on two streams into two complete sets of result arrays, so that
    # the blocks of pass i + 1 start on the CUs whose blocks of pass i have ended.  Reported next to `value`, never as it:
    # `value` is K passes one after the other on one stream.
    pipelined = None
    if world == 1 and args.exchange == "gather" and not args.no_pipelined and 16.0 * (nv + 1) * npt * (hi - lo) <= 40e9:
        tr2 = DeviceTrace(p, r0[lo:hi], n0[lo:hi], device=dev)
        s_a, s_b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s_b):
            tr2.launch(zero_fill=True)
        torch.cuda.synchronize()
        def both(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                with torch.cuda.stream(s_a if k % 2 == 0 else s_b):
                    (tr if k % 2 == 0 else tr2).launch(zero_fill=False)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        both(2)
        el_p = both(args.steps)
        same = bool(torch.equal(tr.npoints, tr2.npoints) and torch.equal(tr.ray_vec, tr2.ray_vec) and
                    torch.equal(tr.residual, tr2.residual) and torch.equal(tr.stop_code, tr2.stop_code))
        if not same:
            raise SystemExit("bench.py: the two result sets of the two-stream passes differ")
        pipelined = {"value_two_streams": total_steps / (el_p / args.steps), "ms_per_step_two_streams": 1e3 * el_p / args.steps,
                     "two_streams_note": "the same K passes issued alternately on two HIP streams into two complete sets of result "
                                         "arrays (identical, checked): what a host with independent fans gets out of the SIMD-time one "
                                         "64k fan leaves idle (simd_idle_frac); `value` is K passes one after the other"}
        del tr2
        torch.cuda.empty_cache()

    if rank == 0 and gather is not None and args.verify_gather:
        # rank 0 re-traces every rank's block (block by block: the same kernel build each rank dispatched -- the
        # tolerance flavour's one-wave and two-waves builds are different compilations of the same arithmetic)
        ok = True
        for r in range(world):
            b0, b1 = shard_bounds(nray_total, world, r)
            blk = DeviceTrace(p, r0[b0:b1], n0[b0:b1], device=dev)
            blk.launch()
            torch.cuda.synchronize()
            ok = ok and (torch.equal(blk.ray_vec, tg.ray_vec[b0:b1]) and torch.equal(blk.residual, tg.residual[b0:b1])
                         and torch.equal(blk.npoints, tg.npoints[b0:b1]) and torch.equal(blk.stop_code, tg.stop_code[b0:b1]))
            del blk
        print(f"[bench] gather verification: {'OK' if ok else 'MISMATCH'}", file=sys.stderr)
        if not ok:
            raise SystemExit("gathered trajectories differ from a single-GPU trace of the same blocks")
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_steps / (elapsed / args.steps)
        bytes_per_launch = 8.0 * (nv + 1) * steps_local  # 8*(nv+1) B per recorded step
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        # counters of the same command collected by tools/profile_bench.sh (rocprofv3 --pmc, separate passes):
        # profiles/counters.json = {config file name: {hbm_bytes_per_launch, fp64_wave_insts: {...}, ...}}
        traffic, fp64 = None, None
        cpath = os.path.join(ROOT, "profiles", "counters.json")
        kname = hip.kernel_name(p, hi - lo)
        if os.path.exists(cpath) and args.fan_scale == 1 and args.nstep_max is None:
            try:
                db = json.load(open(cpath))
                c = db.get(os.path.basename(args.config) + "::" + kname)
                src_now = kernel_source_hash()
                if c and c.get("source_hash") != src_now:
                    print(f"[bench] profiles/counters.json: counters of {kname} were collected from other kernel sources "
                          f"({c.get('source_hash')} != {src_now}): roofline.traffic / roofline_fp64 omitted -- re-run "
                          "tools/profile_bench.sh + tools/update_counters.py", file=sys.stderr)
                    c = None
                if c and c.get("kernel") == kname:
                    traffic = c.get("hbm_bytes_per_launch")
                    w = c.get("fp64_wave_insts")
                    if w:
                        insts = sum(w.values())
                        flop = 64.0 * (w.get("add", 0) + w.get("mul", 0) + w.get("trans", 0)) + 128.0 * w.get("fma", 0)
                        clk = c.get("effective_clock_GHz") or 2.4
                        fp64 = {"fp64_wave_insts": insts, "by_kind": w,
                                "issue_frac": insts * FP64_CLK_PER_WAVE_INST / (N_SIMD * kernel_ms * 1e-3 * clk * 1e9),
                                "tflops": flop / (kernel_ms * 1e-3) / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                                "frac": flop / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                "valu_insts": c.get("valu_insts"), "clock_GHz": clk, "source": c.get("source"),
                                "note": "issue_frac = FP64 wave-instructions x 4 clk / (1024 SIMDs x kernel time x clock): "
                                        "the share of all FP64 issue slots used; tflops counts add/mul/trans as 1 flop "
                                        "and fma as 2 per lane (the exact kernels keep FMA contraction off for bit-parity with the "
                                        "reference; the tolerance flavour lets the compiler fuse)"}
            except Exception as e:
                print(f"[bench] profiles/counters.json unreadable: {e}", file=sys.stderr)
        idle = simd_idle_fraction(tr.npoints.cpu().numpy()) if world == 1 else None
        is_tol_kernel = "<" in kname and bool(int(kname.split("<")[1].split(",")[0]) & 16)
        evidence = None
        if args.fan_scale == 1 and args.nstep_max is None:
            evidence = numerics_evidence(os.path.basename(args.config), "tolerance" if is_tol_kernel else "exact")
            if evidence is not None and world > 1:
                evidence["fan"] = ("measured on the one-GPU fan of this workload; the N-GPU fan is the same fan widened in "
                                   "n_theta (weak scaling), traced by the same kernel")
        line = {
            "metric": metric_name(p, hi - lo),
            "value": value, "unit": "ray-steps/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": os.path.basename(args.config),
                       "rays_per_gpu": hi - lo, "rays_total": nray_total,
                       "recorded_steps_per_pass": total_steps, "nstep_max": p.nstep_max,
                       "ode": "RK4_ODE" if p.ode_solver == 0 else "SG_ODE",
                       "deriv": "cold" if p.ray_deriv == 0 else "numerical",
                       "kernel": hip.kernel_name(p, hi - lo),
                       "numerics": numerics_text(hip.get_numerics(), is_tol_kernel, evidence),
                       "exchange": ("none" if world == 1 else
                                    ("skipped" if args.no_gather else "packed send/recv to rank 0 (RCCL)"))
                       if args.exchange == "gather" else
                       "deposition profile Ptotal(psiN), 100 bins, binned on the GPU"
                       + ("" if world == 1 else (", ray-ordered chain over ranks (RCCL send/recv)" if args.exact_profile
                                                 else ", per-rank sums + RCCL reduce to rank 0"))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "counters": (None if traffic is None else
                                      "traffic / roofline_fp64 counters replayed from profiles/counters.json (rocprofv3 --pmc passes of "
                                      "this command on these kernel sources); achieved, kernel_ms and frac are measured in this run"),
                         "note": "the contract's HBM figure; the path is bound by FP64 instruction issue "
                                 "(roofline_fp64), not by HBM: DESIGN.md 4.5"
                                 + ("; kernel_ms spans the launch = this kernel + rk4_resume_kernel behind it (the exact "
                                    "twin's continuation of the handed-over rays, ~0.05 ms: DESIGN.md 4.6)" if is_tol_kernel else "")},
        }
        if fp64 is not None:
            line["roofline_fp64"] = fp64
        if idle is not None:
            line["simd_idle_frac"] = idle
        if evidence is not None:
            line["numerics_evidence"] = evidence
        line.update(split)
        if other is not None:
            line.update(other)
        if pipelined is not None:
            line.update(pipelined)
        host_bytes = 8.0 * (nv + 1) * (p.nstep_max + 1) * nray_total   # the caller's padded ray_results_m arrays
        if world == 1 and not args.no_host_entry and args.exchange == "gather" and host_bytes <= 8e9:
            try:   # outside the timed region: the rate the reference's own call site sees, both flavours
                del tr
                torch.cuda.empty_cache()
                he = host_entry_time(p, r0, n0)
                other_flavour = "exact" if args.numerics == "tolerance" else "tolerance"
                hip.set_numerics(other_flavour)
                if hip.kernel_name(p, len(r0)) != he["kernel"]:
                    he[other_flavour] = host_entry_time(p, r0, n0)
                hip.set_numerics(args.numerics)
                he["note"] = ("rays_hip_trace (the entry fortran/trace_rays_hip.f90 calls): pack on the GPU, packed D2H through "
                              "pinned staging, host scatter into the caller's resident arrays; PCIe-inclusive, best of 3 calls")
                line["host_entry"] = he
            except Exception as e:
                print(f"[bench] host_entry leg failed: {e}", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
