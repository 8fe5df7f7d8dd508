"""Scheduling model of the one-ray-per-lane RK4 kernel (rays_rk4_body.inc): waves of 64 lanes pull rays from one
counter; a trip advances every lane under way by one RK4 stage; a pass over the parked lanes (ray ends + starts) costs
the wave E trips; it fires when the accumulated idle lane-trips reach C (C = 0: next stage-3 trip), at once when no lane
is under way.  Input: gpurun_out/npoints_<config>.npz (tools/dump_npoints.py).  Prints the pass length in trips for the
slowest wave and the lane utilisation; and, lane by lane, the pass length for the hand-out orders of rays_trace.hpp:
take_rays (developer model; times are measured on the GPU, this explains them)."""
import sys
import heapq
import numpy as np


def simulate(steps, nwaves, C, E, trip_cost=1.0):
    """Event-driven over waves (each wave advances independently; the shared counter couples them through time)."""
    nray = len(steps)
    nxt = [0]
    # per wave state
    def take():
        i = nxt[0]
        if i < nray:
            nxt[0] += 1
            return i
        return -1
    # time-ordered simulation: process waves in order of their clocks so that the counter is shared fairly
    waves = []
    for w in range(nwaves):
        rem = np.zeros(64, np.int64)  # remaining trips per lane (0 = no ray)
        for l in range(64):
            i = take()
            if i >= 0:
                rem[l] = 4 * steps[i] + 1   # initial check + 4 stages per step
        waves.append(dict(t=E * trip_cost, rem=rem, parked=np.zeros(64, bool), idle=0, useful=0, trips=0, events=1))
    heap = [(wv["t"], k) for k, wv in enumerate(waves)]
    heapq.heapify(heap)
    dry = False
    tend = 0.0
    while heap:
        t, k = heapq.heappop(heap)
        wv = waves[k]
        rem, parked = wv["rem"], wv["parked"]
        alive = rem > 0
        if not alive.any() and not parked.any():
            tend = max(tend, t)
            continue
        # advance until next decision point: run trips until a lane ends or fire condition
        fire = False
        if parked.any():
            if not alive.any():
                fire = True
            elif not dry:
                if wv["idle"] >= C and (wv["trips"] % 4 == 0):
                    fire = True
        if fire:
            wv["events"] += 1
            wv["idle"] = 0
            for l in np.nonzero(parked)[0]:
                parked[l] = False
                if not dry:
                    i = take()
                    if i >= 0:
                        rem[l] = 4 * steps[i] + 1
                    else:
                        dry = True
            wv["t"] = t + E * trip_cost
            heapq.heappush(heap, (wv["t"], k))
            continue
        # one trip (or a run of trips until something changes)
        n = int(rem[alive].min())
        if parked.any() and not dry:
            # limit run so that the fire condition is re-evaluated: trips until idle >= C and stage aligned
            need = max(0, C - wv["idle"])
            np_ = int(parked.sum())
            m = -(-need // np_) if need > 0 else 0
            m = max(1, m)
            # align to next multiple of 4 trips
            tt = wv["trips"] + m
            tt = -(-tt // 4) * 4
            n = min(n, tt - wv["trips"]) if tt > wv["trips"] else min(n, 1)
        n = max(1, n)
        rem[alive] -= n
        wv["trips"] += n
        wv["useful"] += int(alive.sum()) * n
        wv["idle"] += int(parked.sum()) * n
        ended = alive & (rem == 0)
        parked |= ended
        wv["t"] = t + n * trip_cost
        heapq.heappush(heap, (wv["t"], k))
    useful = sum(w["useful"] for w in waves)
    trips = np.array([w["trips"] + E * w["events"] for w in waves])
    return tend, useful / (64.0 * trips.sum()), np.mean([w["events"] for w in waves]), trips.max(), trips.mean()



# ---- hand-out order (lane-level model: every lane on its own clock, a ray costs 4 steps + 1 trips) ----------------------
def order_index(steps, lanes):
    """Rays handed out in the given order by one counter; returns the pass length in trips."""
    work = 4 * steps + 1
    heap, pos, t_end = [], 0, 0
    for l in range(min(lanes, len(steps))):
        heapq.heappush(heap, (int(work[pos]), l)); pos += 1
    while heap:
        t, l = heapq.heappop(heap); t_end = max(t_end, t)
        if pos < len(steps):
            heapq.heappush(heap, (t + int(work[pos]), l)); pos += 1
    return t_end


def order_pilots(steps, lanes, S=4, BK=64, probes=384):
    """rays_trace.hpp: take_rays -- blocks of BK * S rays, row 0 = pilots, then sweep 1 over the neighbour slots whose
    pilot is still running, then sweep 2 over what is left."""
    n = len(steps); work = 4 * steps + 1
    M = (n + BK * S - 1) // (BK * S) * BK
    Q = M * (S - 1)
    done = [False] * M; taken = set()
    ppos = [0]; sw = [0, 0]

    def slot(q):
        blk, l = divmod(q, BK); B, k = divmod(blk, S - 1)
        return B * BK + l, (B * S + k + 1) * BK + l

    def take():
        while ppos[0] < M:
            m = ppos[0]; ppos[0] += 1
            r = (m // BK) * BK * S + m % BK
            if r < n: return r
        tries = 0
        while sw[0] < Q and tries < probes:
            q = sw[0]; sw[0] += 1; tries += 1
            m, r = slot(q)
            if r < n and not done[m]:
                taken.add(q); return r
        if sw[0] < Q: return -2
        while sw[1] < Q:
            q = sw[1]; sw[1] += 1
            m, r = slot(q)
            if r < n and q not in taken: return r
        return -1

    heap, cur, hungry, t_end = [], {}, [], 0
    for l in range(lanes):
        r = take()
        if r >= 0: cur[l] = r; heapq.heappush(heap, (int(work[r]), l))
        elif r == -2: hungry.append(l)
    while heap:
        t, l = heapq.heappop(heap); t_end = max(t_end, t)
        r = cur[l]
        if (r // BK) % S == 0: done[r // (BK * S) * BK + r % BK] = True
        hs, hungry = hungry, []
        for l2 in [l] + hs:
            r2 = take()
            if r2 >= 0: cur[l2] = r2; heapq.heappush(heap, (t + int(work[r2]), l2))
            elif r2 == -2: hungry.append(l2)
    return t_end


if __name__ == "__main__":
    f = sys.argv[1]
    nwaves = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    npt = np.load(f)["npoints"].astype(np.int64)
    steps = np.maximum(npt - 1, 0)
    ideal = (4 * steps + 1).sum() / (64.0 * nwaves)
    print(f"{f}: rays {len(steps)} mean steps {steps.mean():.1f} max {steps.max()}  ideal {ideal:.0f} trips per wave")
    print("hand-out order (lane-level model, pass length in trips):")
    print(f"  index order                          {order_index(steps, 64 * nwaves)}")
    print(f"  sorted by true length (not knowable) {order_index(steps[np.argsort(-steps, kind='stable')], 64 * nwaves)}")
    for S in (2, 4, 8):
        print(f"  pilots + two sweeps, S = {S}            {order_pilots(steps, 64 * nwaves, S)}")
    print("batched ray-end passes (wave-level model, index order):")
    for E in (3, 5, 8):
        for C in (0, 64, 128, 256, 512, 1024):
            tend, util, ev, tmax, tmean = simulate(steps, nwaves, C, E)
            print(f"  E={E} C={C:5d}: pass {tend:8.0f} trips ({tend / ideal:.2f}x ideal)  lane util {util:.3f}  events/wave {ev:.0f}")
