"""Scheduling model of the one-ray-per-lane RK4 kernel (rays_rk4_body.inc): waves of 64 lanes pull rays from one
counter; a trip advances every lane under way by one RK4 stage; a pass over the parked lanes (ray ends + starts) costs
the wave E trips; it fires when the accumulated idle lane-trips reach C (C = 0: next stage-3 trip), at once when no lane
is under way.  Input: gpurun_out/npoints_<config>.npz (tools/dump_npoints.py).  Prints the pass length in trips for the
slowest wave and the lane utilisation (developer model; times are measured on the GPU, this explains them)."""
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


if __name__ == "__main__":
    f = sys.argv[1]
    nwaves = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    npt = np.load(f)["npoints"].astype(np.int64)
    steps = np.maximum(npt - 1, 0)
    ideal = (4 * steps + 1).sum() / (64.0 * nwaves)
    print(f"{f}: rays {len(steps)} mean steps {steps.mean():.1f} max {steps.max()}  ideal {ideal:.0f} trips per wave")
    for E in (3, 5, 8):
        for C in (0, 64, 128, 256, 512, 1024):
            tend, util, ev, tmax, tmean = simulate(steps, nwaves, C, E)
            print(f"  E={E} C={C:5d}: pass {tend:8.0f} trips ({tend / ideal:.2f}x ideal)  lane util {util:.3f}  events/wave {ev:.0f}")
