"""TEST INFRASTRUCTURE ONLY: drives the product's C ABI compiled for the host (tests/hip_emul/emul_capi.cpp: emulated
HIP runtime with several devices + the stand-in for RCCL) through rays_amd/hip.py, in a process of its own:

    RAYS_HIP_LIB=tests/hip_emul/build_capi/librays_capi_emul.so RAYS_HIP_RCCL_LIB=.../librccl_fake.so \
      RAYS_EMUL_DEVICES=4 python tests/capi_emul_driver.py

(tests/test_cpu_capi_emul.py starts it, once plain and once under ASan + UBSan.)  Every result is compared with the
CPU oracle's trace of the same rays, bit for bit."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rays_amd import hip                     # noqa: E402
from rays_amd.params import copy_params      # noqa: E402
from tests import oracle_lib                 # noqa: E402
from tests.common import load_golden         # noqa: E402

ARRAYS = ("npoints", "stop_code", "ray_vec", "residual", "end_ray_vec", "end_residuals", "max_residuals")
assert "capi_emul" in hip.LIB_PATH, "this driver is for the emulated build only (RAYS_HIP_LIB)"
lib = hip.load()
rccl = C.CDLL(os.environ["RAYS_HIP_RCCL_LIB"])


def stats():
    a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
    lib.rays_emul_runtime_stats(C.byref(a), C.byref(b), C.byref(c))
    g, p, n = C.c_longlong(), C.c_longlong(), C.c_longlong()
    rccl.fake_rccl_stats(C.byref(g), C.byref(p), C.byref(n))
    return dict(launches=a.value, wrong_device=b.value, live=c.value, groups=g.value, pairs=p.value, bytes=n.value)


def same(out, ora, n=None, what=""):
    for k in ARRAYS:
        a, b = out[k], ora[k]
        if n is not None:
            a, b = a[:n], b[:n]
        np.testing.assert_array_equal(a, b, err_msg=f"{what}: {k}")


def sliced(ora, n):
    return {k: ora[k][:n] for k in ARRAYS}


assert lib.rays_hip_device_count() == int(os.environ.get("RAYS_EMUL_DEVICES", "4")) >= 4

# ---- RK4, Solovev fan (cfg 2): rays of ragged lengths, some stop at once ------------------------------------------------
g, nml, p = load_golden("cfg2_solovev1024_rk4")
q = copy_params(p)
q.nstep_max = 125
r0, n0 = g["rvec0_full"][:301].copy(), g["rindex_vec0_full"][:301].copy()
r0[17, 0] = 10.0      # launched outside the box: npoints = 1
n0[40] *= 3.0         # off the dispersion surface: stops at the initial check
ora = oracle_lib.trace(q, r0, n0)
assert ora["npoints"].min() == 1 and ora["npoints"].max() == 126 and len(np.unique(ora["npoints"])) > 5

# one device: the root traces into the global arrays, no exchange
hip.init_devices([0])
res, out = hip.trace_gather(q, r0, n0)
same(out, ora, what="gather G=1")
s0 = stats()
assert s0["groups"] == 0

# several distinct devices: blocks per device, grouped send / recv to the root, unpack (rays_gather.inc phase 2)
groups = 0
for devs in ([0, 1], [0, 1, 2], [0, 1, 2, 3], [3, 1], [2, 3, 0]):
    hip.init_devices(devs)
    for n in (301, 300, len(devs) + 1, 2, 1):     # ragged last block; nray < G: empty blocks
        res, out = hip.trace_gather(q, r0[:n], n0[:n])
        assert res.device == devs[0] and res.nray == n
        same(out, sliced(ora, n), what=f"gather devices {devs}, {n} rays")
        groups += 1
s1 = stats()
assert s1["groups"] - s0["groups"] == groups and s1["pairs"] > s0["pairs"] and s1["bytes"] > 0, (s0, s1)
assert s1["wrong_device"] == 0, s1

# four slots on device 0 (what rays_hip_trace picks for a large fan) leave device 0's streams and idle
# blocks in cache slots 1..3; a gather over devices 0..3 then has to claim those slots for devices 1..3
hip.init_devices([0, 0, 0, 0])
out = hip.trace_host(q, r0, n0, ngpu=None)
same(out, ora, what="host entry, four slots on device 0")
hip.init_devices([0, 1, 2, 3])
res, out = hip.trace_gather(q, r0, n0)
same(out, ora, what="gather after the slots served device 0")
hip.init_devices([1, 1, 2])            # and back: slots move between devices in rays_hip_trace too
out = hip.trace_host(q, r0, n0, ngpu=None)
same(out, ora, what="host entry, slots [1, 1, 2]")
assert stats()["wrong_device"] == 0, stats()

# rays_hip_trace over distinct devices: every device copies its slab straight into the caller's arrays
for devs in ([0, 1], [2, 0, 3], [0, 1, 2, 3]):
    hip.init_devices(devs)
    for n in (301, 7, 3):
        out = hip.trace_host(q, r0[:n], n0[:n], ngpu=None)
        same(out, sliced(ora, n), what=f"host entry devices {devs}, {n} rays")
with np.testing.assert_raises(hip.RaysHipError):
    hip.init_devices([0, 0])
    hip.trace_gather(q, r0, n0)       # RCCL: one rank per device
hip.init_devices([0, 1, 2, 3])

# ---- SG (per-(device, stream) workspace) and the eqdsk equilibrium + damping (tables uploaded per device) ---------------
for name in ("gold_solovev64_sg_cold", "gold_axisym64_eqdsk_damp_rk4"):
    g, nml, p = load_golden(name)      # (hands the eqdsk tables to hip -- the emulated library here -- and the oracle)
    q = copy_params(p)
    q.nstep_max = min(p.nstep_max, 25)
    r0, n0 = g["rvec0_full"][:37], g["rindex_vec0_full"][:37]
    ora = oracle_lib.trace(q, r0, n0)
    for devs in ([0], [1, 0], [0, 1, 2, 3]):
        hip.init_devices(devs)
        res, out = hip.trace_gather(q, r0, n0)
        same(out, ora, what=f"{name}: gather devices {devs}")
        out = hip.trace_host(q, r0, n0, ngpu=None)
        same(out, ora, what=f"{name}: host entry devices {devs}")

s = stats()
assert s["wrong_device"] == 0, s
lib.rays_hip_finalize()
print("capi emulation ok:", s, "live allocations after finalize:", stats()["live"])
