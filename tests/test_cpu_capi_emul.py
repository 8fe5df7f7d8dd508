"""CPU tier: the C ABI's host logic with SEVERAL devices.  The GPU pool hands out one GPU per call, so the
multi-device branches of librays_hip.so -- rays_hip_trace over distinct devices, and rays_hip_trace_gather's
RCCL phase (ncclCommInitAll, one grouped batch of ncclSend / ncclRecv, per-peer offsets, unpack into slabs;
rays_amd/csrc/rays_gather.inc) -- cannot run on hardware there.  Here rays_capi.hip itself is compiled for the
host against an emulated HIP runtime (tests/hip_emul/hip/hip_runtime_api_emul.h: four devices; streams and memory
belong to a device and fail when used under another) and a stand-in for librccl.so (tests/hip_emul/fake_rccl.cpp),
and tests/capi_emul_driver.py drives it through rays_amd/hip.py in a process of its own: G = 1..4 devices in
several orders, ragged last blocks, empty blocks, cache slots that move between devices, the SG
workspace and the eqdsk tables per device; every result bit-identical to the oracle.  Once plain, once under
ASan + UBSan."""
import os
import subprocess
import sys

import pytest

from tests.common import ROOT

EMUL = os.path.join(ROOT, "tests", "hip_emul")


def _run(san):
    subprocess.check_call(["make", "-s", "-f", "Makefile.capi", "-j4"] + (["SAN=1"] if san else []), cwd=EMUL)
    b = os.path.join(EMUL, "build_capi_san" if san else "build_capi")
    env = dict(os.environ, RAYS_HIP_LIB=os.path.join(b, "librays_capi_emul.so"),
               RAYS_HIP_RCCL_LIB=os.path.join(b, "librccl_fake.so"), RAYS_EMUL_DEVICES="4")
    if san:
        libs = [subprocess.check_output(["gcc", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libubsan.so")]
        if not all(os.path.isabs(x) and os.path.exists(x) for x in libs):
            pytest.skip("libasan / libubsan not installed")
        env.update(LD_PRELOAD=":".join(libs), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "capi_emul_driver.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "capi emulation ok" in r.stdout, r.stdout[-4000:]
    return r.stdout


def test_multi_device_entries_on_the_emulated_runtime():
    out = _run(san=False)
    assert "'wrong_device': 0" in out


def test_multi_device_entries_under_asan_ubsan():
    out = _run(san=True)
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
