"""CPU tier: the library's RCCL binding (rays_amd/csrc/rays_gather.inc: `dlopen` + seven `dlsym`s + two hard-coded
`ncclDataType_t` values) against the image's REAL librccl.so and rccl.h -- the multi-device code has only ever run
against tests/hip_emul's stand-in (one GPU per box on the pool), so this pins what the stand-in cannot: that the names
resolve in the real library, that the enum values are the header's, and that the argument lists the function pointers
are declared with are the header's.  (Reference semantics of the exchange: ray_tracing.f90:62-64)"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GATHER = os.path.join(ROOT, "rays_amd", "csrc", "rays_gather.inc")
RCCL_H = "/opt/rocm/include/rccl/rccl.h"
RCCL_SO = "/opt/rocm/lib/librccl.so.1"

pytestmark = pytest.mark.skipif(not (os.path.exists(RCCL_H) and os.path.exists(RCCL_SO)), reason="no RCCL in this image")


def _bound_symbols():
    src = open(GATHER).read()
    return re.findall(r'RCCL_SYM\(\s*\w+\s*,\s*"(\w+)"\s*\)', src)


def _enum_values(header):
    """name -> value of every `name = <int>` enumerator of rccl.h"""
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(nccl\w+)\s*=\s*(\d+)\s*[,}]", header)}


def test_the_seven_symbols_resolve_in_the_real_library():
    syms = _bound_symbols()
    assert sorted(syms) == sorted(["ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend",
                                   "ncclRecv", "ncclGetErrorString"]), syms
    # in a child process: librccl.so pulls the HIP runtime in (0.5 GB of code objects), which the test session does not need
    code = ("import ctypes, sys\n"
            f"l = ctypes.CDLL({RCCL_SO!r}, mode=ctypes.RTLD_LOCAL)\n"
            f"missing = [s for s in {syms!r} if not hasattr(l, s)]\n"
            "l.ncclGetErrorString.restype = ctypes.c_char_p\n"
            "print('missing', missing, 'err0', l.ncclGetErrorString(0))\n"
            "sys.exit(1 if missing else 0)\n")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "missing []" in r.stdout and "no error" in r.stdout.lower(), r.stdout[-500:]


def test_hard_coded_datatypes_are_the_headers():
    src = open(GATHER).read()
    m = re.search(r"constexpr int kNcclInt32\s*=\s*(\d+)\s*,\s*kNcclDouble\s*=\s*(\d+)\s*;", src)
    assert m, "rays_gather.inc no longer declares kNcclInt32 / kNcclDouble"
    enum = _enum_values(open(RCCL_H).read())
    assert int(m.group(1)) == enum["ncclInt32"] == enum["ncclInt"]
    assert int(m.group(2)) == enum["ncclFloat64"] == enum["ncclDouble"]
    assert enum["ncclSuccess"] == 0      # RCCL_TRY treats every non-zero return as an error


def _decl(header, name):
    m = re.search(r"ncclResult_t\s+" + name + r"\s*\(([^;]*?)\)\s*;", header, flags=re.S)
    assert m, name
    return [re.sub(r"\s+", " ", a.strip()) for a in m.group(1).split(",")]


def test_function_pointer_signatures_follow_the_header():
    h = open(RCCL_H).read()
    types = lambda args: [re.sub(r"\s*\w+$", "", a).replace(" ", "") for a in args]
    assert types(_decl(h, "ncclSend")) == ["constvoid*", "size_t", "ncclDataType_t", "int", "ncclComm_t", "hipStream_t"]
    assert types(_decl(h, "ncclRecv")) == ["void*", "size_t", "ncclDataType_t", "int", "ncclComm_t", "hipStream_t"]
    assert types(_decl(h, "ncclCommInitAll")) == ["ncclComm_t*", "int", "constint*"]
    assert types(_decl(h, "ncclCommDestroy")) == ["ncclComm_t"]
    assert _decl(h, "ncclGroupStart") in ([""], ["void"]) and _decl(h, "ncclGroupEnd") in ([""], ["void"])
    src = open(GATHER).read()
    # the binding's pointers: same arity and order (datatype and peer are plain ints there; ncclDataType_t / ncclResult_t are enums)
    assert "int (*Send)(const void*, size_t, int, int, rccl_comm_t, hipStream_t)" in src
    assert "int (*Recv)(void*, size_t, int, int, rccl_comm_t, hipStream_t)" in src
    assert "int (*CommInitAll)(rccl_comm_t*, int, const int*)" in src


def test_the_stand_in_exports_what_the_real_library_does():
    fake = open(os.path.join(ROOT, "tests", "hip_emul", "fake_rccl.cpp")).read()
    for s in _bound_symbols():
        assert re.search(r"\b" + s + r"\s*\(", fake), f"tests/hip_emul/fake_rccl.cpp lacks {s}"
