"""CPU tier: rays_amd/csrc/rays_libm.hpp (the device exp / pow, a restatement of glibc's table-driven
algorithms with glibc's tables and the FMA placement of its x86-64 FMA build) compiled for the host and
compared with THIS machine's libm -- the libm the reference binary links -- bit for bit: uniform arguments
over the kernels' ranges and beyond, random bit patterns (all exponents, signs, NaN, inf), the exponents the
kernels use.  Any mismatch fails."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "libm_check")
LIB = os.path.join(DIR, "liblibm_check.so")


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


pytestmark = pytest.mark.skipif(not _has_fma(), reason="glibc selects its non-FMA exp/pow on this CPU; the device "
                                "functions reproduce the FMA build (every MI355X host)")


@pytest.fixture(scope="module")
def lib():
    srcs = [os.path.join(DIR, "libm_check.cpp"), os.path.join(ROOT, "rays_amd", "csrc", "rays_libm.hpp"),
            os.path.join(ROOT, "rays_amd", "csrc", "rays_libm_tables.inc")]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", srcs[0],
                               "-o", LIB, "-lm"])
    l = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    l.check_exp_uniform.restype = l.check_exp_bits.restype = C.c_longlong
    l.check_pow_uniform.restype = l.check_pow_bits.restype = l.check_pow_exponents.restype = C.c_longlong
    l.check_exp_uniform.argtypes = [C.c_longlong, C.c_uint64, C.c_double, C.c_double, dp]
    l.check_exp_bits.argtypes = [C.c_longlong, C.c_uint64, dp]
    l.check_pow_uniform.argtypes = [C.c_longlong, C.c_uint64] + [C.c_double] * 4 + [dp]
    l.check_pow_exponents.argtypes = [C.c_longlong, C.c_uint64, C.c_double, C.c_double, dp, C.c_int, dp]
    l.check_pow_bits.argtypes = [C.c_longlong, C.c_uint64, dp]
    l.rays_libm_exp.restype = l.rays_libm_pow.restype = C.c_double
    l.rays_libm_exp.argtypes = [C.c_double]
    l.rays_libm_pow.argtypes = [C.c_double, C.c_double]
    return l


N = 4_000_000


@pytest.mark.parametrize("lo,hi", [(-100.0, 0.0), (-30.0, 30.0), (-1e-3, 1e-3), (-745.5, 710.0), (-1100.0, 1100.0)])
def test_exp_equals_libm_uniform(lib, lo, hi):
    # (-100, 0): exp(-xi^2) of the Z function (|xi| <= 10) and the Gaussian density profile
    bad = (C.c_double * 2)()
    assert lib.check_exp_uniform(N, 12345, lo, hi, bad) == 0, f"first mismatch at x = {bad[0]!r}"


def test_exp_equals_libm_any_bit_pattern(lib):
    bad = (C.c_double * 2)()
    assert lib.check_exp_bits(N, 99, bad) == 0, f"first mismatch at x = {bad[0]!r}"
    for x, want in ((0.0, 1.0), (-0.0, 1.0), (float("inf"), float("inf")), (float("-inf"), 0.0), (1000.0, float("inf")),
                    (-1000.0, 0.0), (-745.13, 5e-324)):
        assert lib.rays_libm_exp(x) == want
    assert np.isnan(lib.rays_libm_exp(float("nan")))


@pytest.mark.parametrize("box", [(0.0, 1.0, 0.0, 3.0), (0.0, 10.0, -5.0, 5.0), (1e-300, 1e-290, 0.1, 2.0),
                                 (0.5, 2.0, -1000.0, 1000.0), (-3.0, 3.0, -4.0, 4.0), (0.0, 1e-4, 0.05, 1.0)])
def test_pow_equals_libm_uniform(lib, box):
    bad = (C.c_double * 2)()
    assert lib.check_pow_uniform(N, 777, *box, bad) == 0, f"first mismatch at pow({bad[0]!r}, {bad[1]!r})"


def test_pow_equals_libm_for_the_kernels_exponents(lib):
    """Profile exponents (alpha, alpha - 1 for alpha in 1.5, 2, 2.5, 3, 0.5) on [0, 1) and the 1/(k+1) roots of the
    Shampine-Gordon step-size update (ode_RAYS.f90:1222) on (0, 1]."""
    bad = (C.c_double * 2)()
    ys = np.array([1.5, 0.5, 2.0, 1.0, 2.5, 3.0, -0.5, 0.25] + [1.0 / (k + 1) for k in range(1, 13)])
    assert lib.check_pow_exponents(N, 5, 0.0, 1.0, ys.ctypes.data_as(C.POINTER(C.c_double)), len(ys), bad) == 0, \
        f"first mismatch at pow({bad[0]!r}, {bad[1]!r})"
    assert lib.check_pow_exponents(N, 6, 0.0, 1e-12, ys.ctypes.data_as(C.POINTER(C.c_double)), len(ys), bad) == 0


def test_pow_equals_libm_any_bit_pattern(lib):
    bad = (C.c_double * 2)()
    assert lib.check_pow_bits(N, 31337, bad) == 0, f"first mismatch at pow({bad[0]!r}, {bad[1]!r})"
    inf, nan = float("inf"), float("nan")
    for x, y, want in ((0.0, 1.5, 0.0), (0.0, -1.0, inf), (-0.0, -1.0, -inf), (-8.0, 3.0, -512.0), (-8.0, 2.0, 64.0),
                       (2.0, 0.5, 2.0 ** 0.5), (1.0, nan, 1.0), (nan, 0.0, 1.0), (0.5, inf, 0.0), (2.0, 1e300, inf),
                       (2.0, -1e300, 0.0), (5e-324, 0.5, 2.2227587494850775e-162)):
        assert lib.rays_libm_pow(x, y) == want, (x, y)
    assert np.isnan(lib.rays_libm_pow(-8.0, 0.5)) and np.isnan(lib.rays_libm_pow(nan, 1.0))
