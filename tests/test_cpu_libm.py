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


# ---- the same header inside a TOLERANCE-flavour translation unit (rays_amd/csrc/Makefile: TOLFLAGS) -----------------
# -fassociative-math would fold the compensated sums of pow (lo2 = t1 - t2 + r, ...) to zero and -ffp-contract=fast would
# fuse where glibc's build does not; rays_libm.hpp switches both off for itself with `#pragma clang fp`, which needs
# -ffp-contract=fast-honor-pragmas.  Built with the ROCm clang and the Makefile's own flags, compared with libm.
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
TOL_LIB = os.path.join(DIR, "liblibm_check_tolflags.so")


def _makefile_tolflags():
    mk = open(os.path.join(ROOT, "rays_amd", "csrc", "Makefile")).read()
    line = next(l for l in mk.splitlines() if l.startswith("TOLFLAGS"))
    flags = [f for f in line.split(":=", 1)[1].split() if f.startswith(("-ffp-contract=fast", "-fassociative", "-fno-signed", "-fno-trapping",
                                                                        "-DRAYS_TOL_FLAVOUR"))]
    assert "-ffp-contract=fast-honor-pragmas" in flags and "-fassociative-math" in flags and "-DRAYS_TOL_FLAVOUR" in flags, flags
    return flags


@pytest.fixture(scope="module")
def tol_lib():
    if not os.path.exists(CLANG):
        pytest.skip("no ROCm clang here")
    srcs = [os.path.join(DIR, "libm_check.cpp"), os.path.join(ROOT, "rays_amd", "csrc", "rays_libm.hpp"),
            os.path.join(ROOT, "rays_amd", "csrc", "rays_libm_tables.inc"), os.path.join(ROOT, "rays_amd", "csrc", "Makefile")]
    if not os.path.exists(TOL_LIB) or any(os.path.getmtime(s) > os.path.getmtime(TOL_LIB) for s in srcs):
        subprocess.check_call([CLANG, "-O3", "-std=c++17", "-mfma", *_makefile_tolflags(), "-fPIC", "-shared", srcs[0],
                               "-o", TOL_LIB, "-lm"])
    l = C.CDLL(TOL_LIB)
    dp = C.POINTER(C.c_double)
    l.check_exp_uniform.restype = l.check_pow_uniform.restype = l.check_pow_bits.restype = l.check_pow_exponents.restype = C.c_longlong
    l.check_exp_uniform.argtypes = [C.c_longlong, C.c_uint64, C.c_double, C.c_double, dp]
    l.check_pow_uniform.argtypes = [C.c_longlong, C.c_uint64] + [C.c_double] * 4 + [dp]
    l.check_pow_exponents.argtypes = [C.c_longlong, C.c_uint64, C.c_double, C.c_double, dp, C.c_int, dp]
    l.check_pow_bits.argtypes = [C.c_longlong, C.c_uint64, dp]
    return l


def test_libm_header_under_the_tolerance_flags_is_still_libm(tol_lib):
    bad = (C.c_double * 2)()
    for lo, hi in ((-100.0, 0.0), (-30.0, 30.0), (-745.5, 710.0)):
        assert tol_lib.check_exp_uniform(N // 4, 12345, lo, hi, bad) == 0, f"exp({bad[0]!r})"
    for box in ((0.0, 1.0, 0.0, 3.0), (0.0, 10.0, -5.0, 5.0), (0.5, 2.0, -1000.0, 1000.0)):
        assert tol_lib.check_pow_uniform(N // 4, 777, *box, bad) == 0, f"pow({bad[0]!r}, {bad[1]!r})"
    assert tol_lib.check_pow_bits(N // 4, 31337, bad) == 0, f"pow({bad[0]!r}, {bad[1]!r})"
    ys = np.array([1.5, 0.5, 2.0, 2.5, 3.0, -0.5, 0.25] + [1.0 / (k + 1) for k in range(1, 13)])
    assert tol_lib.check_pow_exponents(N // 4, 5, 0.0, 1.0, ys.ctypes.data_as(C.POINTER(C.c_double)), len(ys), bad) == 0


def test_without_the_pragmas_the_tolerance_flags_do_break_it(tmp_path):
    """The control: the same build with the header's pragmas disabled differs from libm (so the test above tests something)."""
    if not os.path.exists(CLANG):
        pytest.skip("no ROCm clang here")
    src = open(os.path.join(ROOT, "rays_amd", "csrc", "rays_libm.hpp")).read().replace("#pragma clang fp", "// #pragma clang fp")
    (tmp_path / "rays_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "tests" / "libm_check").mkdir(parents=True)
    (tmp_path / "rays_amd" / "csrc" / "rays_libm.hpp").write_text(src)
    (tmp_path / "rays_amd" / "csrc" / "rays_libm_tables.inc").write_text(open(os.path.join(ROOT, "rays_amd", "csrc", "rays_libm_tables.inc")).read())
    (tmp_path / "tests" / "libm_check" / "libm_check.cpp").write_text(open(os.path.join(DIR, "libm_check.cpp")).read())
    out = str(tmp_path / "lib.so")
    subprocess.check_call([CLANG, "-O3", "-std=c++17", "-mfma", *_makefile_tolflags(), "-fPIC", "-shared",
                           str(tmp_path / "tests" / "libm_check" / "libm_check.cpp"), "-o", out, "-lm"])
    l = C.CDLL(out)
    l.check_pow_uniform.restype = C.c_longlong
    l.check_pow_uniform.argtypes = [C.c_longlong, C.c_uint64] + [C.c_double] * 4 + [C.POINTER(C.c_double)]
    bad = (C.c_double * 2)()
    assert l.check_pow_uniform(200000, 777, 0.0, 1.0, 0.0, 3.0, bad) > 0
