// rays_libm.hpp -- exp() and pow() that return what the reference's libm returns, bit for bit.
//
// The reference calls libm in four places on the hot path: exp in the Gaussian slab density
// (slab_eq_m.f90:262) and in Im Z(xi) of the damping (zfunctions_m.f90:403), pow in the profile
// exponents (slab_eq_m.f90:369-379, solovev_eq_m.f90:219-262, axisym_toroid_eq_m.f90) and in the
// Shampine-Gordon step-size update (ode_RAYS.f90:1222).  The reference's finite-difference dD
// (deriv_num.f90, differences over 1e-6) amplifies one ulp of such a value by ~1e8, and the adaptive
// integrator turns it into a different step sequence; with the device library's own exp/pow (ocml,
// <= 1-2 ulp, rounding differently from glibc) two fixtures missed the "1e-10 relative per step" bar by
// 2.7e-8 and 2.7e-10 (tests/test_gpu_baseline_kernels.py::test_per_step_parity_from_reference_points).
//
// The reference binary is linked against glibc (2.35 in this image; the algorithms date from 2.28):
// Szabolcs Nagy's table-driven exp and pow (sysdeps/ieee754/dbl-64/e_exp.c, e_pow.c; published under MIT
// as ARM optimized-routines math/exp.c, math/pow.c).  This file restates those two algorithms for the
// device: same tables (rays_libm_tables.inc, read from the image's libm by tools/gen_libm_tables.py), same
// operations in the same order, and a fused multiply-add exactly where the x86-64 FMA build of glibc
// (__ieee754_exp_fma / __ieee754_pow_fma, the variant glibc's ifunc selects on every AVX2+FMA CPU,
// which includes the EPYC hosts of MI355X nodes) has one -- read off its disassembly.  errno / fenv side
// effects are dropped; results, including the over/underflow and special-operand paths, are kept.
// tests/test_cpu_libm.py compiles this header for the host and compares it with libm on tens of
// millions of arguments (identical), and with the kernels' argument ranges exhaustively sampled.
#pragma once

#ifdef RAYS_LIBM_HOST  // stand-alone host build (tests): no HIP
#define RAYS_LIBM_FN static inline
#define RAYS_LIBM_CONST static constexpr
#define RAYS_LIBM_TABLE static const
#elif defined(RAYS_HOST_EMUL)
#define RAYS_LIBM_FN static inline
#define RAYS_LIBM_CONST static constexpr
#define RAYS_LIBM_TABLE static const
#else
#define RAYS_LIBM_FN __device__ inline
#define RAYS_LIBM_CONST static constexpr
#define RAYS_LIBM_TABLE static __device__ const
#endif

namespace rays {
namespace libm {

// The tolerance-flavour translation units are compiled with -fassociative-math and FMA contraction (Makefile:
// TOLFLAGS).  Neither may touch this file: the compensated terms below (lo2 = t1 - t2 + r, lo4 = t2 - hi0 + ar2,
// loglo = hi0 - loghi + lo, ...) are exact-arithmetic identities that re-association folds to 0, and every FMA of
// glibc's build is written out.  The pragmas need -ffp-contract=fast-honor-pragmas (plain `fast` fuses in the backend
// whatever the source says); tests/test_cpu_libm.py builds this header with TOLFLAGS and compares it with libm.
#if defined(__clang__) && defined(RAYS_TOL_FLAVOUR)
#pragma clang fp reassociate(off) contract(off)
#endif

#include "rays_libm_tables.inc"

typedef unsigned long long u64;
RAYS_LIBM_FN u64 asu64(double x) { return __builtin_bit_cast(u64, x); }
RAYS_LIBM_FN double asf64(u64 x) { return __builtin_bit_cast(double, x); }
RAYS_LIBM_FN unsigned top12(double x) { return (unsigned)(asu64(x) >> 52); }
RAYS_LIBM_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

constexpr double kInf = __builtin_inf();

// exp.c: specialcase() -- the scale factor 2^(k/N) alone would over/underflow
RAYS_LIBM_FN double exp_specialcase(double tmp, u64 sbits, u64 ki, bool fused_up) {
  if ((ki & 0x80000000ull) == 0) {
    // k > 0: the exponent of scale might have overflowed by <= 460
    sbits -= 1009ull << 52;
    const double scale = asf64(sbits);
    return 0x1p1009 * fma_(scale, tmp, scale);
  }
  // k < 0: take care in the subnormal range
  sbits += 1022ull << 52;
  const double scale = asf64(sbits);
  const double st = scale * tmp;  // (a plain product in the FMA build too)
  double y = scale + st;
  if (__builtin_fabs(y) < 1.0) {
    // round y to the right precision before scaling it into the subnormal range
    const double one = y < 0.0 ? -1.0 : 1.0;
    double lo = scale - y + st;
    const double hi = one + y;
    lo = one - hi + y + lo;
    y = (hi + lo) - one;
    if (y == 0.0) y = asf64(sbits & 0x8000000000000000ull);  // keep the sign of zero
  }
  (void)fused_up;
  return 0x1p-1022 * y;
}

// exp(x)                                   e_exp.c: __exp  (x86-64 FMA build)
RAYS_LIBM_FN double exp(double x) {
  unsigned abstop = top12(x) & 0x7ff;
  if (abstop - 0x3c9u >= 0x3fu) {  // |x| < 2^-54 or |x| >= 512 or NaN
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;  // tiny
    if (abstop >= 0x409u) {                              // |x| >= 1024, inf, NaN
      if (asu64(x) == asu64(-kInf)) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;
      return (asu64(x) >> 63) ? 0.0 : kInf;              // __math_uflow / __math_oflow
    }
    abstop = 0;  // 512 <= |x| < 1024: the result may over/underflow, handled below
  }
  // x = ln2/N k + r,  k integer, r in [-ln2/2N, ln2/2N]
  const double z = fma_(x, kExpInvLn2N, kExpShift);
  const u64 ki = asu64(z);
  const double kd = z - kExpShift;
  double r = fma_(kd, kExpNegLn2HiN, x);
  r = fma_(kd, kExpNegLn2LoN, r);
  // 2^(k/N) ~= scale (1 + tail)
  const u64 idx = 2 * (ki & 127);
  const u64 top = ki << 45;
  const double tail = asf64(kExpTab[idx]);
  const u64 sbits = kExpTab[idx + 1] + top;
  const double r2 = r * r;
  // tmp = tail + r + r2 (C2 + r C3) + r2 r2 (C4 + r C5)
  const double p23 = fma_(r, kExpPoly[1], kExpPoly[0]);
  const double p45 = fma_(r, kExpPoly[3], kExpPoly[2]);
  const double lo = fma_(p23, r2, r + tail);
  const double tmp = fma_(r2 * r2, p45, lo);
  if (abstop == 0) return exp_specialcase(tmp, sbits, ki, true);
  const double scale = asf64(sbits);
  return fma_(scale, tmp, scale);
}

// pow.c: checkint() -- 0: not an integer, 1: odd, 2: even
RAYS_LIBM_FN int checkint(u64 iy) {
  const int e = (int)((iy >> 52) & 0x7ff);
  if (e < 0x3ff) return 0;
  if (e > 0x3ff + 52) return 2;
  if (iy & ((1ull << (0x3ff + 52 - e)) - 1)) return 0;
  if (iy & (1ull << (0x3ff + 52 - e))) return 1;
  return 2;
}
RAYS_LIBM_FN bool zeroinfnan(u64 i) { return 2 * i - 1 >= 2 * asu64(kInf) - 1; }

// pow(x, y)                                e_pow.c: __pow  (x86-64 FMA build)
RAYS_LIBM_FN double pow(double x, double y) {
  u64 sign_bias = 0;
  u64 ix = asu64(x);
  const u64 iy = asu64(y);
  unsigned topx = top12(x);
  const unsigned topy = top12(y);
  if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
    // x is subnormal, zero, inf, NaN or negative; or |y| < 2^-65, >= 2^63, inf or NaN
    if (zeroinfnan(iy)) {
      if (2 * iy == 0) return 1.0;
      if (ix == asu64(1.0)) return 1.0;
      if (2 * ix > 2 * asu64(kInf) || 2 * iy > 2 * asu64(kInf)) return x + y;
      if (2 * ix == 2 * asu64(1.0)) return 1.0;
      if ((2 * ix < 2 * asu64(1.0)) == !(iy >> 63)) return 0.0;  // |x| < 1 && y = inf, |x| > 1 && y = -inf
      return y * y;
    }
    if (zeroinfnan(ix)) {
      double x2 = x * x;
      if ((ix >> 63) && checkint(iy) == 1) x2 = -x2;
      return (iy >> 63) ? 1.0 / x2 : x2;  // (1/+-0 = +-inf: __math_divzero's value)
    }
    // here x and y are non-zero finite
    if (ix >> 63) {  // finite x < 0
      const int yint = checkint(iy);
      if (yint == 0) return (x - x) / (x - x);  // __math_invalid: NaN
      if (yint == 1) sign_bias = 0x800ull << 7;
      ix &= 0x7fffffffffffffffull;
      topx &= 0x7ff;
    }
    if ((topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
      // (sign_bias == 0 here: y is not an odd integer)
      if (ix == asu64(1.0)) return 1.0;
      if ((topy & 0x7ff) < 0x3beu) return ix > asu64(1.0) ? 1.0 + y : 1.0 - y;  // |y| < 2^-65: x^y ~ 1 + y log x
      return (ix > asu64(1.0)) == (topy < 0x800u) ? kInf : 0.0;                // __math_oflow / __math_uflow
    }
    if (topx == 0) {  // normalise a subnormal x
      ix = asu64(x * 0x1p52);
      ix &= 0x7fffffffffffffffull;
      ix -= 52ull << 52;
    }
  }
  // ---- log_inline: hi + lo = log(x), lo ~ 2^-68 relative ----
  // x = 2^k z, z in [0x1.69555p-1, 0x1.69555p0); i = index of the subinterval of z
  const u64 tmpi = ix - 0x3fe6955500000000ull;
  const int i = (int)((tmpi >> 45) & 127);
  const int k = (int)((long long)tmpi >> 52);
  const u64 iz = ix - (tmpi & (0xfffull << 52));
  const double z = asf64(iz);
  const double kd = (double)k;
  const double invc = kPowLogTab[i][0], logc = kPowLogTab[i][1], logctail = kPowLogTab[i][2];
  // log(x) = k ln2 + log(c) + log1p(z/c - 1);  r = z/c - 1 exactly (one FMA)
  const double r = fma_(z, invc, -1.0);
  // k ln2 + log(c) + r
  const double t1 = fma_(kd, kPowLn2Hi, logc);
  const double t2 = t1 + r;
  const double lo1 = fma_(kd, kPowLn2Lo, logctail);
  const double lo2 = t1 - t2 + r;
  // the quadratic term A[0] r^2 = -r^2/2 in extra precision
  const double ar = kPowLogPoly[0] * r;
  const double ar2 = r * ar;
  const double ar3 = r * ar2;
  const double hi0 = t2 + ar2;
  const double lo3 = fma_(ar, r, -ar2);
  const double lo4 = t2 - hi0 + ar2;
  // p = log1p(r) - r - A[0] r^2
  const double q12 = fma_(r, kPowLogPoly[2], kPowLogPoly[1]);
  const double q34 = fma_(r, kPowLogPoly[4], kPowLogPoly[3]);
  const double q56 = fma_(r, kPowLogPoly[6], kPowLogPoly[5]);
  const double pp = fma_(ar2, fma_(q56, ar2, q34), q12);
  const double lo = fma_(ar3, pp, lo1 + lo2 + lo3 + lo4);
  const double loghi = hi0 + lo;
  const double loglo = hi0 - loghi + lo;
  // ---- y log(x) in two pieces ----
  const double ehi = y * loghi;
  const double elo = fma_(y, loglo, fma_(loghi, y, -ehi));
  // ---- exp_inline(ehi, elo, sign_bias) ----
  unsigned abstop = top12(ehi) & 0x7ff;
  if (abstop - 0x3c9u >= 0x3fu) {
    if (abstop - 0x3c9u >= 0x80000000u) {  // |ehi| tiny: the result is 1 (+ a rounding nudge)
      const double one = 1.0 + ehi;
      return sign_bias ? -one : one;
    }
    if (abstop >= 0x409u) {  // |ehi| >= 1024
      const double v = (asu64(ehi) >> 63) ? 0.0 : kInf;
      return sign_bias ? -v : v;
    }
    abstop = 0;
  }
  const double ez = fma_(ehi, kExpInvLn2N, kExpShift);
  const u64 ki = asu64(ez);
  const double ekd = ez - kExpShift;
  double er = fma_(ekd, kExpNegLn2HiN, ehi);
  er = fma_(ekd, kExpNegLn2LoN, er);
  er = elo + er;  // the tail of y log(x)
  const u64 idx = 2 * (ki & 127);
  const u64 top = (ki + sign_bias) << 45;
  const double tail = asf64(kExpTab[idx]);
  const u64 sbits = kExpTab[idx + 1] + top;
  const double er2 = er * er;
  const double p23 = fma_(er, kExpPoly[1], kExpPoly[0]);
  const double p45 = fma_(er, kExpPoly[3], kExpPoly[2]);
  const double elo2 = fma_(p23, er2, er + tail);
  const double etmp = fma_(er2 * er2, p45, elo2);
  if (abstop == 0) return exp_specialcase(etmp, sbits, ki, true);
  const double scale = asf64(sbits);
  return fma_(etmp, scale, scale);
}

#if defined(__clang__) && defined(RAYS_TOL_FLAVOUR)
#pragma clang fp reassociate(on) contract(fast)  // back to the translation unit's TOLFLAGS
#endif

}  // namespace libm
}  // namespace rays
