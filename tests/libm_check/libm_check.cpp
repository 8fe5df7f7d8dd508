// TEST INFRASTRUCTURE: compiles rays_amd/csrc/rays_libm.hpp for the host and counts arguments on which it
// differs from this machine's libm (the libm the reference binary links).  tests/test_cpu_libm.py.
#include <cmath>
#include <cstdint>
#include <cstring>
#define RAYS_LIBM_HOST 1
#include "../../rays_amd/csrc/rays_libm.hpp"

static inline uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline double uniform(uint64_t& s, double lo, double hi) {
  return lo + (hi - lo) * ((splitmix(s) >> 11) * 0x1p-53);
}
static inline bool same(double a, double b) {
  uint64_t x, y;
  std::memcpy(&x, &a, 8);
  std::memcpy(&y, &b, 8);
  return x == y || (a != a && b != b);
}

extern "C" {
double rays_libm_exp(double x) { return rays::libm::exp(x); }
double rays_libm_pow(double x, double y) { return rays::libm::pow(x, y); }

// n uniform arguments in [lo, hi): number of mismatches against ::exp; first offender in *bad
long long check_exp_uniform(long long n, uint64_t seed, double lo, double hi, double* bad) {
  long long miss = 0;
  for (long long i = 0; i < n; i++) {
    const double x = uniform(seed, lo, hi);
    if (!same(rays::libm::exp(x), ::exp(x))) {
      if (!miss) *bad = x;
      miss++;
    }
  }
  return miss;
}
// arguments with random bit patterns (every exponent, both signs, NaNs and infinities included)
long long check_exp_bits(long long n, uint64_t seed, double* bad) {
  long long miss = 0;
  for (long long i = 0; i < n; i++) {
    uint64_t b = splitmix(seed);
    double x;
    std::memcpy(&x, &b, 8);
    if (!same(rays::libm::exp(x), ::exp(x))) {
      if (!miss) *bad = x;
      miss++;
    }
  }
  return miss;
}
long long check_pow_uniform(long long n, uint64_t seed, double xlo, double xhi, double ylo, double yhi, double* bad) {
  long long miss = 0;
  for (long long i = 0; i < n; i++) {
    const double x = uniform(seed, xlo, xhi), y = uniform(seed, ylo, yhi);
    if (!same(rays::libm::pow(x, y), ::pow(x, y))) {
      if (!miss) { bad[0] = x; bad[1] = y; }
      miss++;
    }
  }
  return miss;
}
// x uniform, y from a list of exponents (the profile exponents and the 1/(k+1) of the SG step-size update)
long long check_pow_exponents(long long n, uint64_t seed, double xlo, double xhi, const double* ys, int ny, double* bad) {
  long long miss = 0;
  for (long long i = 0; i < n; i++) {
    const double x = uniform(seed, xlo, xhi), y = ys[i % ny];
    if (!same(rays::libm::pow(x, y), ::pow(x, y))) {
      if (!miss) { bad[0] = x; bad[1] = y; }
      miss++;
    }
  }
  return miss;
}
long long check_pow_bits(long long n, uint64_t seed, double* bad) {
  long long miss = 0;
  for (long long i = 0; i < n; i++) {
    uint64_t a = splitmix(seed), b = splitmix(seed);
    double x, y;
    std::memcpy(&x, &a, 8);
    std::memcpy(&y, &b, 8);
    // keep y in a range where results are not all 0 / inf: random sign, exponent within +-12 of 1
    if (i & 1) {
      b = (b & 0x800fffffffffffffull) | ((uint64_t)(0x3ff - 12 + (splitmix(seed) % 25)) << 52);
      std::memcpy(&y, &b, 8);
    }
    if (!same(rays::libm::pow(x, y), ::pow(x, y))) {
      if (!miss) { bad[0] = x; bad[1] = y; }
      miss++;
    }
  }
  return miss;
}
}
