// Lean fp64 log2 / exp10 for the peak refinement of the analysis kernel (5 transcendentals per spectral peak, ~600 peaks per frame:
// 40 % of that kernel's VALU instructions with the general-purpose library versions, which also carry special-case and denormal
// handling that cannot occur here).  Domain: log2 for positive normal x; exp10 for |x| < 300.  Measured on gfx950 against the host's long
// double libm over 1e-300 .. 1e12 and around 1 (scripts/ubench/lean_math_check.hip): log2 <= 2.7 ulp, exp10 <= 1.1 ulp.
// Every step is a correctly rounded IEEE operation (add, multiply, divide, fma, rint, exponent-field arithmetic), so the results are the
// same bits on any conforming machine: the CPU restatement evaluates the same sequences (oracle/oracle_math.hpp; 0 mismatches in 4 M
// arguments, scripts/ubench/det_math_bits.hip), and the pitch / volume lists of the two agree bit for bit (tests/test_gpu_parity.py) —
// which events need: with autotune a pitch one ulp off decides a NoteChange against a NoteOff + NoteOn (see oracle_math.hpp).
#pragma once
#include <hip/hip_runtime.h>

// log2(x), x positive and normal.  x = 2^e m, m in [sqrt(1/2), sqrt(2));  ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716.
__device__ inline double ra_log2(double x) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  int e = (int)((u >> 52) & 0x7ffull) - 1023;
  double m = __longlong_as_double((long long)((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull)); // [1, 2)
  const bool up = m > 1.4142135623730951;
  m = up ? 0.5 * m : m;
  e += up ? 1 : 0;
  const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
  double p = 1.0 / 25.0;
  p = fma(p, s2, 1.0 / 23.0); p = fma(p, s2, 1.0 / 21.0); p = fma(p, s2, 1.0 / 19.0); p = fma(p, s2, 1.0 / 17.0);
  p = fma(p, s2, 1.0 / 15.0); p = fma(p, s2, 1.0 / 13.0); p = fma(p, s2, 1.0 / 11.0); p = fma(p, s2, 1.0 / 9.0);
  p = fma(p, s2, 1.0 / 7.0); p = fma(p, s2, 1.0 / 5.0); p = fma(p, s2, 1.0 / 3.0);
  // ln m = 2 s (1 + s2 p);  log2 m = ln m * log2(e), the constant split in two so that the product keeps its low bits
  const double t = s2 * p;
  const double lnm = fma(2.0 * s, t, 2.0 * s);
  const double hi = 1.4426950408889634, lo = 2.0355273740931033e-17; // log2(e) = hi + lo
  return (double)e + fma(lnm, hi, lnm * lo);
}
__device__ inline double ra_log10(double x) { return ra_log2(x) * 0.30102999566398120; } // log10(2)

// 10^x = 2^(x log2 10):  n = round(x log2 10), f = x log2 10 - n in [-1/2, 1/2] (formed with the constant split in two), 2^f by its series.
__device__ inline double ra_exp10(double x) {
  const double Lh = 3.3219280948873622, Ll = 1.6616175169735920e-16; // log2(10) = Lh + Ll
  const double n = rint(x * Lh);
  const double f = fma(x, Lh, -n) + x * Ll;
  const double g = f * 0.69314718055994531; // f ln 2, |g| <= 0.3466
  double p = 1.0 / 87178291200.0;            // 1 / 14!
  p = fma(p, g, 1.0 / 6227020800.0); p = fma(p, g, 1.0 / 479001600.0); p = fma(p, g, 1.0 / 39916800.0); p = fma(p, g, 1.0 / 3628800.0);
  p = fma(p, g, 1.0 / 362880.0); p = fma(p, g, 1.0 / 40320.0); p = fma(p, g, 1.0 / 5040.0); p = fma(p, g, 1.0 / 720.0);
  p = fma(p, g, 1.0 / 120.0); p = fma(p, g, 1.0 / 24.0); p = fma(p, g, 1.0 / 6.0); p = fma(p, g, 0.5);
  p = fma(p, g, 1.0); p = fma(p, g, 1.0);
  const long long ni = (long long)n;
  return p * __longlong_as_double((ni + 1023) << 52); // 2^n, n well inside the normal range for |x| < 300
}

// 2^t for |t| < 1000: n = round(t), f = t - n (exact), 2^f by the series above.
__device__ inline double ra_exp2(double t) {
  const double n = rint(t);
  const double f = t - n;
  const double g = f * 0.69314718055994531;
  double p = 1.0 / 87178291200.0;
  p = fma(p, g, 1.0 / 6227020800.0); p = fma(p, g, 1.0 / 479001600.0); p = fma(p, g, 1.0 / 39916800.0); p = fma(p, g, 1.0 / 3628800.0);
  p = fma(p, g, 1.0 / 362880.0); p = fma(p, g, 1.0 / 40320.0); p = fma(p, g, 1.0 / 5040.0); p = fma(p, g, 1.0 / 720.0);
  p = fma(p, g, 1.0 / 120.0); p = fma(p, g, 1.0 / 24.0); p = fma(p, g, 1.0 / 6.0); p = fma(p, g, 0.5);
  p = fma(p, g, 1.0); p = fma(p, g, 1.0);
  const long long ni = (long long)n;
  return p * __longlong_as_double((ni + 1023) << 52);
}
// base^y for a positive normal base, as 2^(y log2 base) (the exponent's rounding error grows with |y|: ~6 ulp at |y| = 70)
__device__ inline double ra_pow(double base, double y) { return ra_exp2(y * ra_log2(base)); }
