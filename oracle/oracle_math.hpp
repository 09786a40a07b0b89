/*
 * TEST INFRASTRUCTURE — CPU oracle, part 0: the three transcendental functions of the decision path, written out.
 *
 * The reference's peak refinement calls L0's SqMagToDb / DbToMag (absent, unpinned) and the platform libm's log2
 * (include/midi.h:223-232).  Their last bits are not reproducible — not between two libms, nor between the reference's own
 * builds — and with autotune they decide events: autotuned pitches sit a few ulps off a semitone grid and `track_pitches`
 * (rt.resynth.lib.algo.cpp:256-305) matches a note when the distance is <= max_track_pitches = 1 exactly, so a 1-ulp
 * difference in a pitch turns a NoteChange into NoteOff + NoteOn.  The restatement therefore fixes the three functions as
 * explicit IEEE-754 double operation sequences (add, multiply, divide, fma, rint, exponent-field arithmetic: every step correctly rounded, hence
 * bit-reproducible on any conforming machine); the product evaluates the same sequences (cpp_audio_amd/csrc/ra_math.cuh).
 * Accuracy against the true values: log2 <= 2.7 ulp, exp10 <= 1.1 ulp (tests/test_oracle_kat.py checks them against libm).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace ora {

// Independent-arithmetic mode (tests only): 0 = the explicit sequences below (what the product evaluates: events comparable bit for bit),
// 1 = the platform libm exactly as the reference's text calls it — std::log2 (include/midi.h:223-232), 10 * std::log10 and
// std::pow(10, dB / 20) for SqMagToDb / DbToMag (call sites lib.cpp:1594, algo.cpp:25), std::pow for midi_pitch_to_freq (midi.h:258-264).
// Mode 1 shares no transcendental code with the product: a mistake common to oracle_math.hpp and ra_math.cuh would show up as a pitch
// error beyond a few ulps (tests/test_gpu_parity.py::test_against_the_libm_oracle).  Set before a run, not during one.
inline int &libm_mode() { static int m = 0; return m; }

inline double bits_to_double(uint64_t u) { double d; std::memcpy(&d, &u, sizeof d); return d; }
inline uint64_t double_to_bits(double d) { uint64_t u; std::memcpy(&u, &d, sizeof u); return u; }

// log2(x), x positive and normal.  x = 2^e m, m in [sqrt(1/2), sqrt(2));  ln m = 2 atanh(s), s = (m - 1) / (m + 1).
inline double det_log2(double x) {
  const uint64_t u = double_to_bits(x);
  int e = static_cast<int>((u >> 52) & 0x7ffull) - 1023;
  double m = bits_to_double((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull); // [1, 2)
  const bool up = m > 1.4142135623730951;
  m = up ? 0.5 * m : m;
  e += up ? 1 : 0;
  const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
  double p = 1.0 / 25.0;
  p = std::fma(p, s2, 1.0 / 23.0); p = std::fma(p, s2, 1.0 / 21.0); p = std::fma(p, s2, 1.0 / 19.0); p = std::fma(p, s2, 1.0 / 17.0);
  p = std::fma(p, s2, 1.0 / 15.0); p = std::fma(p, s2, 1.0 / 13.0); p = std::fma(p, s2, 1.0 / 11.0); p = std::fma(p, s2, 1.0 / 9.0);
  p = std::fma(p, s2, 1.0 / 7.0); p = std::fma(p, s2, 1.0 / 5.0); p = std::fma(p, s2, 1.0 / 3.0);
  const double t = s2 * p;
  const double lnm = std::fma(2.0 * s, t, 2.0 * s);
  const double hi = 1.4426950408889634, lo = 2.0355273740931033e-17; // log2(e) = hi + lo
  return static_cast<double>(e) + std::fma(lnm, hi, lnm * lo);
}
inline double det_log10(double x) { return det_log2(x) * 0.30102999566398120; } // log10(2)

// 10^x for |x| < 300: n = round(x log2 10), f = x log2 10 - n (constant split in two), 2^f by its series.
inline double det_exp10(double x) {
  const double Lh = 3.3219280948873622, Ll = 1.6616175169735920e-16; // log2(10) = Lh + Ll
  const double n = std::rint(x * Lh);
  const double f = std::fma(x, Lh, -n) + x * Ll;
  const double g = f * 0.69314718055994531;
  double p = 1.0 / 87178291200.0; // 1 / 14!
  p = std::fma(p, g, 1.0 / 6227020800.0); p = std::fma(p, g, 1.0 / 479001600.0); p = std::fma(p, g, 1.0 / 39916800.0); p = std::fma(p, g, 1.0 / 3628800.0);
  p = std::fma(p, g, 1.0 / 362880.0); p = std::fma(p, g, 1.0 / 40320.0); p = std::fma(p, g, 1.0 / 5040.0); p = std::fma(p, g, 1.0 / 720.0);
  p = std::fma(p, g, 1.0 / 120.0); p = std::fma(p, g, 1.0 / 24.0); p = std::fma(p, g, 1.0 / 6.0); p = std::fma(p, g, 0.5);
  p = std::fma(p, g, 1.0); p = std::fma(p, g, 1.0);
  const int64_t ni = static_cast<int64_t>(n);
  return p * bits_to_double(static_cast<uint64_t>(ni + 1023) << 52);
}

// 2^t for |t| < 1000: n = round(t), f = t - n (exact), 2^f by the same series.
inline double det_exp2(double t) {
  const double n = std::rint(t);
  const double f = t - n;
  const double g = f * 0.69314718055994531;
  double p = 1.0 / 87178291200.0;
  p = std::fma(p, g, 1.0 / 6227020800.0); p = std::fma(p, g, 1.0 / 479001600.0); p = std::fma(p, g, 1.0 / 39916800.0); p = std::fma(p, g, 1.0 / 3628800.0);
  p = std::fma(p, g, 1.0 / 362880.0); p = std::fma(p, g, 1.0 / 40320.0); p = std::fma(p, g, 1.0 / 5040.0); p = std::fma(p, g, 1.0 / 720.0);
  p = std::fma(p, g, 1.0 / 120.0); p = std::fma(p, g, 1.0 / 24.0); p = std::fma(p, g, 1.0 / 6.0); p = std::fma(p, g, 0.5);
  p = std::fma(p, g, 1.0); p = std::fma(p, g, 1.0);
  const int64_t ni = static_cast<int64_t>(n);
  return p * bits_to_double(static_cast<uint64_t>(ni + 1023) << 52);
}
// base^y for a positive normal base, as 2^(y log2 base)
inline double det_pow(double base, double y) { return det_exp2(y * det_log2(base)); }

} // namespace ora
