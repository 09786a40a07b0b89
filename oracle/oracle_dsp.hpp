/*
 * TEST INFRASTRUCTURE — CPU oracle, part 1: the arithmetic the reference outsources to
 * OlivierSohn/cpp.algorithms (absent from /root/reference, unpinned sibling checkout).
 *
 * PARITY UNPINNED for everything in this file: the reference holds no golden vector for
 * any of these functions.  Each function states what is certain from the
 * reference's call site and which published/standard definition is used here
 * (SURVEY.md Appendix B).  They are cross-checked against numpy / closed forms in tests/.
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>

#include "oracle_math.hpp"

namespace ora {

// get_fft_length_for(n, zero_padding_factor) — call sites rt.resynth.lib.periodicfft.cpp:195,218,275.
// Decision: smallest power of two >= n * z.
inline int get_fft_length_for(int n, int zero_padding_factor) {
  long long want = static_cast<long long>(n) * zero_padding_factor;
  int l = 1;
  while (l < want) l <<= 1;
  return l;
}

// ---------------------------------------------------------------------------------------------
// FFT.  fft::Algo_<Fastest,double>::forward / inverse (periodicfft.cpp:234, vocoder.cpp:458-469).
// Certain: forward real DFT, inverse needs a 1/(scale*L) factor.  Here: unnormalised forward
// X_k = sum x_n e^{-2 pi i n k / L}; inverse is unnormalised too and callers scale by 1/L.
// Iterative radix-2 decimation-in-time, twiddles from cos/sin directly.
// ---------------------------------------------------------------------------------------------
// e^{ia} from separate cos and sin calls that a compiler cannot merge into one sincos call (the second argument passes through a
// volatile): glibc's sincos differs from its sin / cos by an ulp at a few arguments per table (k = 1955 of L = 4096, ...), and g++ -O2
// does merge the plain pair — enough to change single spectrum bins between two builds of the same source.  The product's host code
// builds its tables the same way.
inline std::complex<double> twiddle(double a) {
  volatile double a_again = a;
  const double c = std::cos(a);
  const double s = std::sin(a_again);
  return {c, s};
}

struct FftPlan {
  int n = 0;
  std::vector<std::complex<double>> tw; // e^{-2 pi i k / n}, k < n/2
  std::vector<int> rev;
  explicit FftPlan(int n_ = 0) { if (n_) init(n_); }
  void init(int n_) {
    n = n_;
    tw.resize(n / 2 > 0 ? n / 2 : 1);
    for (int k = 0; k < n / 2; ++k) tw[k] = twiddle(-2.0 * M_PI * k / n);
    rev.assign(n, 0);
    int bits = 0;
    while ((1 << bits) < n) ++bits;
    for (int i = 0; i < n; ++i) {
      int r = 0;
      for (int b = 0; b < bits; ++b) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
      rev[i] = r;
    }
  }
  // in place, forward (sign = -1) or inverse (sign = +1, unnormalised)
  void transform(std::complex<double> *a, bool inverse) const {
    for (int i = 0; i < n; ++i) if (i < rev[i]) std::swap(a[i], a[rev[i]]);
    for (int len = 2; len <= n; len <<= 1) {
      int half = len >> 1, step = n / len;
      for (int i = 0; i < n; i += len) {
        for (int j = 0; j < half; ++j) {
          std::complex<double> w = tw[j * step];
          if (inverse) w = std::conj(w);
          std::complex<double> u = a[i + j], v = a[i + j + half] * w;
          a[i + j] = u + v;
          a[i + j + half] = u - v;
        }
      }
    }
  }
};

// Real forward FFT of length L (power of two >= 2) through one complex FFT of length L/2.
// out has L/2+1 bins.
struct RealFft {
  int L = 0;
  FftPlan half;
  std::vector<std::complex<double>> tw; // e^{-2 pi i k / L}, k <= L/2
  std::vector<std::complex<double>> z;
  explicit RealFft(int L_ = 0) { if (L_) init(L_); }
  void init(int L_) {
    L = L_;
    half.init(L / 2);
    tw.resize(L / 2 + 1);
    // e^{-2 pi i k / L} for k <= L/4 from cos / sin; the upper half of the table is its mirror image, W^(L/2 - k) = -conj(W^k), taken as the
    // definition (not as a property a libm may or may not deliver to the last bit): bins k and L/2 - k of the split then share one twiddle
    // product, which the product's kernels use (ra_real_split_pair) and which leaves every bin's operations as written in forward() below
    for (int k = 0; k <= L / 4; ++k) tw[k] = twiddle(-2.0 * M_PI * k / L);
    for (int k = L / 4 + 1; k <= L / 2; ++k) tw[k] = {-tw[L / 2 - k].real(), tw[L / 2 - k].imag()};
    z.resize(L / 2);
  }
  void forward(const double *x, std::complex<double> *X) {
    int h = L / 2;
    if (L == 1) { X[0] = x[0]; return; }
    for (int n = 0; n < h; ++n) z[n] = {x[2 * n], x[2 * n + 1]};
    half.transform(z.data(), false);
    for (int k = 0; k <= h; ++k) {
      std::complex<double> zk = z[k % h];
      std::complex<double> zc = std::conj(z[(h - k) % h]);
      std::complex<double> e = 0.5 * (zk + zc);
      std::complex<double> o = std::complex<double>(0, -0.5) * (zk - zc);
      X[k] = e + tw[k] * o;
    }
  }
  // inverse of forward(): X (L/2+1 bins, Hermitian-implied) -> L real samples, UNNORMALISED
  // (caller multiplies by 1/L, like the reference's 1/(Algo::scale*fft_length), vocoder.cpp:447).
  void inverse(const std::complex<double> *X, double *x) {
    int h = L / 2;
    if (L == 1) { x[0] = X[0].real(); return; }
    for (int k = 0; k < h; ++k) {
      std::complex<double> xk = X[k];
      std::complex<double> xc = std::conj(X[h - k]);
      std::complex<double> e = xk + xc;
      std::complex<double> o = (xk - xc) * std::conj(tw[k]);
      z[k] = e + std::complex<double>(0, 1) * o;
    }
    half.transform(z.data(), true);
    for (int n = 0; n < h; ++n) { x[2 * n] = z[n].real(); x[2 * n + 1] = z[n].imag(); }
  }
};

// ---------------------------------------------------------------------------------------------
// Windows.  half_gaussian_window<T>(4, n, out) + normalize_window(out)
// (periodicfft.cpp:287-293, main.resynth.cpp:26-30).  Certain: n = N/2 points, the full window
// is the half mirrored (lib.cpp:483), truncated at 4 sigma (main.resynth.cpp:16-24).
// Decision: sigma = n/4 samples, w(d) = exp(-d^2 / 2 sigma^2) with d = j + 0.5 the distance to
// the window centre (which lies between samples N/2-1 and N/2); normalised so that the FULL
// window sums to 2, i.e. a bin-centred real sinusoid of amplitude A reads |X| = A.
// ---------------------------------------------------------------------------------------------
enum WindowKind { WINDOW_GAUSSIAN4 = 0, WINDOW_HANN = 1, WINDOW_RECT = 2 };

inline void make_half_window(int kind, int n, std::vector<double> &hw) {
  hw.resize(n);
  double sigma = n / 4.0;
  for (int j = 0; j < n; ++j) {
    double d = j + 0.5;
    switch (kind) {
      case WINDOW_GAUSSIAN4: hw[j] = std::exp(-(d * d) / (2.0 * sigma * sigma)); break;
      case WINDOW_HANN: hw[j] = 0.5 * (1.0 + std::cos(M_PI * d / n)); break;
      default: hw[j] = 1.0; break;
    }
  }
  double s = 0;
  for (double v : hw) s += v;
  for (double &v : hw) v /= s; // sum(half) = 1  => sum(full) = 2
}

// FrequenciesSqMag<double> (periodicfft.cpp:259, vocoder.cpp:128-129).
struct FrequenciesSqMag {
  std::vector<double> frequencies_sqmag; // bins 0 .. L/2-1 (Nyquist bin dropped — decision)
  int fft_length = 0;
  int get_fft_length() const { return fft_length; }
  double bin_index_to_Hz(double sr) const { return sr / fft_length; }
};

// findFrequenciesSqMag<Fastest>(begin, end, stride=1, half_window, zero_pad=1, work, work, out)
// (periodicfft.cpp:304-311): window * x, zero pad to L, forward real FFT, |X|^2.
struct SqMagFft {
  RealFft fft;
  std::vector<double> sig;
  std::vector<std::complex<double>> bins;
  void operator()(const double *x, int n, const std::vector<double> &half_window, FrequenciesSqMag &out) {
    int L = get_fft_length_for(n, 1);
    if (fft.L != L) fft.init(L);
    sig.assign(L, 0.0);
    int h = n / 2;
    for (int j = 0; j < h; ++j) {
      sig[h + j] = x[h + j] * half_window[j];
      sig[h - 1 - j] = x[h - 1 - j] * half_window[j];
    }
    bins.resize(L / 2 + 1);
    fft.forward(sig.data(), bins.data());
    out.fft_length = L;
    out.frequencies_sqmag.resize(L / 2);
    for (int i = 0; i < L / 2; ++i) out.frequencies_sqmag[i] = std::norm(bins[i]);
  }
};

// SqMagToDb / DbToMag (lib.cpp:1594, algo.cpp:25, resynth.hpp:86).
// Decision: 10 log10(x) with x floored at 1e-300 (keeps exact zeros finite); 10^(dB/20).  On the analysis path (peak dB values, peak
// volumes) the logarithm and the power are the explicit sequences of oracle_math.hpp — events depend on their last bits, see there;
// the note driver's volume (resynth.hpp:86, continuous only) keeps the libm power, as the product's host code does.
inline double sqmag_to_db(double x) {
  if (libm_mode()) return 10.0 * std::log10(x > 1e-300 ? x : 1e-300);
  return 10.0 * det_log10(x > 1e-300 ? x : 1e-300);
}
inline double db_to_mag_analysis(double db) { return libm_mode() ? std::pow(10.0, db / 20.0) : det_exp10(db / 20.0); }
inline double db_to_mag(double db) { return std::pow(10.0, db / 20.0); }

struct FreqMag { double freq; double mag_db; };

// extractLocalMaxFreqsMags(sr, sqmag, SqMagToDb, out) (lib.cpp:1592-1595).  Certain: ascending
// freqs (asserted at algo.cpp:15-21), dB magnitudes, quadratic peak interpolation
// (rt.resynth.txt:187-189).  Decision: bin i is a peak iff s[i-1] < s[i] >= s[i+1];
// parabola through the three dB values.
inline void extract_local_max_freqs_mags(double sr, const FrequenciesSqMag &s, std::vector<FreqMag> &out) {
  out.clear();
  int nb = static_cast<int>(s.frequencies_sqmag.size());
  const double *v = s.frequencies_sqmag.data();
  double bin_hz = sr / s.fft_length;
  for (int i = 1; i + 1 < nb; ++i) {
    if (v[i - 1] < v[i] && v[i] >= v[i + 1]) {
      double a = sqmag_to_db(v[i - 1]), b = sqmag_to_db(v[i]), c = sqmag_to_db(v[i + 1]);
      double den = a - 2.0 * b + c;
      double p = den != 0.0 ? 0.5 * (a - c) / den : 0.0;
      out.push_back({(i + p) * bin_hz, b - 0.25 * (a - c) * p});
    }
  }
}

// Filter<T,1,LOW_PASS,1> (audioelement.h:1204-1215,1227): first order low pass driven by an
// "angle increment" w (rad/pi per sample).  Decision: RC one-pole, k = pi w / (1 + pi w),
// y <- y + k (x - y).
struct OnePoleLowPass {
  double k = 1.0, y = 0.0;
  void initWithAngleIncrement(double w) { double a = M_PI * w; k = a / (1.0 + a); }
  void setInitialValue(double v) { y = v; }
  void feed(double x) { y = y + k * (x - y); }
  double filtered() const { return y; }
};

// Limiter<double> (lib.cpp:908, 1253-1258): default-constructed, feedOneFrame(CArray<2, double>) in place, "compressed and limited",
// getTargetCompressionLevel() (:891).  Everything else is cpp.algorithms.  Decision: a stereo-linked feed-forward limiter whose constants
// are in frames — the gain both channels need for a 0.95 ceiling; a target that drops to it at once and recovers by 2e-5 of the
// remaining distance per frame; the applied compression following its target by 1 % per frame; the product clipped to [-1, 1].
struct Limiter {
  double compression = 1.0, target = 1.0;
  void feedOneFrame(double *a) {
    double const peak = std::max(std::abs(a[0]), std::abs(a[1]));
    double needed = 1.0;
    if (peak > 0.95) needed = 0.95 / peak;
    target = needed < target ? needed : target + (1.0 - target) * 2e-5;
    compression = compression + (target - compression) * 0.01;
    for (int j = 0; j < 2; ++j) a[j] = std::min(1.0, std::max(-1.0, a[j] * compression));
  }
  double getTargetCompressionLevel() const { return target; }
};

// itp::interpolate(LINEAR, t, b, c, d) (audioelement.h:733-737, 994-998): Penner form.
inline double itp_linear(double t, double b, double c, double d) { return c * t / d + b; }

// EqualGainXFade<T>::set(len, Sinusoidal) / get(i) (vocoder.cpp:430-431, 515).  Certain: len = S-1
// even, gains sum to one.  Decision: new(i) = 0.5 (1 - cos(pi i / (len + 1))), old = 1 - new.
inline void equal_gain_xfade(int len, int i, double &new_mult, double &old_mult) {
  new_mult = 0.5 * (1.0 - std::cos(M_PI * i / (len + 1.0)));
  old_mult = 1.0 - new_mult;
}

// Declared replacement for mersenne<SEEDED::Yes/No>() draws of
// std::uniform_real_distribution<float>{-1,1} (gen.crtp.h:152, lib.cpp:116): a counter hash of
// (seed, note id, salt) -> 24 bit mantissa -> [-1, 1).
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline float hash_uniform_pm1(uint64_t seed, uint64_t noteid, uint64_t salt) {
  uint64_t h = splitmix64(seed ^ splitmix64(noteid * 2 + salt));
  uint32_t r = static_cast<uint32_t>(h >> 40); // 24 bits
  return -1.0f + 2.0f * (static_cast<float>(r) * (1.0f / 16777216.0f));
}

} // namespace ora
