/*
 * TEST INFRASTRUCTURE — CPU oracle, part 7 (fourth mode): what gen.voice's WIND engine adds to the other three.
 *
 * SoundEngine<WIND> (include/soundengine.h:194-230): every ramp is
 *   FreqCtrl_< Mix< LowPassAlgo<PinkNoiseAlgo>, AsymBandPassAlgo<PinkNoiseAlgo>, AsymBandRejectAlgo<PinkNoiseAlgo>, LoudnessVolumeAdjusted<Sine> >,
 *              ShortTermNoiseAdderCtrl< SoundEngineFreqCtrl<SlowIter<AbsIter<PinkNoiseIter>>>, SlowIter<PinkNoiseIter> > >
 * All of it is in the reference and restated here as it is written:
 *   - the pink-noise sound buffer: GaussianPinkNoiseAlgo (include/noise.h:82-165, Voss-McCartney over white_gaussian_noise), soundBuffer's
 *     generate / generate_with_smooth_transition / normalize_audio (source/sound.cpp:84-132,156-168; include/sound.h:96-118), getPinkNoise
 *     and its abs mean (sound.cpp:7-10,22-43), soundBufferWrapperAlgo (audioelement.h:1507-1583), BufferIter (sound.h:180-240);
 *   - the iterators AbsIter, SlowIter, WindFreqIter and Ctrl (audioelement.h:2818-3007);
 *   - SoundEngineFreqCtrl, ShortTermNoiseAdderCtrl (soundengine.h:47-176);
 *   - FilterAlgo, BandPassAlgo_, BandRejectAlgo_, BandAlgo_ (audioelement.h:2004-2324), Mix (:1784-1880).
 * What it takes from the absent cpp.algorithms — PARITY UNPINNED, decided here and in DESIGN.md §5:
 *   - mersenne<SEEDED::No>() behind white_gaussian_noise (noise.h:3-5): draw k of the pink generator = the Box-Muller transform of the
 *     declared counter hash of (key "pinknois", k), as white_noise_sample (oracle_carrier.hpp) does with its own key;
 *   - the same engine behind BufferIter::initializeForRun's start position (sound.h:196-200): a draw is the counter hash of what it is
 *     FOR — (stream key, note id, which spec is played, which of the ramp's six iterators, initializeForRun or set_sample_rate) —
 *     so that a ramp's six start positions do not depend on how many draws anything else made (salt 11).  Only the draws of playNextSpec
 *     survive to a sample: forgetPastSignals redraws every iterator (and takes `prev` there), set_sample_rate redraws the position;
 *   - relevantBits (noise.h:75) = the bit width; count_trailing_zeroes, pow2 = what they say; FLOAT_EPSILON (sound.h:113) = FLT_EPSILON;
 *   - range<float>::getAt(x) = min + x (max - min), in float (audioelement.h:2291);
 *   - NormalizedInterpolation<>::get_unfiltered_value(t, d, from, to) = itp::interpolate as decided for LogRamp, evaluated in double from
 *     the float operands and rounded to the float it returns;
 *   - Filter<double, 1, KIND, VariableOrder> with setOrder(n): n first-order sections in series, each the one-pole of DESIGN.md §5
 *     (y += k (x - y), k = pi w / (1 + pi w); high-pass: x - y), states zero after forgetPastSignals.  That an order-n filter is n such
 *     sections is what BandPassAlgo_::setCompensation states (audioelement.h:2129-2143: the first-order inverse magnitude at the centre,
 *     to the power ORDER);
 *   - VariableOrder = 0 (audioelement.h:1995-1997: "for order >= 2 and 0(adjustable)") and expt<0>(x) = 1: the band-pass' compensation is
 *     the compile-time power ORDER of (1 + 1 / width_factor^2), and SoundEngine instantiates the band algorithms with ORDER = VariableOrder
 *     (soundengine.h:201): as written, the compensation of a run-time-order band-pass is 1 whatever the order;
 *   - the float transcendentals of the controls (std::exp, std::pow(2.f, .), std::log of a float) = the double function rounded to float.
 * What the reference leaves undefined: the Mix's first voice, LowPassAlgo<PinkNoiseAlgo>, is stepped every sample, but nothing ever calls its
 * setFilterAngleIncrements (audioelement.h:2079-2081 is reached only from BandAlgo_::doSetAngleIncrements; Mix::setAngleIncrements reaches
 * FilterAlgo::setAngleIncrements, which is the oscillator's — the noise wrapper's is empty, :1541).  Its output is whatever a Filter whose
 * frequency was never set computes.  Every program of the reference has PINK_NOISE_LP_GAIN = 0 (make_common, gen.voice.h:497): the
 * restatement steps the voice's buffer index and takes its contribution as 0 * (a finite number); a non-zero gain is refused.
 * Reference behaviour kept: the outer control (long-term + short-term noise) drives only the sine voice — FreqCtrl_::setAngleIncrements
 * of the band voices goes to their own SoundEngineFreqCtrl, whose setAngleIncrements is empty (soundengine.h:77-79, audioelement.h:2993);
 * all noise wrappers start at the same index (setStartAngle(0) -> size / 2 - 1) and read the same sample from then on.
 */
#pragma once
#include "oracle_synth.hpp"
#include <cfloat>
#include <mutex>
#include <unordered_map>

namespace ora {

inline double itp_interpolate(int kind, double t, double b, double c, double d); // oracle_voice.hpp

// ---- the pink-noise sound buffer ------------------------------------------------------------------------------------------------------
inline int relevant_bits(unsigned v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }
inline int pink_n_levels(int sample_rate) { // noise.h:59-78: lowest_pink_frequency = 10 Hz, n_changes_min_per_sec = 20
  constexpr float n_changes_min_per_sec = 10.f * 2.f;
  return relevant_bits(static_cast<unsigned int>(sample_rate / n_changes_min_per_sec));
}
inline float pink_white_draw(uint64_t k) { // white_gaussian_noise() of the pink generator: its k-th call (declared policy, see the header)
  uint64_t const h = splitmix64(0x70696e6b6e6f6973ull ^ splitmix64(k * 2 + 9)); // key: "pinknois"; salt 9
  double const u1 = (static_cast<double>(h >> 40) + 1.0) * (1.0 / 16777216.0);
  double const u2 = static_cast<double>((h >> 16) & 0xffffffull) * (1.0 / 16777216.0);
  return static_cast<float>(std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2));
}
struct InterpolatedSignal { // noise.h:11-56
  float value = 0.f, prev_value = 0.f;
  int update_counter = 0;
  void step(int c, float new_value) { update_counter = c; prev_value = value; value = new_value; }
  float get(int c, int level) const {
    int const period = 1 << level; // pow2(level)
    float const interp = (1 + c - update_counter) / static_cast<float>(period);
    return interp * value + (1.f - interp) * prev_value;
  }
};
struct GaussianPinkNoiseAlgo { // noise.h:82-165
  unsigned int counter = 0;
  std::vector<InterpolatedSignal> levels;
  uint64_t draws = 0;
  explicit GaussianPinkNoiseAlgo(int sample_rate) : levels(static_cast<size_t>(pink_n_levels(sample_rate))) {
    for (int i = static_cast<int>(levels.size()) - 1; i >= 1; --i) {
      counter = (1u << (levels.size() - 2)) - (1u << (i - 1));
      do_step();
    }
  }
  void step() { ++counter; do_step(0); do_step(); }
  float get() const {
    float sum = 0.f;
    for (size_t i = 0; i < levels.size(); ++i) sum += levels[i].get(static_cast<int>(counter), static_cast<int>(i));
    return sum / levels.size();
  }
  void do_step() {
    unsigned int index;
    if (counter) index = 1 + static_cast<unsigned>(__builtin_ctz(counter));
    else index = 1 + sizeof(counter) * 8;
    if (index > levels.size() - 1) index = static_cast<unsigned>(levels.size() - 1);
    do_step(static_cast<int>(index));
  }
  void do_step(int index) { levels[static_cast<size_t>(index)].step(static_cast<int>(counter), pink_white_draw(draws++)); }
};
struct PinkBuffer { std::vector<double> values; float abs_mean = 0.f; };
inline int pink_noise_period(int sample_rate) { // soundId{sr, PINK_NOISE, noise_duration = .05f}: freq_to_period_in_samples, sound.functions.h:45-50
  float const freq_hz = .05f, sr = static_cast<float>(sample_rate);
  return static_cast<int>(sr / freq_hz);
}
// soundBuffer<double>::soundBuffer for Sound::PINK_NOISE (sound.cpp:156-168): generate_with_smooth_transition (:102-132) over ONE generator
// (the lambda's static: the transition's samples are drawn first and the buffer's follow on), then normalize_audio (sound.h:96-118).
// One buffer and one generator per sample rate here; the reference's statics make them per process (a second rate would be served the first
// buffer, getPinkNoise :7-10).  The period is >= 20: the centring of sound.cpp:196-219 does not apply.
inline PinkBuffer build_pink_noise(int sample_rate) {
  PinkBuffer pb;
  int const period = std::max(1, pink_noise_period(sample_rate));
  GaussianPinkNoiseAlgo a(sample_rate);
  auto f = [&a] { a.step(); return a.get(); };
  int const transition_length = std::min(10 + period / 10, period);
  std::vector<double> pre(static_cast<size_t>(transition_length));
  for (auto &v : pre) v = f();
  pb.values.resize(static_cast<size_t>(period));
  for (auto &v : pb.values) v = f();
  for (int i = 0; i < transition_length; ++i) {
    float const ratio_pre = (i + 1) / static_cast<float>(transition_length + 1);
    float const angle = ratio_pre * static_cast<float>(M_PI_2);
    float const gain_pre = static_cast<float>(std::sin(static_cast<double>(angle))), gain_v = static_cast<float>(std::cos(static_cast<double>(angle)));
    double &v = pb.values[pb.values.size() - static_cast<size_t>(transition_length) + static_cast<size_t>(i)];
    v = gain_pre * pre[static_cast<size_t>(i)] + gain_v * v;
  }
  { // normalize_audio: the extrema are taken over the values narrowed to float (range<float>::extend)
    float mn = static_cast<float>(pb.values[0]), mx = mn;
    for (double v : pb.values) { float const fv = static_cast<float>(v); mn = std::min(mn, fv); mx = std::max(mx, fv); }
    float M = std::max(-mn, mx);
    float const just_below_one = 1.f - FLT_EPSILON;
    M = just_below_one / M;
    for (double &v : pb.values) v *= M;
  }
  { // getAbsMean, sound.cpp:22-32 (a double sum, returned through the float of getPinkNoiseAbsMean)
    double sum = 0;
    for (double v : pb.values) sum += std::abs(v);
    sum /= static_cast<double>(pb.values.size());
    pb.abs_mean = static_cast<float>(sum);
  }
  return pb;
}
inline PinkBuffer const &getPinkNoise(int sample_rate) {
  static std::mutex mu;
  static std::unordered_map<int, PinkBuffer> bufs;
  std::lock_guard<std::mutex> g(mu);
  auto it = bufs.find(sample_rate);
  if (it == bufs.end()) it = bufs.emplace(sample_rate, build_pink_noise(sample_rate)).first;
  return it->second;
}

// ---- the declared start positions ----------------------------------------------------------------------------------------------------------
struct WindDrawCtx { uint64_t key = 0; int64_t noteid = 0; int32_t play = -1; }; // what the draws of one playNextSpec are for
enum WindIter : int { WI_LONG = 0, WI_SHORT = 1, WI_BP_CENTER = 2, WI_BP_WIDTH = 3, WI_BR_CENTER = 4, WI_BR_WIDTH = 5 };
// static_cast<int>(std::uniform_real_distribution<>{0.f, static_cast<float>(size - 1)}(mersenne<SEEDED::No>())), sound.h:196-200
inline int wind_start_position(WindDrawCtx const &c, int iter, int which, size_t size) {
  uint64_t const id = ((static_cast<uint64_t>(c.noteid) * 64 + static_cast<uint64_t>(c.play)) * 8 + static_cast<uint64_t>(iter)) * 2 + static_cast<uint64_t>(which);
  uint64_t const h = splitmix64(c.key ^ splitmix64(id * 2 + 11));
  double const u = static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0);
  return static_cast<int>(u * static_cast<double>(static_cast<float>(size - 1)));
}

// BufferIter<Sound::PINK_NOISE>, sound.h:180-240
struct PinkIter {
  PinkBuffer const *buf = nullptr;
  WindDrawCtx const *ctx = nullptr;
  int id = 0, pos = 0;
  void set_sample_rate(int s) { buf = &getPinkNoise(s); draw(1); }
  void initializeForRun() { draw(0); }
  void draw(int which) { pos = (ctx && ctx->play >= 0) ? wind_start_position(*ctx, id, which, buf->values.size()) : 0; } // (outside a playNextSpec: a position nobody reads)
  void inc() { ++pos; if (pos == static_cast<int>(buf->values.size())) pos = 0; }
  float get() const { return static_cast<float>(buf->values[static_cast<size_t>(pos)]); }
  float getAbsMean() const { return buf->abs_mean; }
};
// SlowIter<AbsIter<PinkNoiseIter>> (ABS) and SlowIter<PinkNoiseIter>, audioelement.h:2846-2927 over :2818-2841
template <bool ABS> struct SlowPinkIter {
  int32_t n_steps = -1, slow_it = 0;
  int interp = 0; // itp::LINEAR
  float prev = 0.f;
  PinkIter it;
  float cur() const { float const v = it.get(); return ABS ? std::abs(v) : v; }
  void set_sample_rate(int s) { it.set_sample_rate(s); }
  void set_interpolation(int i) { interp = i; }
  void set_n_slow_steps(int32_t n) {
    if (n == n_steps) return;
    if (slow_it) {
      float const ratio = (slow_it + .5f) / static_cast<float>(n_steps);
      slow_it = static_cast<int32_t>(ratio * static_cast<float>(n));
      if (slow_it == n) onMajorStep();
    }
    n_steps = n;
  }
  void initializeForRun() { it.initializeForRun(); onMajorStep(); }
  bool increment() {
    ++slow_it;
    if (slow_it < n_steps) return false;
    onMajorStep();
    return true;
  }
  float value() const { // NormalizedInterpolation<>::get_unfiltered_value(slow_it, n_steps, prev, *it)
    double const from = prev, to = cur();
    return static_cast<float>(itp_interpolate(interp, static_cast<double>(slow_it), from, to - from, static_cast<double>(n_steps)));
  }
  bool isDiminishing() const { return prev > cur(); }
  float getAbsMean() const { return it.getAbsMean(); }
  void onMajorStep() { slow_it = 0; prev = cur(); it.inc(); }
};
// Ctrl<ITER> = WindFreqIter<SlowIter<AbsIter<PinkNoiseIter>>, 3>, audioelement.h:2932-3007: ascending three times faster than descending
struct WindCtrlIter {
  SlowPinkIter<true> it;
  void set_sample_rate(int s) { it.set_sample_rate(s); }
  void forgetPastSignals() { it.initializeForRun(); }
  void step() {
    int const n = it.isDiminishing() ? 1 : 3;
    for (int i = 0; i < n; ++i) if (it.increment()) return;
  }
  float imag() const { return it.value(); }
  float getAbsMean() const { return it.getAbsMean(); }
};
// SoundEngineFreqCtrl<ITER>, soundengine.h:47-106 (T = float)
struct WindFreqCtrl {
  float invApproxRange = 0.f, log_increment_max = 0.f, log_increment_min = 0.f;
  WindCtrlIter ctrl;
  void set_sample_rate(int s) { ctrl.set_sample_rate(s); }
  void setAngleIncrementsRange(float mn, float mx) {
    log_increment_min = static_cast<float>(std::log(static_cast<double>(mn)));
    log_increment_max = static_cast<float>(std::log(static_cast<double>(mx)));
    invApproxRange = 1.f / (2.f * ctrl.getAbsMean());
  }
  void forgetPastSignals() { ctrl.forgetPastSignals(); }
  float step() {
    ctrl.step();
    float const v = ctrl.imag() * invApproxRange;
    float const log_increment = log_increment_min + (log_increment_max - log_increment_min) * v;
    return static_cast<float>(std::exp(static_cast<double>(log_increment)));
  }
};
// ShortTermNoiseAdderCtrl<SoundEngineFreqCtrl<...>, SlowIter<PinkNoiseIter>>, soundengine.h:108-176: a ramp spec's control in WIND
struct WindMainCtrl {
  float noise_amplitude = 0.f, ratio = -1.f;
  WindFreqCtrl ctrl;
  SlowPinkIter<false> noise;
  void set_sample_rate(int s) { ctrl.set_sample_rate(s); noise.set_sample_rate(s); }
  void forgetPastSignals() { ctrl.forgetPastSignals(); noise.initializeForRun(); }
  float step() {
    float const long_term_angle_increment = ctrl.step();
    noise.set_n_slow_steps(static_cast<int32_t>(1 + ratio / long_term_angle_increment));
    noise.increment();
    return long_term_angle_increment * static_cast<float>(std::exp2(static_cast<double>(noise.value() * noise_amplitude)));
  }
  float getAngleIncrementFrom() const { return 0.f; }
  int get_duration_in_samples() const { return std::numeric_limits<int>::max() / 2; }
};

// ---- the voices ------------------------------------------------------------------------------------------------------------------------------
// Filter<double, 1, KIND, VariableOrder> as decided: `order` one-poles in series
struct CascadeFilter {
  bool high_pass = false;
  double k = 0;
  std::vector<double> y;
  double out = 0;
  void setOrder(int n) { y.assign(static_cast<size_t>(n), 0.0); }
  void forgetPastSignals() { std::fill(y.begin(), y.end(), 0.0); out = 0; }
  void initWithAngleIncrement(double w) { double const a = M_PI * w; k = a / (1.0 + a); }
  void feed(double x) {
    for (double &s : y) {
      s = s + k * (x - s);
      x = high_pass ? x - s : s;
    }
    out = x;
  }
  double filtered() const { return out; }
};
// soundBufferWrapperAlgo<Sound::PINK_NOISE>, audioelement.h:1507-1583
struct PinkVoice {
  int index = -1;
  PinkBuffer const *sb = nullptr;
  void set_sample_rate(int s) { sb = &getPinkNoise(s); }
  void setStartAngle(double a) {
    index = static_cast<int>(((a + 1.) * sb->values.size() * 0.5) + 0.5);
    if (index < 0) index = 0;
    --index;
  }
  void step() { ++index; if (index >= static_cast<int>(sb->values.size())) index = 0; }
  double imag() const { return sb->values[static_cast<size_t>(index)]; }
};
// AsymBandPassAlgo / AsymBandRejectAlgo (soundengine.h:180-192) = FreqCtrl_<BandAlgo_<Ctrl<ITER>, BandPassAlgo_ | BandRejectAlgo_>, SoundEngineFreqCtrl<ITER>>
struct WindBand {
  bool reject = false;
  WindFreqCtrl center;       // FreqCtrl_'s control: the band's centre
  WindCtrlIter width;        // BandAlgo_::width
  float width_min = 0.f, width_max = 0.f; // width_range
  double increment = 0, compensation = 1; // expt<VariableOrder>(.) = 1 (see the header)
  // band-pass: cascade = HighPassAlgo<LowPassAlgo<noise>> — one noise wrapper; band-reject: lp and hp each over a wrapper of their own
  PinkVoice noise_a, noise_b;
  CascadeFilter lp, hp;
  WindBand() { hp.high_pass = true; }
  void set_sample_rate(int s) { center.set_sample_rate(s); width.set_sample_rate(s); noise_a.set_sample_rate(s); noise_b.set_sample_rate(s); }
  void setFiltersOrder(int n) { lp.setOrder(n); hp.setOrder(n); }
  void forgetPastSignals() { center.forgetPastSignals(); hp.forgetPastSignals(); lp.forgetPastSignals(); width.forgetPastSignals(); }
  void setStartAngle(double a) { noise_a.setStartAngle(a); noise_b.setStartAngle(a); }
  void step() { // FreqCtrl_::step :3126-3129 -> BandAlgo_::setAngleIncrements / step :2278-2300
    increment = static_cast<double>(center.step());
    width.step();
    float const at = width_min + std::abs(width.imag()) * (width_max - width_min); // range<float>::getAt
    double const width_factor = std::exp2(static_cast<double>(at));
    double const inv_width_factor = 1.0 / width_factor;
    double const low = increment * inv_width_factor, high = increment * width_factor;
    if (!reject) { // BandPassAlgo_: low_index 0, high_index 1 (:2117-2118): HP at `low`, LP at `high`
      hp.initWithAngleIncrement(low);
      lp.initWithAngleIncrement(high);
      noise_a.step();
      lp.feed(noise_a.imag());
      hp.feed(lp.filtered());
    } else { // BandRejectAlgo_: low_index 1, high_index 0 (:2220-2221): HP at `high`, LP at `low`
      hp.initWithAngleIncrement(high);
      lp.initWithAngleIncrement(low);
      noise_a.step();
      lp.feed(noise_a.imag());
      noise_b.step();
      hp.feed(noise_b.imag());
    }
  }
  double imag() const { return reject ? lp.filtered() + hp.filtered() : compensation * hp.filtered(); }
};
// The three noise voices of MixOf<WIND> (the fourth, the sine, is VoiceRamp's own)
struct WindVoices {
  PinkVoice lp_noise; // LowPassAlgo<PinkNoiseAlgo>: stepped; its filter's output is undefined (see the header)
  WindBand bp, br;
  WindVoices() { br.reject = true; }
  void set_sample_rate(int s) { lp_noise.set_sample_rate(s); bp.set_sample_rate(s); br.set_sample_rate(s); }
  void setFiltersOrder(int n) { bp.setFiltersOrder(n); br.setFiltersOrder(n); }
  void forgetPastSignals() { bp.forgetPastSignals(); br.forgetPastSignals(); }
  void setStartAngle(double a) { lp_noise.setStartAngle(a); bp.setStartAngle(a); br.setStartAngle(a); }
  void step() { lp_noise.step(); bp.step(); br.step(); }
};

} // namespace ora
