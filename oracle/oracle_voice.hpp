/*
 * TEST INFRASTRUCTURE — CPU oracle, part 6: gen.voice — the SWEEP, BIRDS, ROBOTS and WIND sound engines (WIND's own parts: oracle_wind.hpp).
 *
 * voice::Impl_<..., MODE, ...> (include/gen.voice.h:1087-1190) = ImplCRTP (include/gen.crtp.h, SynchronizePhase::No,
 * DefaultStartPhase::Zero) over VolumeAdjusted<SoundEngine<MODE, 2, ...>> (include/soundengine.h:240-1127), whose three "ramps" are
 *   StereoPanned<VolumeAdjusted<Enveloped<FreqCtrl_<Mix<LoudnessVolumeAdjusted<SineOscillatorAlgo<double>>>, LogRamp<double>>, AHDSREnvelope>>>
 * (soundengine.h:175-180,205-211,246-258; audioelement.h:1784-1880 Mix, :1275-1317 LoudnessVolumeAdjusted, :2464-2572 LogRamp).
 * The engine is restated as it is written: per sample, three ramp objects, a list of up to 30 ramp specs filled by a Markov chain at
 * note-on (create_birds :565-595, create_robot :597-695, create_sweep :697-719, do_initialize :963-1003) and played one after the other
 * by orchestrate_algos / playNextSpec (:744-803).  WIND (:721-735): the same engine over another Mix and another control, oracle_wind.hpp.
 *
 * What this path takes from the absent cpp.algorithms — PARITY UNPINNED, decided here and in DESIGN.md §5:
 *   - MarkovChain: its contract is visible at the call sites — `emplace(lambda)` makes a node, the lambda is called with
 *     (MarkovMove::ENTER_NODE | LEAVE_NODE, me, from_to); `def_markov_transition(a, b, p)`; `initialize(start)`; `step<Exec>(r)` moves
 *     along the first transition whose cumulated probability exceeds r in [0, 1) and stays when none does; `step_normalized<Exec>(r)` does
 *     the same with the probabilities rescaled to sum 1 (r * sum against the raw sums).  Decision: transitions in definition order; on a
 *     move the LEAVE lambda of the node left runs before the ENTER lambda of the node entered.
 *   - mersenne<SEEDED::Yes>(): a global engine that setupAudioElement reseeds per note with SEED (1 .. 257; 0 = a counter that advances
 *     per note, gen.voice.h:910-925).  Decision (the declared counter hash of this build, salt 5): the k-th draw after a reseed with value
 *     s is u_k = top 24 bits of splitmix64(seed ^ splitmix64(((s << 32) | k) * 2 + 5)) / 2^24; uniform_real{a, b} = (b - a) u + a in
 *     float, uniform_int{0, 1} = u < 0.5 ? 0 : 1.  In SWEEP mode the engine is not reseeded and its draws are inert (below).
 *   - itp::interpolate / NormalizedInterpolation::get_unfiltered_value(t, d, from, to): Penner's easing equations in their published
 *     (t, b, c, d) form with b = from, c = to - from (audioelement.h:733-737 shows the argument order); PROPORTIONAL_VALUE_DERIVATIVE =
 *     from * (to / from)^(t / d).  The enum's numbering is cpp.algorithms'; this build's own is `VoiceItp`.
 *   - random pan (gen.voice.h:980-995): the first draw after the reseed (BIRDS / ROBOTS); in SWEEP the declared hash with salt 3.
 * SWEEP: create_sweep is two nodes joined by a certain transition and uniform{min_exp, max_exp} has min_exp == max_exp
 * (set_length_exp(ex, ex), gen.voice.h:927-929): the voice is deterministic whatever is drawn.
 * Reference quirks restated as they are: Mix::angleIncrements() starts its minimum at 0 (audioelement.h:1856-1865), so for positive
 * increments FreqCtrl_::angleIncrements() falls back to the control's `from`; a note-off makes the element inactive at once
 * (SoundEngine::onKeyReleased clears rt_active, soundengine.h:345-353: "TODO we should delay this"); LogRamp's proportionality constant is
 * 1.f for a constant-frequency ramp (audioelement.h:2570), so such a ramp's time runs at `from` per sample; make_bird stores `i`, not
 * `freq_i`, as FREQ_TRANSITION_INTERPOLATION (gen.voice.h:607-612); the ramps' envelopes never learn a minimum change duration
 * (nobody calls Enveloped::setAngleIncrements on them).
 */
#pragma once
#include "oracle_synth.hpp"
#include "oracle_wind.hpp"

namespace ora {

enum VoiceMode : int32_t { VOICE_BIRDS = 0, VOICE_ROBOTS = 1, VOICE_SWEEP = 2, VOICE_WIND = 3 }; // SoundEngineMode, soundengine.h:31-40
enum VoiceXfade : int32_t { XFADE_NO = 0, XFADE_NONTRIVIAL = 1, XFADE_ALL = 2 }; // FreqXfade, soundengine.h:3-11
enum VoiceItp : int32_t {
  ITP_LINEAR = 0, ITP_PROPORTIONAL_VALUE_DERIVATIVE = 1,
  ITP_EASE_IN_QUAD, ITP_EASE_OUT_QUAD, ITP_EASE_INOUT_QUAD, ITP_EASE_IN_CUBIC, ITP_EASE_OUT_CUBIC, ITP_EASE_INOUT_CUBIC,
  ITP_EASE_IN_QUART, ITP_EASE_OUT_QUART, ITP_EASE_INOUT_QUART, ITP_EASE_IN_QUINT, ITP_EASE_OUT_QUINT, ITP_EASE_INOUT_QUINT,
  ITP_EASE_IN_SINE, ITP_EASE_OUT_SINE, ITP_EASE_INOUT_SINE, ITP_EASE_IN_EXPO, ITP_EASE_OUT_EXPO, ITP_EASE_INOUT_EXPO,
  ITP_EASE_IN_CIRC, ITP_EASE_OUT_CIRC, ITP_EASE_INOUT_CIRC, ITP_END
};

struct ora_voice_params { // a program's parameters, de-normalised (gen.voice.h:55-208, 474-529)
  int32_t sample_rate;
  int32_t n_channels;           // 2 * nVoices
  float length_ms;              // LENGTH
  float length_exponent;        // LENGTH_EXPONENT
  float xfade_length_seconds;   // XFADE_LENGTH
  float low_freq, high_freq;    // LOW_FREQ, HIGH_FREQ (SWEEP)
  float sine_gain;              // SINE_GAIN
  float loudness_level;         // LOUDNESS_LEVEL
  float loudness_compensation;  // LOUDNESS_COMPENSATION_AMOUNT (log_ratio)
  int32_t loudness_ref_freq_index;
  float pan;                    // PAN
  int32_t random_pan;           // RANDOM_PAN (as a boolean: 1 = draw the pan per note)
  uint64_t seed;                // the declared hash's key (per stream)
  // ---- BIRDS / ROBOTS ----
  int32_t mode;                 // VoiceMode
  int32_t interpolation;        // INTERPOLATION (VoiceItp)
  int32_t markov_start_node, markov_pre_tries, markov_min_path_length, markov_additional_tries;
  float articulative_pause_seconds; // MARKOV_ARTICULATIVE_PAUSE_LENGTH
  float freq_scatter;           // FREQ_SCATTER
  float length_exponent_scatter;// LENGTH_EXPONENT_SCATTER
  float phase_ratio1, phase_ratio2;
  float d1, d2;                 // D1, D2 (ROBOTS; halftones)
  float harmonic_attenuation;   // HARMONIC_ATTENUATION (ROBOTS)
  int32_t xfade_freq;           // MARKOV_XFADE_FREQ (BIRDS; VoiceXfade)
  float freq_transition_seconds;// FREQ_TRANSITION_LENGTH (BIRDS)
  int32_t freq_interpolation;   // FREQ_TRANSITION_INTERPOLATION (BIRDS)
  int32_t seed_param;           // SEED: 0 = another seed per note (1 .. 257, cycling), else fixed
  // ---- WIND (gen.voice.h:142-175) ----
  float pink_noise_lp_gain, pink_noise_bp_gain, pink_noise_br_gain; // PINK_NOISE_LP / BP / BR_GAIN
  float bp_octave_width_min, bp_octave_width_max;                   // PINK_NOISE_BP_OCTAVE_WIDTH_MIN / MAX
  float center_octave_min_long_term, center_octave_max_long_term;   // CENTER_OCTAVE_MIN / MAX_LONG_TERM: 10 Hz * 2^value
  float center_short_term_ratio;                                    // CENTER_SHORT_TERM_RATIO
  float seconds_slow_iter_short_term, seconds_slow_iter_long_term;  // SECONDS_SLOW_ITER_*: the parameter x of sample_rate * 2.268^x steps
  int32_t order_filters;                                            // ORDER_FILTERS (1 .. 258)
};

namespace loudness {
// loudness.h:272-274 -> :258-270 -> :231-256 over the float table `freqs`
inline std::pair<int, float> closest_f(float value) {
  int min_ = 0, max_ = n_freq - 1;
  do {
    int center = (min_ + max_) / 2;
    if (value < freqs[center]) max_ = center; else min_ = center;
  } while (max_ - min_ >= 2);
  if (value <= freqs[min_]) return {min_, 1.f};
  if (value >= freqs[max_]) return {max_, 1.f};
  return {max_, (value - freqs[min_]) / (freqs[max_] - freqs[min_])};
}
inline float equal_loudness_volume_from_freq(float freq, int index_freq_ref, float log_ratio, float level) {
  int const i = phons_to_index(level);
  auto const &elv = tables().elvs[i];
  float const max_db = elv[index_freq_ref];
  auto const [k, ratio] = closest_f(freq);
  float const db = (ratio == 1.f) ? elv[k] : ratio * elv[k] + (1.f - ratio) * elv[k - 1];
  if (db > max_db) return 1.f;                                   // db_to_amplitude :246-256
  return std::pow(10.f, log_ratio * (db - max_db) / 20.f);
}
} // namespace loudness

// itp::interpolate(kind, t, b, c, d): Penner's equations (declared, see the header)
inline double itp_interpolate(int kind, double t, double b, double c, double d) {
  switch (kind) {
    default:
    case ITP_LINEAR: return c * t / d + b;
    case ITP_PROPORTIONAL_VALUE_DERIVATIVE: return b * std::pow((b + c) / b, t / d);
    case ITP_EASE_IN_QUAD: t /= d; return c * t * t + b;
    case ITP_EASE_OUT_QUAD: t /= d; return -c * t * (t - 2) + b;
    case ITP_EASE_INOUT_QUAD: t /= d / 2; if (t < 1) return c / 2 * t * t + b; --t; return -c / 2 * (t * (t - 2) - 1) + b;
    case ITP_EASE_IN_CUBIC: t /= d; return c * t * t * t + b;
    case ITP_EASE_OUT_CUBIC: t = t / d - 1; return c * (t * t * t + 1) + b;
    case ITP_EASE_INOUT_CUBIC: t /= d / 2; if (t < 1) return c / 2 * t * t * t + b; t -= 2; return c / 2 * (t * t * t + 2) + b;
    case ITP_EASE_IN_QUART: t /= d; return c * t * t * t * t + b;
    case ITP_EASE_OUT_QUART: t = t / d - 1; return -c * (t * t * t * t - 1) + b;
    case ITP_EASE_INOUT_QUART: t /= d / 2; if (t < 1) return c / 2 * t * t * t * t + b; t -= 2; return -c / 2 * (t * t * t * t - 2) + b;
    case ITP_EASE_IN_QUINT: t /= d; return c * t * t * t * t * t + b;
    case ITP_EASE_OUT_QUINT: t = t / d - 1; return c * (t * t * t * t * t + 1) + b;
    case ITP_EASE_INOUT_QUINT: t /= d / 2; if (t < 1) return c / 2 * t * t * t * t * t + b; t -= 2; return c / 2 * (t * t * t * t * t + 2) + b;
    case ITP_EASE_IN_SINE: return -c * std::cos(t / d * (M_PI / 2)) + c + b;
    case ITP_EASE_OUT_SINE: return c * std::sin(t / d * (M_PI / 2)) + b;
    case ITP_EASE_INOUT_SINE: return -c / 2 * (std::cos(M_PI * t / d) - 1) + b;
    case ITP_EASE_IN_EXPO: return (t == 0) ? b : c * std::pow(2., 10 * (t / d - 1)) + b;
    case ITP_EASE_OUT_EXPO: return (t == d) ? b + c : c * (-std::pow(2., -10 * t / d) + 1) + b;
    case ITP_EASE_INOUT_EXPO:
      if (t == 0) return b;
      if (t == d) return b + c;
      t /= d / 2;
      if (t < 1) return c / 2 * std::pow(2., 10 * (t - 1)) + b;
      --t;
      return c / 2 * (-std::pow(2., -10 * t) + 2) + b;
    case ITP_EASE_IN_CIRC: t /= d; return -c * (std::sqrt(1 - t * t) - 1) + b;
    case ITP_EASE_OUT_CIRC: t = t / d - 1; return c * std::sqrt(1 - t * t) + b;
    case ITP_EASE_INOUT_CIRC: t /= d / 2; if (t < 1) return -c / 2 * (std::sqrt(1 - t * t) - 1) + b; t -= 2; return c / 2 * (std::sqrt(1 - t * t) + 1) + b;
  }
}

// the declared replacement of mersenne<SEEDED::Yes>() on this path (see the header)
struct VoiceRng {
  uint64_t key = 0;
  uint32_t seedval = 0, k = 0;
  void seed(uint32_t s) { seedval = s; k = 0; }
  float u01() {
    uint64_t const h = splitmix64(key ^ splitmix64(((static_cast<uint64_t>(seedval) << 32) | k) * 2 + 5));
    ++k;
    return static_cast<float>(static_cast<uint32_t>(h >> 40)) * (1.0f / 16777216.0f);
  }
  float uniform_real(float a, float b) { return (b - a) * u01() + a; }
  int uniform_int01() { return u01() < 0.5f ? 0 : 1; }
};

// cpp.algorithms' MarkovChain as its call sites show it (see the header)
struct MarkovChain {
  enum Move { ENTER_NODE, LEAVE_NODE };
  struct Node { std::function<void(Move)> f; std::vector<std::pair<float, int>> dest; };
  std::vector<Node> nodes;
  int current = 0;
  int emplace(std::function<void(Move)> f) { nodes.push_back({std::move(f), {}}); return static_cast<int>(nodes.size()) - 1; }
  void def_transition(int a, int b, float p) { nodes[a].dest.emplace_back(p, b); }
  void initialize(int start) { current = start; }
  template <bool EXEC> void step(float proba) {
    float accum = 0.f;
    for (auto const &[p, to] : nodes[current].dest) {
      accum += p;
      if (proba < accum) {
        if (EXEC) { nodes[current].f(LEAVE_NODE); nodes[to].f(ENTER_NODE); }
        current = to;
        break;
      }
    }
  }
  template <bool EXEC> void step_normalized(float proba) {
    float sum = 0.f;
    for (auto const &d : nodes[current].dest) sum += d.first;
    if (sum > 0.f) step<EXEC>(proba * sum);
  }
};

// LogRamp<double> audioelement.h:2464-2572
struct LogRamp {
  double from = 0, to = 0, cur_sample = 0, duration_in_samples = 0, C = 0;
  int interp = ITP_LINEAR;
  void setup(double from_increments, double to_increments, double duration_in_samples_, double start_sample, int i) {
    // start_sample >= 0 on this path (phase_ratio1 * n_frames with a clamped ratio, or 0)
    cur_sample = start_sample;
    from = from_increments;
    to = to_increments;
    duration_in_samples = duration_in_samples_;
    C = (to == from) ? 1.f : -std::log(from / to) / (to - from);
    interp = i;
  }
  double step() { // do_step(C) :2522-2542
    if (cur_sample + .5f > duration_in_samples) { cur_sample = 0; std::swap(from, to); }
    double const f_result = itp_interpolate(interp, cur_sample, from, to - from, duration_in_samples);
    double const f = from + (to - from) * (cur_sample + .5f) / duration_in_samples;
    cur_sample += C * f;
    return f_result;
  }
  double getAngleIncrementFrom() const { return from; }
  double getAngleIncrementTo() const { return to; }
  double get_duration_in_samples() const { return duration_in_samples; }
};

// one ramp of the sound engine
struct VoiceRamp {
  Envelope env;
  LogRamp ctrl;
  // Mix<LoudnessVolumeAdjusted<SineOscillatorAlgo<double>>>
  SineOsc sine;
  std::optional<double> l_volume, l_target; // BaseVolumeAdjusted of the loudness adjustment
  double l_finit = 0;
  OnePoleLowPass l_filter;
  int sample_rate = 0, low_index = 0;
  float log_ratio = 1.f, loudness_level = 0.f;
  float gain0 = 1.f; // Mix::gains
  // the ramp's own VolumeAdjusted
  std::optional<double> volume, volume_target;
  double finit = 0;
  OnePoleLowPass filter;
  StereoGain gain_{{0, 0}};
  // WIND: the control is a ShortTermNoiseAdderCtrl, the Mix has three noise voices before the sine (gains: lp, bp, br; gain0 stays the sine's)
  bool wind = false;
  WindMainCtrl wctrl;
  WindVoices wv;
  float gain_lp = 0.f, gain_bp = 0.f, gain_br = 0.f;

  double mixAngleIncrements() const { // Mix::angleIncrements :1856-1865
    double minAngle = 0;
    if (gain0) minAngle = std::min(minAngle, sine.angleIncrements());
    return minAngle;
  }
  double angleIncrements() const { // FreqCtrl_::angleIncrements :3117-3124
    double const ai = mixAngleIncrements();
    if (ai) return ai;
    return wind ? static_cast<double>(wctrl.getAngleIncrementFrom()) : ctrl.getAngleIncrementFrom();
  }
  void forgetPastSignals() { // StereoPanned -> BaseVolumeAdjusted -> Enveloped -> FreqCtrl_ (LogRamp: nothing) -> Mix -> LoudnessVolumeAdjusted
    env.forgetPastSignals();
    if (wind) { wctrl.forgetPastSignals(); wv.forgetPastSignals(); }
    l_volume.reset(); l_target.reset();
    volume.reset(); volume_target.reset();
  }
  void step() {
    env.step();
    double const inc = wind ? static_cast<double>(wctrl.step()) : ctrl.step();
    // LoudnessVolumeAdjusted::setAngleIncrements :1291-1298 (angle_increment_to_freq, sound.functions.h:64-68, narrowed to the float parameter)
    l_target = loudness::equal_loudness_volume_from_freq(static_cast<float>(inc * 0.5 * sample_rate), low_index, log_ratio, loudness_level);
    sine.setAngleIncrements(inc);
    if (wind) wv.step(); // Mix::step: the voices in order (the noise wrappers ignore setAngleIncrements, the bands hand it to their own, empty, control)
    { // BaseVolumeAdjusted::step of the loudness adjustment :1195-1216
      sine.step();
      if (!l_volume) {
        l_finit = std::min(std::numeric_limits<double>::max(), std::abs(sine.angleIncrements()));
        l_filter.initWithAngleIncrement(l_finit);
        l_filter.setInitialValue(*l_target);
      } else if (*l_volume != *l_target) {
        double const i2 = std::min(std::numeric_limits<double>::max(), std::abs(sine.angleIncrements()));
        if (l_finit != i2) { l_finit = i2; l_filter.initWithAngleIncrement(l_finit); }
      }
      l_filter.feed(*l_target);
      l_volume = l_filter.filtered();
    }
    { // the ramp's VolumeAdjusted
      if (!volume) {
        finit = std::min(std::numeric_limits<double>::max(), std::abs(angleIncrements()));
        filter.initWithAngleIncrement(finit);
        filter.setInitialValue(*volume_target);
      } else if (*volume != *volume_target) {
        double const i2 = std::min(std::numeric_limits<double>::max(), std::abs(angleIncrements()));
        if (finit != i2) { finit = i2; filter.initWithAngleIncrement(finit); }
      }
      filter.feed(*volume_target);
      volume = filter.filtered();
    }
  }
  double mixImag() const { // Mix::imag :1867-1873
    double sum = 0.f;
    if (wind) {
      // gain_lp * LowPassAlgo<PinkNoiseAlgo>::imag(): 0 * (the output of a filter nobody configured, taken as finite) — a non-zero gain is refused at setup
      sum += gain_bp * wv.bp.imag();
      sum += gain_br * wv.br.imag();
    }
    sum += gain0 * (*l_volume * sine.imag());
    return sum;
  }
  double imag(int i) const { return gain_.gains[i] * (*volume * (mixImag() * env.value())); }
};

// one line of the trace the CPU tests compare the product's note expansion with: playNextSpec as it happened
struct VoicePlayed { int64_t step; int32_t ramp, spec; };

// VolumeAdjusted<SoundEngine<MODE, 2, ...>>
struct VoiceElement {
  void set_sample_rate(int s) { sample_rate = s; for (auto &r : ramps) r.sample_rate = s; } // gen.crtp.h:424 (the initializer sets the same value again)
  std::array<VoiceRamp, 3> ramps;
  // SoundEngine
  int mode = VOICE_SWEEP;
  unsigned oddOn = 0;
  bool rt_active = false;
  double start_angle = 0;
  int xfade_len = 0, freq_xfade = 0, articulative_pause_length = 0, remaining_silence_steps = 0, sample_rate = 0;
  int interpolation = ITP_LINEAR, freq_interpolation = ITP_EASE_IN_EXPO, xfade_freq = XFADE_NO;
  float d1 = 0, d2 = 0, har_att = 0, length = 0, base_freq = 0, freq_scatter = 0, phase_ratio1 = 0, phase_ratio2 = 0, min_exp = 0, max_exp = 0;
  float state_freq = 0, state_factor = 0;
  float freq1_robot = 0, freq2_robot = 0, vol1 = 1.f, vol2 = 1.f;
  VoiceRng *rng = nullptr;          // mersenne<SEEDED::Yes>(): one engine per synth here
  WindDrawCtx wind_ctx;             // WIND: what BufferIter's start positions are drawn for (oracle_wind.hpp)
  std::unique_ptr<MarkovChain> markov;
  // setupAudioElement's arguments to initialize_*, handed over by the initializer (the reference passes them in one call)
  float low_freq = 0, high_freq = 0;
  int start_node = 0, pre_tries = 0, min_path_length = 0, additional_tries = 0, pause_param = 0;
  bool setup_ok = true;
  // RampSpecs :1035-1120
  struct Spec { LogRamp ctrl; WindMainCtrl wctrl; float vol = 1.f; bool silenceAfter = true; };
  struct RampSpecs {
    static constexpr int n_specs = 30;
    int it = -1, end = 0; // (`it` is a 5-bit unsigned in the reference: its -1 is 31 >= end)
    std::array<Spec, n_specs> a;
    void reset() { it = -1; end = 0; }
    bool done() const { return it == end; }
    Spec *get_next_ramp_for_build() {
      ++it;
      if (it == n_specs) { --it; return nullptr; }
      ++end;
      return &a[it];
    }
    void cancel_last_ramp() { --it; --end; }
    Spec *get_current() { return (it < 0 || it >= end) ? nullptr : &a[it]; }
    void finalize() { it = -1; }
    Spec *get_next_ramp_for_run() {
      ++it;
      if (it == end) return nullptr;
      return &a[it];
    }
  } ramp_specs;
  // the outer BaseVolumeAdjusted
  std::optional<double> volume, volume_target;
  double finit = 0;
  OnePoleLowPass filter;
  // trace (tests)
  std::vector<VoicePlayed> *trace = nullptr;
  int64_t steps_taken = 0;
  // gen.crtp's view of the envelope (`env` names what ImplCRTP reaches through getEnvelope())
  struct EnvView {
    VoiceElement *e;
    bool isEnvelopeRTActive() const { return e->rt_active && !e->isEnvelopeFinished(); }
    bool tryAcquire() { if (!e->isEnvelopeFinished()) return false; ++e->oddOn; return true; } // soundengine.h:301-317, one thread
    bool acquireStates() const { return e->oddOn & 1u; }
    void onKeyPressed(int32_t) { e->rt_active = true; }
    bool canHandleExplicitKeyReleaseNow(int32_t) const { return e->goOn() != 0; }
    void onKeyReleased(int32_t delay) {
      e->rt_active = false;
      if (unsigned i = e->goOn()) { e->oddOn = i + 1; for (auto &r : e->ramps) r.env.onKeyReleased(delay); }
    }
  } env{this};
  VoiceElement() = default;
  VoiceElement(VoiceElement const &) = delete;             // (the chain's lambdas and `env` point at this object)
  VoiceElement &operator=(VoiceElement const &) = delete;
  unsigned goOn() const { return (oddOn & 1u) ? oddOn : 0; }
  bool isEnvelopeFinished() const { // isEnvelopeFinished_internal :470-480
    if (oddOn & 1u) return false;
    for (auto const &r : ramps) if (r.env.isEnvelopeRTActive()) return false;
    return true;
  }
  void forgetPastSignals() { // VolumeAdjusted -> SoundEngine :324-327
    remaining_silence_steps = 0;
    start_angle = 0;
    volume.reset(); volume_target.reset();
    steps_taken = 0;
  }
  void setVolumeTarget(double v) { volume_target = v; }
  void setStartAngle(double a) { start_angle = a; }
  void synchronizeAnglesFrom(VoiceElement const &) {} // SynchronizePhase::No
  // engine.setAngleIncrements (:393-395), then — last step of setupAudioElement, gen.voice.h:999-1034 — engine.initialize_<mode>
  void setAngleIncrements(double ai) {
    base_freq = static_cast<float>(ai * 0.5 * sample_rate);
    switch (mode) {
      case VOICE_SWEEP: setup_ok = initialize_sweep(low_freq, high_freq); break;
      case VOICE_BIRDS: setup_ok = initialize_birds(); break;
      case VOICE_WIND: setup_ok = initialize_wind(); break;
      default: setup_ok = initialize_robot(); break;
    }
  }
  double angleIncrements() const { // :397-410
    for (auto const &r : ramps)
      if (r.env.state == EnvelopeState::KeyReleased || r.env.state == EnvelopeState::KeyPressed) return r.angleIncrements();
    return 0;
  }
  float rand_0_1() { return rng->uniform_real(0.f, 1.f); }

  // play :497-562 (BIRDS)
  void play(float length_, float freq1, float freq2, float /*phase_ratio1*/, float /*phase_ratio2*/, float freq_scatter_) {
    length_ *= std::pow(2.f, rng->uniform_real(min_exp, max_exp));
    float n_frames = static_cast<float>(ms_to_frames(length_, sample_rate));
    if (n_frames <= 0) return;
    Spec *current = ramp_specs.get_current();
    if (Spec *ramp_spec = ramp_specs.get_next_ramp_for_build()) {
      if (state_freq == freq1) {
        // use previous scatter when the markov transitions specify the same base value
      } else {
        float scatter = 1.f + freq_scatter_;
        state_factor = rng->uniform_real(1.f / scatter, scatter);
      }
      state_freq = freq2;
      freq1 *= state_factor;
      freq2 *= state_factor;
      ramp_spec->ctrl.setup(freq_to_angle_increment(freq1, sample_rate), freq_to_angle_increment(freq2, sample_rate), n_frames, 0, interpolation);
      ramp_spec->silenceAfter = true;
      ramp_spec->vol = 1.f;
      if (xfade_freq == XFADE_NO) return;
      if (current) {
        double from_inc = current->ctrl.getAngleIncrementTo();
        double to_inc = ramp_spec->ctrl.getAngleIncrementFrom();
        double diff = from_inc - to_inc;
        if (xfade_freq == XFADE_ALL || diff) {
          if (Spec *ramp_spec2 = ramp_specs.get_next_ramp_for_build()) {
            *ramp_spec2 = *ramp_spec;
            if (from_inc == to_inc) from_inc *= 1.00001f;
            ramp_spec->ctrl.setup(from_inc, to_inc, freq_xfade, 0, freq_interpolation);
            ramp_spec->silenceAfter = true;
            ramp_spec->vol = 1.f;
          } else {
            ramp_specs.cancel_last_ramp();
          }
        }
      }
    }
  }
  std::unique_ptr<MarkovChain> create_birds() { // :565-595
    auto mc = std::make_unique<MarkovChain>();
    int node1 = mc->emplace([](MarkovChain::Move) {});
    int node2 = mc->emplace([this](MarkovChain::Move m) {
      if (m == MarkovChain::ENTER_NODE) play(length, base_freq * 4, base_freq * 3, phase_ratio1, phase_ratio2, freq_scatter);
      else play(length, base_freq * 2, base_freq * 4, phase_ratio1, phase_ratio2, freq_scatter);
    });
    int node3 = mc->emplace([this](MarkovChain::Move m) {
      if (m == MarkovChain::ENTER_NODE) play(length, base_freq * 4, base_freq * 3, phase_ratio1, phase_ratio2, freq_scatter);
    });
    mc->def_transition(node1, node2, 0.5f);
    mc->def_transition(node2, node1, 0.015f);
    mc->def_transition(node1, node3, 0.5f);
    mc->def_transition(node3, node1, 0.015f);
    mc->def_transition(node3, node2, 0.885f);
    return mc;
  }
  double transpose_frequency(double freq, int n) const { return freq * std::pow(std::pow(2., 1. / 12.), static_cast<double>(n)); } // midi.h:243-247,266-269
  float drawn_n_frames(float len) { // the lambdas' common head :602-605
    len *= std::pow(2.f, rng->uniform_real(min_exp, max_exp));
    return static_cast<float>(ms_to_frames(len, sample_rate));
  }
  void build_spec(double from, double to, float n_frames, float vol, bool silence) {
    if (Spec *s = ramp_specs.get_next_ramp_for_build()) {
      s->ctrl.setup(from, to, n_frames, phase_ratio1 * n_frames, interpolation);
      s->vol = vol;
      s->silenceAfter = silence;
    }
  }
  std::unique_ptr<MarkovChain> create_robot() { // :597-695
    auto mc = std::make_unique<MarkovChain>();
    int node0 = mc->emplace([this](MarkovChain::Move m) {
      if (m == MarkovChain::LEAVE_NODE) {
        float n_frames = drawn_n_frames(length);
        build_spec(freq_to_angle_increment(freq1_robot, sample_rate), freq_to_angle_increment(freq1_robot, sample_rate), n_frames, vol1, false);
        build_spec(freq_to_angle_increment(freq2_robot, sample_rate), freq_to_angle_increment(freq2_robot, sample_rate), n_frames, vol2, true);
      }
    });
    int node1 = mc->emplace([](MarkovChain::Move) {});
    int node2 = mc->emplace([this](MarkovChain::Move m) {
      if (m == MarkovChain::ENTER_NODE) {
        float n_frames = drawn_n_frames(length);
        build_spec(freq_to_angle_increment(freq2_robot, sample_rate), freq_to_angle_increment(freq2_robot, sample_rate), n_frames, vol2, true);
        { double inc = 2 * transpose_frequency(freq2_robot, 2) / sample_rate; build_spec(inc, inc, n_frames, vol2, true); } // freq_to_angle_increment<double>
        { double inc = 2 * transpose_frequency(freq2_robot, 4) / sample_rate; build_spec(inc, inc, n_frames, vol2, true); }
      }
    });
    int node3 = mc->emplace([this](MarkovChain::Move m) {
      if (m == MarkovChain::ENTER_NODE) {
        constexpr float slide_length_scale = 2.f;
        float n_frames = drawn_n_frames(slide_length_scale * length);
        build_spec(freq_to_angle_increment(freq2_robot, sample_rate), freq_to_angle_increment(freq1_robot, sample_rate), n_frames, std::min(vol2, vol1), true);
      }
    });
    mc->def_transition(node0, node1, 1.f);
    mc->def_transition(node1, node2, 0.2f);
    mc->def_transition(node2, node1, 0.1f);
    mc->def_transition(node1, node3, 0.2f);
    mc->def_transition(node3, node1, 1.f);
    return mc;
  }
  std::unique_ptr<MarkovChain> create_sweep() { // :697-719 (the spec keeps its defaults: volume 1, silence follows)
    auto mc = std::make_unique<MarkovChain>();
    int node0 = mc->emplace([this](MarkovChain::Move m) {
      if (m == MarkovChain::LEAVE_NODE) {
        float n_frames = drawn_n_frames(length);
        if (Spec *s = ramp_specs.get_next_ramp_for_build())
          s->ctrl.setup(freq_to_angle_increment(freq1_robot, sample_rate), freq_to_angle_increment(freq2_robot, sample_rate), n_frames, phase_ratio1 * n_frames, interpolation);
      }
    });
    int node1 = mc->emplace([](MarkovChain::Move) {});
    mc->def_transition(node0, node1, 1.f);
    return mc;
  }
  std::unique_ptr<MarkovChain> create_wind() { // :721-735: the lambda runs on ENTER and on LEAVE alike
    auto mc = std::make_unique<MarkovChain>();
    int node0 = mc->emplace([this](MarkovChain::Move) {
      if (Spec *s = ramp_specs.get_next_ramp_for_build()) s->wctrl.ctrl.ctrl.it.set_interpolation(interpolation); // getUnderlyingIter(): the long-term SlowIter
    });
    int node1 = mc->emplace([](MarkovChain::Move) {});
    mc->def_transition(node0, node1, 1.f);
    mc->def_transition(node1, node0, 1.f);
    return mc;
  }
  bool initialize_wind() { // :913-924 with SoundEngineInitPolicy::StartAfresh (gen.voice.h:1017-1022); the pause keeps the member's value (0)
    if (!markov) markov = create_wind();
    return do_initialize(true, start_node, pre_tries, min_path_length, additional_tries, articulative_pause_length);
  }
  bool initialize_sweep(float low, float high) { // :883-896
    if (!markov) markov = create_sweep();
    freq1_robot = low;
    freq2_robot = high;
    return do_initialize(true, 0, 0, 1, 0, articulative_pause_length);
  }
  bool initialize_birds() { // :898-911 with SoundEngineInitPolicy::StartAfresh (gen.voice.h:1013)
    if (!markov) markov = create_birds();
    return do_initialize(true, start_node, pre_tries, min_path_length, additional_tries, pause_param);
  }
  bool initialize_robot() { // :926-961
    float scatter = 1.f + freq_scatter;
    constexpr float detune = 0.985f;
    freq1_robot = rng->uniform_real(base_freq / scatter, base_freq * scatter);
    freq2_robot = rng->uniform_real(freq1_robot * detune, freq1_robot / detune);
    vol1 = vol2 = 1.f;
    if (!rng->uniform_int01()) {
      freq1_robot = static_cast<float>(transpose_frequency(freq1_robot, static_cast<int>(d1)));
      vol1 = std::pow(har_att, d1);
    } else {
      freq2_robot = static_cast<float>(transpose_frequency(freq2_robot, static_cast<int>(d2)));
      vol2 = std::pow(har_att, d2);
    }
    float n_frames = static_cast<float>(ms_to_frames(length, sample_rate));
    if (n_frames <= 0) return false;
    if (!markov) markov = create_robot();
    return do_initialize(true, start_node, pre_tries, min_path_length, additional_tries, pause_param);
  }
  bool do_initialize(bool initialize, int start_node_, int pre_tries_, int min_path_length_, int additional_tries_, int articulative_pause_length_) { // :963-1003
    float n_frames = static_cast<float>(ms_to_frames(length, sample_rate));
    if (n_frames <= 0) return false;
    if (base_freq <= 0.f) return false;
    state_freq = 0.f;
    state_factor = 0.f;
    ramp_specs.reset();
    articulative_pause_length = articulative_pause_length_;
    if (initialize) {
      markov->initialize(start_node_);
      for (int i = 0; i < pre_tries_; ++i) markov->step_normalized<false>(rand_0_1());
    }
    for (int i = 0; i < min_path_length_; ++i) markov->step_normalized<true>(rand_0_1());
    for (int i = 0; i < additional_tries_; ++i) markov->step<true>(rand_0_1());
    ramp_specs.finalize();
    return true;
  }
  void playNextSpec() { // :776-803
    VoiceRamp *new_ramp = nullptr;
    for (auto &r : ramps) if (r.env.state == EnvelopeState::EnvelopeDone2 && goOn()) new_ramp = &r; // get_ramps :277-296: the last one wins
    if (!new_ramp) return;
    Spec *new_spec = ramp_specs.get_next_ramp_for_run();
    if (!new_spec) return;
    if (trace) trace->push_back({steps_taken, static_cast<int32_t>(new_ramp - ramps.data()), static_cast<int32_t>(ramp_specs.it)});
    new_ramp->ctrl = new_spec->ctrl;
    if (mode == VOICE_WIND) { new_ramp->wctrl = new_spec->wctrl; wind_ctx.play = ramp_specs.it; } // (the draws of this play: oracle_wind.hpp)
    new_ramp->forgetPastSignals();
    new_ramp->sample_rate = sample_rate;
    if (mode == VOICE_WIND) { new_ramp->wctrl.set_sample_rate(sample_rate); new_ramp->wv.set_sample_rate(sample_rate); new_ramp->wv.setStartAngle(start_angle); wind_ctx.play = -1; }
    new_ramp->sine.setStartAngle(start_angle);
    new_ramp->volume_target = new_spec->vol;
    new_ramp->env.setAHDSR(AHDSR{xfade_len, 0, 0, xfade_len, 1.f}, sample_rate);
    new_ramp->env.tryAcquire();
    int const time_to_release = (mode == VOICE_WIND ? static_cast<int>(.5f + new_ramp->wctrl.get_duration_in_samples()) // (.5f + an int: a float sum)
                                                    : static_cast<int>(.5f + new_ramp->ctrl.get_duration_in_samples())) - xfade_len;
    new_ramp->env.onKeyPressed(0);
    new_ramp->env.onKeyReleased(std::max(time_to_release, 0));
  }
  void orchestrate_algos() { // :744-774
    if (ramp_specs.done()) return;
    Spec *cur = ramp_specs.get_current();
    VoiceRamp const *pressed = nullptr;
    for (auto const &r : ramps) if (r.env.state == EnvelopeState::KeyPressed) pressed = &r;
    if (pressed) {
      int const remaining = pressed->env.currentReleaseDelay; // countStepsToRelease, audioelement.h:783-785
      if (remaining > 0) return;
      if (remaining == 0 && cur && cur->silenceAfter) remaining_silence_steps = articulative_pause_length;
    }
    if (cur && cur->silenceAfter) {
      if (remaining_silence_steps > 0) { --remaining_silence_steps; return; }
    }
    playNextSpec();
  }
  void step() { // BaseVolumeAdjusted::step over SoundEngine::step :738-741
    ++steps_taken;
    orchestrate_algos();
    for (auto &r : ramps) if (r.env.state != EnvelopeState::EnvelopeDone2) r.step(); // step_algos :412-425
    if (!volume) {
      finit = std::min(std::numeric_limits<double>::max(), std::abs(angleIncrements()));
      filter.initWithAngleIncrement(finit);
      filter.setInitialValue(*volume_target);
    } else if (*volume != *volume_target) {
      double const i2 = std::min(std::numeric_limits<double>::max(), std::abs(angleIncrements()));
      if (finit != i2) { finit = i2; filter.initWithAngleIncrement(finit); }
    }
    filter.feed(*volume_target);
    volume = filter.filtered();
  }
  double imag(int i) const { // BaseVolumeAdjusted::imag(i) over SoundEngine::imag(i) :447-464
    double v = 0;
    for (auto const &r : ramps)
      if (r.env.state == EnvelopeState::KeyReleased || r.env.state == EnvelopeState::KeyPressed) v += r.imag(i);
    return *volume * v;
  }
};
struct VoiceElemTraits {
  // VolumeAdjusted<SoundEngine>::baseVolume = SoundEngine::baseVolume = the ramp element's = Mix's minimum = LoudnessVolumeAdjusted's:
  // SineOscillatorAlgo::baseVolume / reduceUnadjustedVolumes (audioelement.h:1279) = (0.1f * 1.) / 0.1f
  static constexpr double baseVolume = sineBaseVolume / reduceUnadjustedVolumes;
  static constexpr uint64_t phase_salt = 4; // unused: DefaultStartPhase::Zero
};

// the reference's clamps of the engine setters (soundengine.h:19-29,827-837)
inline float clamp_phase_ratio(float v) { return v > 1.f ? 1.f : (v < 0.f ? 0.f : v); }
inline float clamp_har_att(float v) { return v > 0.99f ? 0.99f : (v < 0.f ? 0.f : v); }

// SynthImpl of gen.voice (ImplBase<MODE>::setupAudioElement, gen.voice.h:883-1035) as the synchronous initializer + setup of one note
struct VoiceInitializer {
  ora_voice_params p{};
  std::shared_ptr<VoiceRng> rng = std::make_shared<VoiceRng>();
  std::shared_ptr<int> seed_counter = std::make_shared<int>(0); // `thread_local int seed`, gen.voice.h:910
  std::vector<VoicePlayed> *trace = nullptr;
  bool operator==(VoiceInitializer const &) const { return true; }
  void operator()(VoiceElement &e, bool, int64_t noteid) const {
    // c.elem.set_sample_rate precedes the volume target in onEvent (gen.crtp.h:423-424)
    e.mode = p.mode;
    e.rng = rng.get();
    e.trace = trace;
    rng->key = p.seed;
    e.sample_rate = p.sample_rate;
    for (auto &r : e.ramps) r.sample_rate = p.sample_rate;
    e.interpolation = (p.interpolation >= 0 && p.interpolation < ITP_END) ? p.interpolation : ITP_LINEAR; // set_itp :841-846
    e.xfade_len = static_cast<int>(0.5f + (p.xfade_length_seconds * p.sample_rate));  // setEnvelopeCharacTime :893
    float const ex = p.length_exponent;
    if (p.mode != VOICE_SWEEP) {
      float const variation = p.length_exponent_scatter;
      e.min_exp = ex * (1.f - variation); e.max_exp = ex * (1.f + variation);         // :901
      e.freq_scatter = p.freq_scatter;
      e.phase_ratio1 = clamp_phase_ratio(p.phase_ratio1);
      e.phase_ratio2 = clamp_phase_ratio(p.phase_ratio2);
      int &seed = *seed_counter;                                                       // :910-925
      if (p.seed_param) seed = p.seed_param;
      else if (seed == 257) seed = 1;
      else ++seed;
      rng->seed(static_cast<uint32_t>(seed));
    } else {
      e.min_exp = e.max_exp = ex;                                                      // set_length_exp(ex, ex) :927-929
      e.phase_ratio1 = 0.f;
    }
    e.length = p.length_ms;                                                            // :933 (set after setAngleIncrements in the reference; no dependence)
    e.low_freq = p.low_freq; e.high_freq = p.high_freq;
    for (auto &r : e.ramps) { r.low_index = p.loudness_ref_freq_index; r.log_ratio = p.loudness_compensation; r.loudness_level = p.loudness_level; r.gain0 = p.sine_gain; }
    float pan;
    if (p.random_pan) pan = (p.mode == VOICE_SWEEP) ? hash_uniform_pm1(p.seed, static_cast<uint64_t>(noteid), 3) : rng->uniform_real(-1.f, 1.f); // :980-995
    else pan = p.pan;
    for (auto &r : e.ramps) r.gain_ = stereo(pan);
    e.start_node = p.markov_start_node; e.pre_tries = p.markov_pre_tries; e.min_path_length = p.markov_min_path_length; e.additional_tries = p.markov_additional_tries;
    e.pause_param = static_cast<int>(p.articulative_pause_seconds * p.sample_rate);    // denorm<...>() * sample_rate into the int parameter :1015,1033
    if (p.mode == VOICE_BIRDS) {
      e.freq_xfade = static_cast<int>(p.freq_transition_seconds * p.sample_rate);       // set_freq_xfade(int) :1004
      e.freq_interpolation = (p.freq_interpolation >= 0 && p.freq_interpolation < ITP_END) ? p.freq_interpolation : ITP_EASE_IN_EXPO; // :848-853
      e.xfade_freq = p.xfade_freq;
    } else {
      e.xfade_freq = XFADE_NO;                                                          // the constructor's value :271
    }
    if (p.mode == VOICE_ROBOTS) { e.d1 = p.d1; e.d2 = p.d2; e.har_att = clamp_har_att(p.harmonic_attenuation); }
    if (p.mode == VOICE_WIND) setup_wind(e, noteid);
  }
  // setupAudioElement's WIND branches (gen.voice.h:936-978) over SetFilterWidths / ConfigureFilters / SetGains / SetSlowParams (:334-408)
  void setup_wind(VoiceElement &e, int64_t noteid) const {
    int const sr = p.sample_rate;
    e.wind_ctx.key = p.seed; e.wind_ctx.noteid = noteid; e.wind_ctx.play = -1;
    auto bind = [&e](PinkIter &it, int id) { it.ctx = &e.wind_ctx; it.id = id; };
    for (auto &r : e.ramps) {
      r.wind = true;
      bind(r.wctrl.ctrl.ctrl.it.it, WI_LONG); bind(r.wctrl.noise.it, WI_SHORT);
      bind(r.wv.bp.center.ctrl.it.it, WI_BP_CENTER); bind(r.wv.bp.width.it.it, WI_BP_WIDTH);
      bind(r.wv.br.center.ctrl.it.it, WI_BR_CENTER); bind(r.wv.br.width.it.it, WI_BR_WIDTH);
      r.wctrl.set_sample_rate(sr); r.wv.set_sample_rate(sr);       // SoundEngine::set_sample_rate :484-492 (gen.crtp.h:424)
      r.wv.setFiltersOrder(p.order_filters);                       // engine.setFiltersOrder(value<ORDER_FILTERS>()) :939
      float const m = p.bp_octave_width_min, M = p.bp_octave_width_max; // :941-950
      r.wv.bp.width_min = r.wv.br.width_min = std::min(m, M);
      r.wv.bp.width_max = r.wv.br.width_max = std::max(m, M);
      r.gain_lp = p.pink_noise_lp_gain; r.gain_bp = p.pink_noise_bp_gain; r.gain_br = p.pink_noise_br_gain; // SetGains<WIND> :399-406 (gain0 = SINE_GAIN)
    }
    for (auto &s : e.ramp_specs.a) { bind(s.wctrl.ctrl.ctrl.it.it, WI_LONG); bind(s.wctrl.noise.it, WI_SHORT); s.wctrl.set_sample_rate(sr); }
    // octaveRangeToAngleIncrementsRange<CENTER_OCTAVE_MIN_LONG_TERM, CENTER_OCTAVE_MAX_LONG_TERM> :854-879
    constexpr float lowest_f = 10.f;
    float fm = lowest_f * std::pow(2.f, p.center_octave_min_long_term), fM = lowest_f * std::pow(2.f, p.center_octave_max_long_term);
    if (fm > fM) std::swap(fm, fM);
    float const ra_min = freq_to_angle_increment(fm, sr), ra_max = freq_to_angle_increment(fM, sr);
    constexpr float max_seconds_slow_iter = 2.268f; // :623
    float const n_slow_steps = sr * std::pow(max_seconds_slow_iter, p.seconds_slow_iter_long_term);
    float const n_slow_steps_short = sr * std::pow(max_seconds_slow_iter, p.seconds_slow_iter_short_term);
    for (auto &r : e.ramps) { // ConfigureFilters<WIND>::configure(mix, unsigned int n_slow_steps, ra) :367-388
      unsigned int const n = static_cast<unsigned int>(n_slow_steps);
      for (WindBand *b : {&r.wv.bp, &r.wv.br}) {
        b->center.ctrl.it.set_n_slow_steps(static_cast<int32_t>(n));
        b->center.setAngleIncrementsRange(ra_min, ra_max);
        b->width.it.set_n_slow_steps(static_cast<int32_t>(n));
      }
    }
    for (auto &s : e.ramp_specs.a) { // SetSlowParams<WIND>::set(ctrl, int n_slow_steps_short, int n_slow_steps_long, float ratio) :332-341, then the range :976
      s.wctrl.ctrl.ctrl.it.set_n_slow_steps(static_cast<int>(n_slow_steps));
      s.wctrl.ratio = static_cast<float>(static_cast<int>(n_slow_steps_short)); // set_short_term_noise_rate(float) fed the int
      s.wctrl.noise_amplitude = p.center_short_term_ratio;                      // set_short_term_noise_amplitude
      s.wctrl.ctrl.setAngleIncrementsRange(ra_min, ra_max);
    }
  }
};

struct VoiceSynth : SynthT<VoiceElement, VoiceInitializer, 2, VoiceElemTraits> {
  using SynthT::SynthT;
};

} // namespace ora
