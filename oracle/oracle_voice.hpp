/*
 * TEST INFRASTRUCTURE — CPU oracle, part 6 (first slice): gen.voice in SWEEP mode.
 *
 * voice::Impl_<..., Mode::SWEEP, ...> (include/gen.voice.h:1087-1190) = ImplCRTP (include/gen.crtp.h, SynchronizePhase::No,
 * DefaultStartPhase::Zero) over VolumeAdjusted<SoundEngine<SWEEP, 2, ...>> (include/soundengine.h:240-1127), whose three "ramps" are
 *   StereoPanned<VolumeAdjusted<Enveloped<FreqCtrl_<Mix<LoudnessVolumeAdjusted<SineOscillatorAlgo<double>>>, LogRamp<double>>, AHDSREnvelope>>>
 * (soundengine.h:175-180,205-211,246-258; audioelement.h:1784-1880 Mix, :1275-1317 LoudnessVolumeAdjusted, :2464-2572 LogRamp).
 *
 * What SWEEP needs from the absent cpp.algorithms, and what stands in for it:
 *   - MarkovChain: create_sweep (soundengine.h:697-720) is two nodes joined by a transition of probability 1, and initialize_sweep runs
 *     exactly one normalised step from node 0 (do_initialize(true, 0, 0, 1, 0, ...), :905-918, 1001-1044): whatever number is drawn, the
 *     chain moves 0 -> 1 and runs node 0's LEAVE lambda, which builds the one ramp spec.  No chain is needed.
 *   - the two mersenne<> draws on that path are inert: rand_0_1() feeds the certain transition, and uniform{min_exp, max_exp} has
 *     min_exp == max_exp in SWEEP mode (set_length_exp(ex, ex), gen.voice.h:927-929): it returns ex.  The voice is deterministic.
 *   - NormalizedInterpolation: LINEAR only (the SWEEP programs use itp::LINEAR, gen.voice.h:752-757), from + (to - from) * (t / d) as on
 *     the rest of this path.
 *   - random pan (gen.voice.h:980-995) draws from mersenne<>: the declared counter hash, salt 3; or the fixed PAN parameter.
 * Reference quirks restated as they are: Mix::angleIncrements() starts its minimum at 0 (audioelement.h:1856-1865), so for positive
 * increments FreqCtrl_::angleIncrements() falls back to the control's `from`; a note-off makes the element inactive at once
 * (SoundEngine::onKeyReleased clears rt_active, soundengine.h:345-353: "TODO we should delay this").
 */
#pragma once
#include "oracle_synth.hpp"

namespace ora {

struct ora_voice_params { // the SWEEP program's parameters, de-normalised (gen.voice.h:186-208, 474-529, 560-580)
  int32_t sample_rate;
  int32_t n_channels;           // 2 * nVoices
  float length_ms;              // LENGTH
  float length_exponent;        // LENGTH_EXPONENT
  float xfade_length_seconds;   // XFADE_LENGTH
  float low_freq, high_freq;    // LOW_FREQ, HIGH_FREQ
  float sine_gain;              // SINE_GAIN
  float loudness_level;         // LOUDNESS_LEVEL
  float loudness_compensation;  // LOUDNESS_COMPENSATION_AMOUNT (log_ratio)
  int32_t loudness_ref_freq_index;
  float pan;                    // PAN
  int32_t random_pan;           // RANDOM_PAN (as a boolean: 1 = draw the pan per note)
  uint64_t seed;
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

// LogRamp<double> audioelement.h:2464-2572
struct LogRamp {
  double from = 0, to = 0, cur_sample = 0, duration_in_samples = 0, C = 0;
  void setup(double from_increments, double to_increments, double duration_in_samples_, double start_sample) {
    // start_sample >= 0 on this path (phase_ratio1 * n_frames, soundengine.h:709)
    cur_sample = start_sample;
    from = from_increments;
    to = to_increments;
    duration_in_samples = duration_in_samples_;
    C = (to == from) ? 1.f : -std::log(from / to) / (to - from);
  }
  double step() { // do_step(C) :2522-2542
    if (cur_sample + .5f > duration_in_samples) { cur_sample = 0; std::swap(from, to); }
    double const f_result = from + (to - from) * (cur_sample / duration_in_samples);
    double const f = from + (to - from) * (cur_sample + .5f) / duration_in_samples;
    cur_sample += C * f;
    return f_result;
  }
  double getAngleIncrementFrom() const { return from; }
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

  double mixAngleIncrements() const { // Mix::angleIncrements :1856-1865
    double minAngle = 0;
    if (gain0) minAngle = std::min(minAngle, sine.angleIncrements());
    return minAngle;
  }
  double angleIncrements() const { // FreqCtrl_::angleIncrements :3117-3124
    double const ai = mixAngleIncrements();
    if (ai) return ai;
    return ctrl.getAngleIncrementFrom();
  }
  void forgetPastSignals() { // StereoPanned -> BaseVolumeAdjusted -> Enveloped -> FreqCtrl_ (LogRamp: nothing) -> Mix -> LoudnessVolumeAdjusted
    env.forgetPastSignals();
    l_volume.reset(); l_target.reset();
    volume.reset(); volume_target.reset();
  }
  void step() {
    env.step();
    double const inc = ctrl.step();
    // LoudnessVolumeAdjusted::setAngleIncrements :1291-1298 (angle_increment_to_freq, sound.functions.h:64-68, narrowed to the float parameter)
    l_target = loudness::equal_loudness_volume_from_freq(static_cast<float>(inc * 0.5 * sample_rate), low_index, log_ratio, loudness_level);
    sine.setAngleIncrements(inc);
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
    sum += gain0 * (*l_volume * sine.imag());
    return sum;
  }
  double imag(int i) const { return gain_.gains[i] * (*volume * (mixImag() * env.value())); }
};

// VolumeAdjusted<SoundEngine<SWEEP, 2, ...>>
struct VoiceElement {
  std::array<VoiceRamp, 3> ramps;
  // SoundEngine
  unsigned oddOn = 0;
  bool rt_active = false;
  double start_angle = 0;
  int xfade_len = 0, articulative_pause_length = 0, remaining_silence_steps = 0, sample_rate = 0;
  float length = 0, base_freq = 0, min_exp = 0, max_exp = 0, freq1 = 0, freq2 = 0, phase_ratio1 = 0;
  // RampSpecs with its single spec
  struct Spec { LogRamp ctrl; float vol = 1.f; bool silenceAfter = true; } spec;
  int spec_it = -1, spec_end = 0;
  // the outer BaseVolumeAdjusted
  std::optional<double> volume, volume_target;
  double finit = 0;
  OnePoleLowPass filter;
  // gen.crtp's view of the envelope (`env` names what ImplCRTP reaches through getEnvelope())
  struct EnvView {
    VoiceElement *e;
    bool isEnvelopeRTActive() const { return e->rt_active && !e->isEnvelopeFinished(); }
    bool tryAcquire() { if (!e->isEnvelopeFinished()) return false; ++e->oddOn; return true; } // soundengine.h:291-308, one thread
    bool acquireStates() const { return e->oddOn & 1u; }
    void onKeyPressed(int32_t) { e->rt_active = true; }
    bool canHandleExplicitKeyReleaseNow(int32_t) const { return e->goOn() != 0; }
    void onKeyReleased(int32_t delay) {
      e->rt_active = false;
      if (unsigned i = e->goOn()) { e->oddOn = i + 1; for (auto &r : e->ramps) r.env.onKeyReleased(delay); }
    }
  } env{this};
  VoiceElement() = default;
  VoiceElement(VoiceElement const &o) { *this = o; }
  VoiceElement &operator=(VoiceElement const &o) {
    ramps = o.ramps; oddOn = o.oddOn; rt_active = o.rt_active; start_angle = o.start_angle; xfade_len = o.xfade_len;
    articulative_pause_length = o.articulative_pause_length; remaining_silence_steps = o.remaining_silence_steps; sample_rate = o.sample_rate;
    length = o.length; base_freq = o.base_freq; min_exp = o.min_exp; max_exp = o.max_exp; freq1 = o.freq1; freq2 = o.freq2; phase_ratio1 = o.phase_ratio1;
    low_freq = o.low_freq; high_freq = o.high_freq; setup_ok = o.setup_ok;
    spec = o.spec; spec_it = o.spec_it; spec_end = o.spec_end; volume = o.volume; volume_target = o.volume_target; finit = o.finit; filter = o.filter;
    env.e = this;
    return *this;
  }
  unsigned goOn() const { return (oddOn & 1u) ? oddOn : 0; }
  bool isEnvelopeFinished() const { // isEnvelopeFinished_internal :457-467
    if (oddOn & 1u) return false;
    for (auto const &r : ramps) if (r.env.isEnvelopeRTActive()) return false;
    return true;
  }
  void forgetPastSignals() { // VolumeAdjusted -> SoundEngine :314-317
    remaining_silence_steps = 0;
    start_angle = 0;
    volume.reset(); volume_target.reset();
  }
  void setVolumeTarget(double v) { volume_target = v; }
  void setStartAngle(double a) { start_angle = a; }
  void synchronizeAnglesFrom(VoiceElement const &) {} // SynchronizePhase::No
  float low_freq = 0, high_freq = 0; // LOW_FREQ / HIGH_FREQ, handed to initialize_sweep at the end of setupAudioElement
  bool setup_ok = true;
  // engine.setAngleIncrements (:403-405), then — last step of setupAudioElement, gen.voice.h:999-1002 — engine.initialize_sweep
  void setAngleIncrements(double ai) {
    base_freq = static_cast<float>(ai * 0.5 * sample_rate);
    setup_ok = initialize_sweep(low_freq, high_freq);
  }
  double angleIncrements() const { // :407-421
    for (auto const &r : ramps)
      if (r.env.state == EnvelopeState::KeyReleased || r.env.state == EnvelopeState::KeyPressed) return r.angleIncrements();
    return 0;
  }
  bool initialize_sweep(float low, float high) { // :905-918 with do_initialize :1001-1044 and create_sweep's node 0 :700-713
    freq1 = low; freq2 = high;
    float n_frames = static_cast<float>(ms_to_frames(length, sample_rate));
    if (n_frames <= 0 || base_freq <= 0.f) return false;
    spec_it = -1; spec_end = 0;            // ramp_specs.reset()
    float len = length;
    len *= std::pow(2.f, min_exp);         // uniform_real_distribution{min_exp, max_exp} with min_exp == max_exp
    n_frames = static_cast<float>(ms_to_frames(len, sample_rate));
    ++spec_it; ++spec_end;                 // get_next_ramp_for_build
    spec.ctrl.setup(freq_to_angle_increment(freq1, sample_rate), freq_to_angle_increment(freq2, sample_rate), n_frames, phase_ratio1 * n_frames);
    spec_it = -1;                          // finalize
    return true;
  }
  void playNextSpec() { // :775-801
    VoiceRamp *new_ramp = nullptr;
    for (auto &r : ramps) if (r.env.state == EnvelopeState::EnvelopeDone2 && goOn()) new_ramp = &r; // get_ramps :275-290: the last one wins
    if (!new_ramp) return;
    ++spec_it;                             // get_next_ramp_for_run
    if (spec_it == spec_end) return;
    new_ramp->ctrl = spec.ctrl;
    new_ramp->forgetPastSignals();
    new_ramp->sample_rate = sample_rate;
    new_ramp->sine.setStartAngle(start_angle);
    new_ramp->volume_target = spec.vol;
    new_ramp->env.setAHDSR(AHDSR{xfade_len, 0, 0, xfade_len, 1.f}, sample_rate);
    new_ramp->env.tryAcquire();
    int const time_to_release = static_cast<int>(.5f + new_ramp->ctrl.get_duration_in_samples()) - xfade_len;
    new_ramp->env.onKeyPressed(0);
    new_ramp->env.onKeyReleased(std::max(time_to_release, 0));
  }
  void orchestrate_algos() { // :741-773
    if (spec_it == spec_end) return;
    Spec *cur = (spec_it >= 0 && spec_it < spec_end) ? &spec : nullptr;
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
  void step() { // BaseVolumeAdjusted::step over SoundEngine::step :736-739
    orchestrate_algos();
    for (auto &r : ramps) if (r.env.state != EnvelopeState::EnvelopeDone2) r.step(); // step_algos :423-436
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
  double imag(int i) const { // BaseVolumeAdjusted::imag(i) over SoundEngine::imag(i) :458-477
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

// SynthImpl of gen.voice (ImplBase<SWEEP>::setupAudioElement, gen.voice.h:883-1035) as the synchronous initializer + setup of one note
struct VoiceInitializer {
  ora_voice_params p{};
  bool operator==(VoiceInitializer const &) const { return true; }
  void operator()(VoiceElement &e, bool, int64_t noteid) const {
    // c.elem.set_sample_rate precedes the volume target in onEvent (gen.crtp.h:423-424)
    e.sample_rate = p.sample_rate;
    for (auto &r : e.ramps) r.sample_rate = p.sample_rate;
    e.xfade_len = static_cast<int>(0.5f + (p.xfade_length_seconds * p.sample_rate)); // setEnvelopeCharacTime :893
    e.min_exp = e.max_exp = p.length_exponent;                                        // set_length_exp(ex, ex) :927-929
    e.length = p.length_ms;                                                           // :933 (set after setAngleIncrements in the reference; no dependence)
    e.low_freq = p.low_freq; e.high_freq = p.high_freq;
    for (auto &r : e.ramps) { r.low_index = p.loudness_ref_freq_index; r.log_ratio = p.loudness_compensation; r.loudness_level = p.loudness_level; r.gain0 = p.sine_gain; }
    float pan = p.random_pan ? hash_uniform_pm1(p.seed, static_cast<uint64_t>(noteid), 3) : p.pan;
    for (auto &r : e.ramps) r.gain_ = stereo(pan);
  }
};

struct VoiceSynth : SynthT<VoiceElement, VoiceInitializer, 2, VoiceElemTraits> {
  using SynthT::SynthT;
};

} // namespace ora
