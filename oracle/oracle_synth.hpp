/*
 * TEST INFRASTRUCTURE — CPU oracle, part 3: audio elements, polyphonic synth, stepper.
 * Restates include/audioelement.h (hot subset), include/gen.crtp.h, include/gen.sine.h,
 * include/simple_audio_context.h, include/stereo.h of /root/reference, per sample, in double,
 * single thread (the offline mode of the reference is single-threaded: lib.cpp:1185-1235).
 */
#pragma once
#include "oracle_algo.hpp"
#include <deque>
#include <functional>
#include <limits>
#include <memory>

namespace ora {

constexpr int n_frames_per_buffer = 16; // audioelement.h:3

// ---- include/stereo.h:18-27 -------------------------------------------------------------------
struct StereoGain { double gains[2] = {0, 0}; };
inline StereoGain stereo(double pan) {
  pan = std::min(pan, 1.);
  pan = std::max(pan, -1.);
  double angle = 0.25 * M_PI * (pan + 1.);
  return {{std::cos(angle), std::sin(angle)}};
}

// ---- audioelement.h:157-163 -------------------------------------------------------------------
enum class EnvelopeState : unsigned char { SoonKeyPressed, KeyPressed, KeyReleased, EnvelopeDone1, EnvelopeDone2 };

struct AHDSR { // audioelement.h:314-335 (all interpolations LINEAR on this path, lib.cpp:75-81)
  int32_t attack, hold, decay, release;
  float sustain;
  bool operator==(AHDSR const &o) const { return attack == o.attack && hold == o.hold && decay == o.decay && release == o.release && sustain == o.sustain; }
};

// sound.functions.h:29-35
inline int ms_to_frames(float duration_ms, int sample_rate) {
  float fval = (static_cast<float>(sample_rate) / 1000.f) * duration_ms;
  return static_cast<int>(0.5f + fval);
}
// lib.cpp:63-82
inline AHDSR mkAHDSR(int sr, float a, float h, float d, float r, float s) {
  auto f = [sr](float secs) { return ms_to_frames(1000.f * secs, sr); };
  return AHDSR{f(a), f(h), f(d), f(r), s};
}

// AHDSREnvelope<Atomic, double, WaitForKeyRelease, AllowZeroAttack::No>
// = EnvelopeCRT (audioelement.h:665-842) over AHDSREnvelopeBase (:912-1099) over WithMinChangeDuration (:855-888)
struct Envelope {
  // WithMinChangeDuration
  int32_t minChangeDuration = 0, nextMinChangeDuration = 0;
  void setMinChangeDurationSamples(int n) { nextMinChangeDuration = n; }
  void updateMinChangeDuration() { minChangeDuration = nextMinChangeDuration; }
  // AHDSREnvelopeBase
  int32_t ahdCounter = 0, A = 100000, H = 0, D = 100000, R = 100000;
  float SMinusOne = -0.5f;
  int ahdState = -1; // -1 none, 0 Attacking, 1 Holding, 2 Decaying
  // EnvelopeCRT
  double _value = 0, _topValue = 0;
  EnvelopeState state = EnvelopeState::EnvelopeDone2;
  int32_t counter = 0, currentReleaseDelay = -1;

  void setAHDSR(AHDSR const &s, int sample_rate) { // :923-952
    int32_t const min_dt = sample_rate / 1000;     // :863-872
    bool const hasDecay = s.sustain < 0.999999;
    SMinusOne = hasDecay ? std::min(std::max(s.sustain, 0.f), 1.f) - 1.f : 0.f;
    A = std::max(s.attack, min_dt);
    H = std::max(s.hold, 0);
    D = hasDecay ? std::max(s.decay, min_dt) : 0;
    R = std::max(s.release, min_dt);
  }
  int32_t getMaxCounterForAHD() const { // :1066-1084
    switch (ahdState) {
      case 0: return std::max(minChangeDuration, A);
      case 1: return H;
      default: return (D == 0) ? 0 : std::max(minChangeDuration, D);
    }
  }
  void onAHDStateChange() { // :1040-1053
    ahdCounter = 0;
    updateMinChangeDuration();
    if (ahdState < 0) return;
    if (0 == getMaxCounterForAHD()) stepAHD();
  }
  void stepAHD() { // :1055-1064
    if (ahdCounter < getMaxCounterForAHD()) return;
    ahdState = (ahdState == 2) ? -1 : ahdState + 1; // rotateAHD :898-910
    onAHDStateChange();
  }
  void startPressed() { ahdState = 0; onAHDStateChange(); } // :955-958
  double stepPressed() { // :960-1000 (WaitForKeyRelease => always a value)
    if (ahdState >= 0) { stepAHD(); ++ahdCounter; }
    if (ahdState < 0) return 1.0 + static_cast<double>(SMinusOne);
    double from, toMinusFrom;
    switch (ahdState) {
      case 0: from = 0; toMinusFrom = 1; break;
      case 1: from = 1; toMinusFrom = 0; break;
      default: from = 1; toMinusFrom = static_cast<double>(SMinusOne); break;
    }
    return itp_linear(static_cast<double>(ahdCounter), from, toMinusFrom, static_cast<double>(getMaxCounterForAHD()));
  }
  int32_t getReleaseTime() const { return std::max(minChangeDuration, R); } // :1002-1004

  bool tryAcquire() { // :363-369
    if (state != EnvelopeState::EnvelopeDone2) return false;
    state = EnvelopeState::SoonKeyPressed;
    return true;
  }
  bool acquireStates() const { return state == EnvelopeState::SoonKeyPressed; } // :374-376
  bool isEnvelopeRTActive() const { return state != EnvelopeState::SoonKeyPressed && state != EnvelopeState::EnvelopeDone2; } // :390-395
  bool isEnvelopeFinished() const { return state == EnvelopeState::EnvelopeDone2; }

  void forgetPastSignals() { counter = 0; updateMinChangeDuration(); } // :694-700
  void startReleased() { // :836-841
    updateMinChangeDuration();
    state = EnvelopeState::KeyReleased;
    _topValue = _value;
    counter = 0;
  }
  void step() { // :702-756
    if (currentReleaseDelay >= 0) {
      if (currentReleaseDelay == 0) startReleased();
      --currentReleaseDelay;
    }
    ++counter;
    switch (state) {
      case EnvelopeState::KeyPressed:
        if (counter < 1) return;
        if (counter == 1) startPressed();
        _value = stepPressed();
        break;
      case EnvelopeState::KeyReleased:
        _value = itp_linear(static_cast<double>(counter), _topValue, -_topValue, static_cast<double>(getReleaseTime()));
        if (counter >= getReleaseTime()) {
          state = EnvelopeState::EnvelopeDone1;
          counter = 0;
          _value = 0;
        }
        break;
      case EnvelopeState::EnvelopeDone1:
        if (counter > n_frames_per_buffer) {
          state = EnvelopeState::EnvelopeDone2;
          counter = 0;
        }
        _value = 0;
        break;
      default: _value = 0; break;
    }
  }
  double value() const { return _value; }
  void onKeyPressed(int32_t delay) { // :762-767
    state = EnvelopeState::KeyPressed;
    counter = -delay;
    currentReleaseDelay = -1;
    _value = 0;
  }
  void onKeyReleased(int32_t delay) { // :769-781
    if (state != EnvelopeState::KeyPressed) return;
    if (counter + delay <= 0) state = EnvelopeState::EnvelopeDone1;
    else currentReleaseDelay = delay;
  }
  bool canHandleExplicitKeyReleaseNow(int32_t delay) const { // :802-811
    return state == EnvelopeState::KeyPressed && currentReleaseDelay == -1 && (-counter <= delay);
  }
};

// audioelement.h:466-483
inline double freqAliasingMultiplicator(double inc) {
  if (inc == 0.) return 1.;
  double halfN = 1. / inc;
  if (halfN < 1.) return 0.;
  if (halfN > 4.) return 1.;
  return (halfN - 1.) * (1. / 3.);
}

// SineOscillatorAlgo<double, FAST> audioelement.h:2365-2452
struct SineOsc {
  std::complex<double> cur{1, 0}, mult{1, 0};
  double aliasingMult = 1;
  std::optional<double> angle_increments;
  SineOsc() { setAngleIncrements(0); }
  void setStartAngle(double f) { cur = std::polar(1.0, M_PI * f); }
  void synchronizeAngles(SineOsc const &o) { cur = o.cur; }
  void setAngleIncrements(double f) {
    if (angle_increments && *angle_increments == f) return;
    angle_increments = f;
    mult = std::polar(1.0, M_PI * f);
    aliasingMult = freqAliasingMultiplicator(f);
  }
  void step() {
    // complex product written out (std::complex operator* may take the NaN-recovery path)
    double re = cur.real() * mult.real() - cur.imag() * mult.imag();
    double im = cur.real() * mult.imag() + cur.imag() * mult.real();
    double n = re * re + im * im;
    double g = 0.5 * (3.0 - n); // approx_normalize :2443-2447
    cur = {re * g, im * g};
  }
  double imag() const { return aliasingMult * cur.imag(); }
  double angleIncrements() const { return *angle_increments; }
};

// InterpolatedFreq<double> audioelement.h:2705-2817
struct InterpolatedFreq {
  double from = 0, to = 0, cur_sample = 0, duration_in_samples = 0, C = 0;
  std::optional<double> f_result;
  void forgetPastSignals() { f_result.reset(); }
  double prop() const { return (to == from) ? 1.f : -std::log(from / to) / (to - from); } // :2804-2816
  void setup(double dur) { // :2739-2750
    duration_in_samples = dur;
    cur_sample = 0;
    if (f_result) { from = *f_result; C = prop(); }
  }
  void setAngleIncrements(double inc) { // :2752-2762
    cur_sample = 0;
    to = inc;
    from = f_result ? *f_result : inc;
    C = prop();
  }
  double step() { // :2764-2788
    if (f_result && *f_result == to) return to;
    if (cur_sample + .5f > duration_in_samples) cur_sample = duration_in_samples;
    // NormalizedInterpolation::get_unfiltered_value(t, d, from, to), LINEAR
    f_result = from + (to - from) * (cur_sample / duration_in_samples);
    if (cur_sample < duration_in_samples) {
      double f = from + (to - from) * (cur_sample + .5f) / duration_in_samples;
      cur_sample += C * f;
    }
    return *f_result;
  }
  double getAngleIncrementFrom() const { return from; }
};

// StereoPanned<VolumeAdjusted<Enveloped<FreqCtrl_<SineOscillatorAlgo, InterpolatedFreq>, AHDSREnvelope>>>
// (lib.cpp:6-19).  with_ctrl=false gives driver (A)'s element
// Enveloped<VolumeAdjusted<SineOscillatorAlgo<double>>, AHDSREnvelope> (resynth.hpp:13-15): same
// arithmetic, the increment reaching the oscillator directly.
struct ResynthElement {
  bool with_ctrl = true;
  // Enveloped
  Envelope env;
  // FreqCtrl_
  InterpolatedFreq ctrl;
  SineOsc osc;
  // BaseVolumeAdjusted (audioelement.h:1110-1236)
  std::optional<double> volume, volume_target;
  double filter_init_with_inc = 0;
  double max_filter_increment = std::numeric_limits<double>::max();
  OnePoleLowPass volume_filter;
  // StereoPanned
  StereoGain gain_{{1, 1}};

  // FreqCtrl_::angleIncrements audioelement.h:3117-3124
  double angleIncrements() const {
    double ai = osc.angleIncrements();
    if (!with_ctrl) return ai;
    if (ai) return ai;
    return ctrl.getAngleIncrementFrom();
  }
  void forgetPastSignals() { // StereoPanned:1358 -> BaseVolumeAdjusted:1168-1172 -> Enveloped:192-195 -> FreqCtrl_:3096-3101
    env.forgetPastSignals();
    ctrl.forgetPastSignals();
    volume.reset();
    volume_target.reset();
  }
  void setVolumeTarget(double v) { volume_target = v; }
  void set_sample_rate(int) {} // StereoPanned:1350 -> ... -> SineOscillatorAlgo: nothing on this element depends on it
  // what setPhase reaches through getOsc() (gen.crtp.h:141,148-152)
  void synchronizeAnglesFrom(ResynthElement const &o) { osc.synchronizeAngles(o.osc); }
  void setStartAngle(double a) { osc.setStartAngle(a); }
  void setMaxFilterIncrement(double ai) { max_filter_increment = ai; }
  void setStereoGain(StereoGain const &g) { gain_ = g; }
  void setAngleIncrements(double a) { // Enveloped:218-225 -> FreqCtrl_:3103-3108
    double period = (a == 0.) ? 1. : 2. / std::abs(a); // sound.functions.h:69-76
    env.setMinChangeDurationSamples(static_cast<int>(0.5f + 2.5 * period));
    if (with_ctrl) ctrl.setAngleIncrements(a); else osc.setAngleIncrements(a);
  }
  void step() { // StereoPanned:1378 -> BaseVolumeAdjusted:1195-1216 -> Enveloped:197-200 -> FreqCtrl_:3126-3129
    env.step();
    if (with_ctrl) osc.setAngleIncrements(ctrl.step());
    osc.step();
    if (!volume) {
      filter_init_with_inc = std::min(max_filter_increment, std::abs(angleIncrements()));
      volume_filter.initWithAngleIncrement(filter_init_with_inc);
      volume_filter.setInitialValue(*volume_target);
    } else if (*volume != *volume_target) {
      double inc = std::min(max_filter_increment, std::abs(angleIncrements()));
      if (filter_init_with_inc != inc) {
        filter_init_with_inc = inc;
        volume_filter.initWithAngleIncrement(filter_init_with_inc);
      }
    }
    volume_filter.feed(*volume_target);
    volume = volume_filter.filtered();
  }
  // driver (A): Enveloped<VolumeAdjusted<Sine>>::imag() = (volume * osc.imag()) * env   (:231 over :1128)
  double imag_mono() const { return (*volume * osc.imag()) * env.value(); }
  // rt.resynth: StereoPanned(:1383) over VolumeAdjusted(:1135) over Enveloped(:231): gain * (volume * (osc.imag() * env))
  double imag(int i) const { return gain_.gains[i] * (*volume * (osc.imag() * env.value())); }
};

// ---- events.h ---------------------------------------------------------------------------------
enum class EventType : uint8_t { NoteOn, NoteOff, NoteChange };
struct Event {
  EventType type;
  int64_t noteid;
  float frequency = 0; // noteOn.frequency / noteChange.changed_frequency
  float velocity = 0;  // noteOn.velocity / noteChange.changed_velocity
  int32_t delay = 0;   // frames between the one-shot and the key press / release (the "on time" branch of gen.crtp.h:506-512,556-560);
                       // 0 = the "late" branch, which is what the offline job always takes
};
enum class onEventResult { OK = 0, DROPPED_NOTE };

// sound.functions.h:58-62 (instantiated with T = float by gen.sine.h:19 and gen.crtp.h:414,544)
inline float freq_to_angle_increment(float freq, int sample_rate) { return 2 * freq / sample_rate; }

constexpr float reduceUnadjustedVolumes = 0.1f;                              // audioelement.h:1270
constexpr double sineBaseVolume = reduceUnadjustedVolumes * 1.;             // audioelement.h:2370, sound.h:84

// One record of what reached the synth, for event-level parity tests.
struct EventLogEntry {
  int64_t sample;  // output sample index at which the one-shot ran
  int32_t type;    // 0 on, 1 off, 2 change, 3 dropped note-on
  int32_t channel; // -1 when dropped
  int64_t noteid;
  float frequency, velocity;
};

// SimpleAudioOutContext<2, MasterLockFree> — simple_audio_context.h:5-133 (offline: one thread, FIFO one-shots)
struct Stepper {
  static constexpr int nOuts = 2;
  using OneShot = std::function<void()>;
  using Compute = std::function<bool(double *, int)>;
  std::deque<OneShot> oneShots;
  std::vector<Compute> computes;
  std::function<void(double *, int)> post;
  void enqueueOneShot(OneShot f) { oneShots.push_back(std::move(f)); }
  bool registerSimpleCompute(Compute f) { computes.push_back(std::move(f)); return true; }
  void step(float *out, int nFrames) { // :69-112
    while (!oneShots.empty()) { auto f = std::move(oneShots.front()); oneShots.pop_front(); f(); }
    double buf[n_frames_per_buffer * nOuts];
    while (nFrames > 0) {
      int nLocal = std::min(nFrames, n_frames_per_buffer);
      int nSamples = nLocal * nOuts;
      std::fill(buf, buf + nSamples, 0.0);
      for (size_t i = 0; i < computes.size();) {
        if (computes[i](buf, nLocal)) ++i; else computes.erase(computes.begin() + i);
      }
      if (post) post(buf, nLocal);
      for (int i = 0; i != nSamples; ++i, ++out) *out = static_cast<float>(buf[i]);
      nFrames -= n_frames_per_buffer;
    }
  }
};

// ResynthElementInitializer<double> lib.cpp:84-134
struct ElementInitializer {
  int sample_rate = 0, stride = 0;
  float stereo_spread = 0;
  AHDSR ahdsr{};
  uint64_t seed = 0;
  bool operator==(ElementInitializer const &o) const { return sample_rate == o.sample_rate && stride == o.stride && stereo_spread == o.stereo_spread && ahdsr == o.ahdsr; }
  void operator()(ResynthElement &e, bool new_note, int64_t noteid) const {
    e.env.setAHDSR(ahdsr, sample_rate);
    e.setMaxFilterIncrement(2. / static_cast<double>(stride));
    e.ctrl.setup(stride);
    if (new_note) e.setStereoGain(stereo(stereo_spread * hash_uniform_pm1(seed, static_cast<uint64_t>(noteid), 1)));
  }
};

// sine::Synth<nOuts, Element, Jitter::No, SynchronizePhase::Yes, DefaultStartPhase::Random, HandleNoteOff::Yes, nVoices, Initializer>
// = ImplCRTP (gen.crtp.h:181-629) with SynthImpl (gen.sine.h:6-22).  Instantiated twice by rt.resynth (lib.cpp:201-221): the
// resynthesis synth (2 outs, ResynthElement) and the mono vocoder carrier synth (VocoderCarrierElement, oracle_carrier.hpp).
// ElemTraits: baseVolume (Element::baseVolume as the double it converts to) and the salt of the declared start-phase hash.
struct ResynthElemTraits {
  static constexpr double baseVolume = sineBaseVolume; // StereoPanned<VolumeAdjusted<...>>: the algo's baseVolume (audioelement.h:1257,2370)
  static constexpr uint64_t phase_salt = 0;
};
template <typename Elem, typename Init, int NOUT, typename Traits>
struct SynthT {
  struct Channel {
    Elem elem;
    std::optional<int64_t> noteid;
    float ref_angle_increments = 0;
  };
  std::vector<Channel> channels; // n_channels = 2 * nVoices (gen.crtp.h:225)
  float last_angle_increment_multiplier = 1.f;
  std::optional<Init> initializer;
  int phase_policy = 0;
  uint64_t seed = 0;
  std::vector<EventLogEntry> *log = nullptr;
  int64_t const *now = nullptr; // output sample index the next stepper.step() will produce

  explicit SynthT(int n_channels) : channels(n_channels) {}

  void initialize(Stepper &chans) { // :248-262
    chans.enqueueOneShot([this, &chans]() { chans.registerSimpleCompute([this](double *b, int n) { return compute(b, n); }); });
  }

  bool compute(double *buffer, int n_frames) { // :350-378
    for (auto &c : channels) {
      if (!c.elem.env.isEnvelopeRTActive()) continue;
      for (int i = 0; i != n_frames; ++i) {
        c.elem.step();
        if (!c.elem.env.isEnvelopeRTActive()) break;
        for (int j = 0; j < NOUT; ++j) buffer[NOUT * i + j] += c.elem.imag(j);
      }
    }
    return true;
  }

  void setPhase(Channel &e, int64_t noteid) { // :128-156
    double const ai = e.elem.angleIncrements();
    for (auto &other : channels) {
      if (&other == &e) continue;
      if (!other.elem.env.isEnvelopeRTActive()) continue;
      if (other.elem.angleIncrements() == ai) { e.elem.synchronizeAnglesFrom(other.elem); return; }
    }
    float a = phase_policy == 0 ? hash_uniform_pm1(seed, static_cast<uint64_t>(noteid), Traits::phase_salt) : 0.f;
    e.elem.setStartAngle(a);
  }

  void logEvent(int type, int channel, Event const &e) {
    if (log) log->push_back({now ? *now : -1, type, channel, e.noteid, e.frequency, e.velocity});
  }

  onEventResult onEvent(int sample_rate, Event const &e, Stepper &chans) { // :382-622
    switch (e.type) {
      case EventType::NoteOn: {
        Channel *channel = nullptr;
        for (auto &c : channels) {
          if (!c.elem.env.tryAcquire()) continue;
          channel = &c;
          break;
        }
        if (!channel) { logEvent(3, -1, e); return onEventResult::DROPPED_NOTE; }
        auto &c = *channel;
        c.noteid = e.noteid;
        c.elem.forgetPastSignals();
        c.elem.set_sample_rate(sample_rate);                     // :424
        c.elem.setVolumeTarget(Traits::baseVolume * e.velocity); // Element::baseVolume * velocity(float), :425
        if (initializer) (*initializer)(c.elem, true, e.noteid);
        // SynthImpl::setupAudioElement gen.sine.h:9-21
        c.elem.setAngleIncrements(freq_to_angle_increment(e.frequency * last_angle_increment_multiplier, sample_rate));
        logEvent(0, static_cast<int>(channel - channels.data()), e);
        float ref = freq_to_angle_increment(e.frequency, sample_rate);
        chans.enqueueOneShot([this, &c, ref, noteid = e.noteid, delay = e.delay]() {
          if (!c.elem.env.acquireStates()) return;
          c.ref_angle_increments = ref;
          setPhase(c, noteid);
          c.elem.env.onKeyPressed(delay);
        });
      } break;
      case EventType::NoteOff: {
        chans.enqueueOneShot([this, noteid = e.noteid, delay = e.delay]() {
          for (auto &c : channels) {
            if (!c.noteid || *c.noteid != noteid) continue;
            if (c.elem.env.canHandleExplicitKeyReleaseNow(delay)) c.elem.env.onKeyReleased(delay);
            return;
          }
        });
        int ch = -1;
        for (size_t i = 0; i < channels.size(); ++i) if (channels[i].noteid && *channels[i].noteid == e.noteid) { ch = static_cast<int>(i); break; }
        logEvent(1, ch, e);
      } break;
      case EventType::NoteChange: {
        float const angleIncrements = freq_to_angle_increment(e.frequency, sample_rate);
        float const angleIncrementsMult = angleIncrements * last_angle_increment_multiplier;
        chans.enqueueOneShot([this, e, angleIncrements, angleIncrementsMult]() {
          for (auto &c : channels) {
            if (!c.noteid || *c.noteid != e.noteid) continue;
            c.elem.setVolumeTarget(Traits::baseVolume * e.velocity);
            c.elem.setAngleIncrements(angleIncrementsMult);
            c.ref_angle_increments = angleIncrements;
          }
        });
        int ch = -1;
        for (size_t i = 0; i < channels.size(); ++i) if (channels[i].noteid && *channels[i].noteid == e.noteid) ch = static_cast<int>(i);
        logEvent(2, ch, e);
      } break;
    }
    return onEventResult::OK;
  }
};
using Synth = SynthT<ResynthElement, ElementInitializer, 2, ResynthElemTraits>;

} // namespace ora
