/*
 * TEST INFRASTRUCTURE — CPU oracle, part 5: the vocoder carrier synth and the time-domain band-pass vocoder.
 *
 *  (a) VocoderCarrierElement<double> (source/rt.resynth.lib.cpp:21-52) with its initializer (:137-196), driven by the mono
 *      SynthVocoderCarier (:212-221) = the same ImplCRTP as the resynthesis synth (oracle_synth.hpp: SynthT).
 *      Restated from include/audioelement.h: Phased (:1395-1477), FOscillatorAlgo_ (:1651-1676), PulseTrainAlgo_ (:1691-1748),
 *      soundBufferWrapperAlgo (:1507-1583), UnityGainMix (:1882-1982), BaseVolumeAdjusted / VolumeAdjusted (:1110-1265),
 *      Enveloped (:181-291), FreqCtrl_ (:3055-3155), InterpolatedFreq (:2705-2817), and include/sound.functions.h:86-138.
 *      The noise voice plays the process-wide white-noise sound buffer (getWhiteNoise, source/sound.cpp:3-21 -> soundBuffer::soundBuffer
 *      :134-147 -> generate :84-100 over white_gaussian_noise, include/noise.h:3-5; the buffer's length from include/sound.h:53-60 and
 *      include/sound.functions.h:45-50), all of it in the reference; the only thing outside it is the unseeded mersenne<SEEDED::No>()
 *      engine behind std::normal_distribution<float> — implementation-defined twice over — for which white_noise_sample() below is this
 *      build's DECLARED policy (parity unpinned like the other draws).  Its constant angleIncrements() (0.5f,
 *      include/sound.h:256-258) takes part in UnityGainMix::angleIncrements().
 *  (b) the band-pass bank the author replaced by FFTs (source/rt.resynth.lib.vocoder.cpp:46-79 BandPass / EnvelopeFollower,
 *      :310-394 Modulator, :667-732 Carrier; orders 1, :737-738).  Its L0 filters are PARITY UNPINNED like every Filter<> on this
 *      path; decisions below.
 */
#pragma once
#include "oracle_synth.hpp"
#include <mutex>
#include <unordered_map>

namespace ora {

// ---- include/sound.functions.h:86-138 ----------------------------------------------------------
inline double wave_square(double ang) { return (0.5 < ang && ang < 1.5) ? -1.0 : 1.0; }
inline double wave_pulse(double ang, double pulse_width, double high, double low) { return ang < pulse_width ? high : low; }
inline double wave_triangle(double ang) {
  if (ang < 0.5) return 2.0 * ang;
  if (ang < 1.5) return 2.0 + -2.0 * ang;
  return -4.0 + 2.0 * ang;
}
inline double wave_saw(double ang) { return ang <= 1.0 ? ang : -2.0 + ang; }

inline double phaseToNormalForm(double phase) { // audioelement.h:418-428
  double nf = std::fmod(phase, 2.0);
  if (nf < 0) nf += 2.0;
  return nf;
}

// Phased<Algo> audioelement.h:1395-1477.  Algo: imag(angle), onNewPeriod(), onForgetPastSignals().
template <typename Algo>
struct Phased {
  double angle_ = 0, angle_increments = 0, aliasingMult = 1;
  Algo algo;
  void forgetPastSignals() { algo.onForgetPastSignals(); }
  void synchronizeAngles(Phased const &o) { setStartAngle(o.angle_); }
  void setStartAngle(double a) { angle_ = phaseToNormalForm(a); }
  void setAngleIncrements(double v) {
    angle_increments = phaseToNormalForm(v);
    aliasingMult = freqAliasingMultiplicator(v);
  }
  double angleIncrements() const { return angle_increments; }
  void step() {
    angle_ += angle_increments;
    if (angle_ > 2.0) { angle_ -= 2.0; algo.onNewPeriod(); }
  }
  double imag() const { return aliasingMult * algo.imag(angle_); }
};
struct SawAlgo { void onNewPeriod() {} void onForgetPastSignals() {} double imag(double a) const { return wave_saw(a); } };
struct TriangleAlgo { void onNewPeriod() {} void onForgetPastSignals() {} double imag(double a) const { return wave_triangle(a); } };
struct SquareAlgo { void onNewPeriod() {} void onForgetPastSignals() {} double imag(double a) const { return wave_square(a); } };
struct PulseTrainAlgo_ { // audioelement.h:1691-1748
  std::optional<double> used_pulse_width;
  double used_high = 0, used_low = 0, pulse_width = 0, high = 0, low = 0;
  void setPulseWidth(double w) {
    w = std::max(0., w);
    w = std::min(2., w);
    pulse_width = w;
    high = 0.5 * (2. - w);
    low = high - 1.;
    if (!used_pulse_width) onNewPeriod();
  }
  double imag(double angle) const { return wave_pulse(angle, *used_pulse_width, used_high, used_low); }
  void onForgetPastSignals() { used_pulse_width.reset(); }
  void onNewPeriod() { used_pulse_width = pulse_width; used_high = high; used_low = low; }
};
// ---- the white-noise sound buffer: getWhiteNoise(sample_rate), source/sound.cpp:3-21 ---------------------------------------------
// soundId{sample_rate, Sound::NOISE, noise_duration = .05f}: the constant is consumed as a FREQUENCY (include/sound.h:53-60), so the buffer
// holds freq_to_period_in_samples(.05f, sample_rate) = int(float(sample_rate) / .05f) frames (include/sound.functions.h:45-50): 960 000 at
// 48 kHz.  soundBuffer::soundBuffer (sound.cpp:134-147) -> generate(period, white_gaussian_noise) (:84-100): values[i] = the float draw,
// widened to the buffer's double.  (The period is >= 20: the centring / maximising of :196-219 does not apply.)
inline int white_noise_period(int sample_rate) {
  float const freq_hz = .05f, sr = static_cast<float>(sample_rate);
  return static_cast<int>(sr / freq_hz);
}
// DECLARED POLICY for `std::normal_distribution<float>{0.f, 1.f}(mersenne<SEEDED::No>())` (include/noise.h:3-5): the reference draws the
// buffer once per process from an engine seeded by the OS, so no two runs of it agree; here sample i is, once and for all, the Box-Muller
// transform of two 24-bit uniforms taken from the counter hash of (a fixed key, i): u1 in (0, 1], u2 in [0, 1),
// z = sqrt(-2 ln u1) cos(2 pi u2) evaluated in double and rounded to the float the reference's draw is.
inline float white_noise_sample(uint64_t i) {
  uint64_t const h = splitmix64(0x77686974656e6f69ull ^ splitmix64(i * 2 + 7)); // key: "whitenoi"; salt 7
  double const u1 = (static_cast<double>(h >> 40) + 1.0) * (1.0 / 16777216.0);
  double const u2 = static_cast<double>((h >> 16) & 0xffffffull) * (1.0 / 16777216.0);
  return static_cast<float>(std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2));
}
inline std::vector<double> const &getWhiteNoise(int sample_rate) { // one buffer per sample rate and process, like the reference's static
  static std::mutex mu;
  static std::unordered_map<int, std::vector<double>> bufs;
  std::lock_guard<std::mutex> g(mu);
  auto it = bufs.find(sample_rate);
  if (it == bufs.end()) {
    std::vector<double> v(static_cast<size_t>(std::max(1, white_noise_period(sample_rate))));
    for (size_t i = 0; i < v.size(); ++i) v[i] = white_noise_sample(i);
    it = bufs.emplace(sample_rate, std::move(v)).first;
  }
  return it->second;
}
// soundBufferWrapperAlgo<Sound::NOISE> audioelement.h:1507-1583
struct NoiseVoice {
  int index = -1;
  std::vector<double> const *sb = nullptr;
  void set_sample_rate(int s) { sb = &getWhiteNoise(s); } // :1524-1527
  void forgetPastSignals() {}
  void synchronizeAngles(NoiseVoice const &o) { index = o.index; } // :1532-1537
  void setStartAngle(double a) { // :1544-1557: a in [-1, 1]
    index = static_cast<int>(((a + 1.) * sb->size() * 0.5) + 0.5);
    if (index < 0) index = 0;
    --index; // "because index will be incremented in step before being used"
  }
  void setAngleIncrements(double) {}
  double angleIncrements() const { return 0.5f; } // FGetBuffer<NOISE>::getAngleIncrements, include/sound.h:256-258
  void step() { // :1568-1574
    ++index;
    if (index >= static_cast<int>(sb->size())) index = 0;
  }
  double imag() const { return (*sb)[index]; } // :1563-1566
};
// SineOscillatorAlgo needs the same three entry points under VolumeAdjusted
struct SineVoice : SineOsc {
  void forgetPastSignals() {}
};

// VolumeAdjusted<ALGO> = BaseVolumeAdjusted (audioelement.h:1110-1265) without a maximum filter increment
template <typename Osc>
struct VolAdj {
  Osc osc;
  std::optional<double> volume, volume_target;
  double filter_init_with_inc = 0;
  OnePoleLowPass volume_filter;
  void forgetPastSignals() { osc.forgetPastSignals(); volume.reset(); volume_target.reset(); }
  void setVolumeTarget(double v) { volume_target = v; }
  void setAngleIncrements(double ai) { osc.setAngleIncrements(ai); }
  double angleIncrements() const { return osc.angleIncrements(); }
  void setStartAngle(double a) { osc.setStartAngle(a); }
  void synchronizeAngles(VolAdj const &o) { osc.synchronizeAngles(o.osc); }
  void step() { // :1195-1216
    osc.step();
    if (!volume) {
      filter_init_with_inc = std::min(std::numeric_limits<double>::max(), std::abs(osc.angleIncrements()));
      volume_filter.initWithAngleIncrement(filter_init_with_inc);
      volume_filter.setInitialValue(*volume_target);
    } else if (*volume != *volume_target) {
      double const inc = std::min(std::numeric_limits<double>::max(), std::abs(osc.angleIncrements()));
      if (filter_init_with_inc != inc) { filter_init_with_inc = inc; volume_filter.initWithAngleIncrement(filter_init_with_inc); }
    }
    volume_filter.feed(*volume_target);
    volume = volume_filter.filtered();
  }
  double imag() const { return *volume * osc.imag(); }
};

// UnityGainMix<VolumeAdjusted<noise>, VolumeAdjusted<saw>, <triangle>, <square>, <sine>, <pulse>> audioelement.h:1882-1982
struct CarrierMix {
  VolAdj<NoiseVoice> noise;
  VolAdj<Phased<SawAlgo>> saw;
  VolAdj<Phased<TriangleAlgo>> triangle;
  VolAdj<Phased<SquareAlgo>> square;
  VolAdj<SineVoice> sine;
  VolAdj<Phased<PulseTrainAlgo_>> pulse;
  template <typename F> void each(F f) { f(noise); f(saw); f(triangle); f(square); f(sine); f(pulse); }
  template <typename F> void each(F f) const { f(noise); f(saw); f(triangle); f(square); f(sine); f(pulse); }
  void set_sample_rate(int s) { noise.osc.set_sample_rate(s); } // UnityGainMix:1915-1919; the other voices' are empty (:1416, :2016-2018 -> no-op)
  void forgetPastSignals() { each([](auto &ae) { ae.forgetPastSignals(); }); }
  void step() { each([](auto &ae) { ae.step(); }); }
  void setAngleIncrements(double v) { each([v](auto &ae) { ae.setAngleIncrements(v); }); }
  void setStartAngle(double v) { each([v](auto &ae) { ae.setStartAngle(v); }); }
  double angleIncrements() const { // the minimum, :1948-1958
    std::optional<double> minAngle;
    each([&minAngle](auto const &ae) {
      double const ai = std::abs(ae.angleIncrements());
      if (!minAngle || *minAngle > ai) minAngle = ai;
    });
    return minAngle ? *minAngle : 0;
  }
  void synchronizeAngles(CarrierMix const &o) {
    noise.synchronizeAngles(o.noise); saw.synchronizeAngles(o.saw); triangle.synchronizeAngles(o.triangle);
    square.synchronizeAngles(o.square); sine.synchronizeAngles(o.sine); pulse.synchronizeAngles(o.pulse);
  }
  double imag() const {
    double sum = 0.f;
    each([&sum](auto const &ae) { sum += ae.imag(); });
    return sum;
  }
};

// VocoderCarrierElement<double> = VolumeAdjusted<Enveloped<FreqCtrl_<UnityGainMix<...>, InterpolatedFreq>, AHDSREnvelope>, BaseVolumeDef::One>
struct CarrierElement {
  Envelope env;          // Enveloped
  InterpolatedFreq ctrl; // FreqCtrl_
  CarrierMix mix;
  // the outer BaseVolumeAdjusted
  std::optional<double> volume, volume_target;
  double filter_init_with_inc = 0;
  OnePoleLowPass volume_filter;

  double angleIncrements() const { // FreqCtrl_::angleIncrements audioelement.h:3117-3124
    double const ai = mix.angleIncrements();
    if (ai) return ai;
    return ctrl.getAngleIncrementFrom();
  }
  void forgetPastSignals() { // BaseVolumeAdjusted:1168-1172 -> Enveloped:192-195 -> FreqCtrl_:3096-3101
    env.forgetPastSignals();
    ctrl.forgetPastSignals();
    mix.forgetPastSignals();
    volume.reset();
    volume_target.reset();
  }
  void setVolumeTarget(double v) { volume_target = v; }
  void set_sample_rate(int s) { mix.set_sample_rate(s); } // BaseVolumeAdjusted:1155 -> Enveloped:276-279 -> FreqCtrl_ -> UnityGainMix
  void synchronizeAnglesFrom(CarrierElement const &o) { mix.synchronizeAngles(o.mix); }
  void setStartAngle(double a) { mix.setStartAngle(a); }
  void setAngleIncrements(double a) { // VolumeAdjusted:1251 -> Enveloped:218-225 -> FreqCtrl_:3103-3108
    double period = (a == 0.) ? 1. : 2. / std::abs(a);
    env.setMinChangeDurationSamples(static_cast<int>(0.5f + 2.5 * period));
    ctrl.setAngleIncrements(a);
  }
  void step() { // BaseVolumeAdjusted:1195-1216 -> Enveloped:197-200 -> FreqCtrl_:3126-3129
    env.step();
    mix.setAngleIncrements(ctrl.step());
    mix.step();
    if (!volume) {
      filter_init_with_inc = std::min(std::numeric_limits<double>::max(), std::abs(angleIncrements()));
      volume_filter.initWithAngleIncrement(filter_init_with_inc);
      volume_filter.setInitialValue(*volume_target);
    } else if (*volume != *volume_target) {
      double inc = std::min(std::numeric_limits<double>::max(), std::abs(angleIncrements()));
      if (filter_init_with_inc != inc) { filter_init_with_inc = inc; volume_filter.initWithAngleIncrement(filter_init_with_inc); }
    }
    volume_filter.feed(*volume_target);
    volume = volume_filter.filtered();
  }
  // BaseVolumeAdjusted::imag (:1126-1129) over Enveloped::imag (:227-231) over UnityGainMix::imag (:1966-1972)
  double imag(int) const { return *volume * (mix.imag() * env.value()); }
};
struct CarrierElemTraits {
  static constexpr double baseVolume = 1.0; // BaseVolumeDef::One (lib.cpp:49, audioelement.h:1243)
  static constexpr uint64_t phase_salt = 2; // declared policy: the carrier synth draws from its own hash stream
};

// VocoderCarrierElementInitializer<double> lib.cpp:137-196
struct CarrierInitializer {
  int sample_rate = 0;
  AHDSR ahdsr{};
  float noise_volume = 0, saw_volume = 0, triangle_volume = 0, square_volume = 0, sine_volume = 0, pulse_volume = 0, pulse_width = 0;
  bool operator==(CarrierInitializer const &o) const {
    return sample_rate == o.sample_rate && ahdsr == o.ahdsr && noise_volume == o.noise_volume && saw_volume == o.saw_volume &&
           triangle_volume == o.triangle_volume && square_volume == o.square_volume && sine_volume == o.sine_volume &&
           pulse_volume == o.pulse_volume && pulse_width == o.pulse_width;
  }
  void operator()(CarrierElement &e, bool, int64_t) const {
    e.env.setAHDSR(ahdsr, sample_rate);
    e.ctrl.setup(100); // ctrl.setup(100, itp::LINEAR)
    e.mix.noise.setVolumeTarget(noise_volume);
    e.mix.saw.setVolumeTarget(saw_volume);
    e.mix.triangle.setVolumeTarget(triangle_volume);
    e.mix.square.setVolumeTarget(square_volume);
    e.mix.sine.setVolumeTarget(sine_volume);
    e.mix.pulse.setVolumeTarget(pulse_volume);
    e.mix.pulse.osc.algo.setPulseWidth(pulse_width);
  }
};
using CarrierSynth = SynthT<CarrierElement, CarrierInitializer, 1, CarrierElemTraits>;

// NoteIdsGenerator include/events.h:228-281: note-off of a key releases its EARLIEST note-on
struct NoteIdsGenerator {
  int64_t next = 0;
  std::unordered_map<uint64_t, std::deque<int64_t>> noteids;
  int64_t NoteOnId(uint64_t key) { ++next; noteids[key].push_back(next); return next; }
  std::optional<int64_t> NoteOffId(uint64_t key) {
    auto it = noteids.find(key);
    if (it == noteids.end()) return {}; // the reference asserts; the boundary refuses such an event
    int64_t res = it->second.front();
    if (it->second.size() == 1) noteids.erase(it); else it->second.pop_front();
    return res;
  }
};

// ---- the band-pass bank (vocoder.cpp:46-79, 310-394, 667-732) ---------------------------------------------------------
// Filter<T, 1, HIGH_PASS, 1> (L0, absent).  Certain: first order, and the band-pass compensation identity of
// audioelement.h:2129-2143 fixes |H_hp(f)|^2 = (f/fc)^2 / (1 + (f/fc)^2) beside |H_lp|^2 = 1 / (1 + (f/fc)^2).
// Decision: the complement of the one-pole low-pass decided in oracle_dsp.hpp with the same coefficient, hp = x - lp(x)
// (which is the RC high-pass y[n] = a (y[n-1] + x[n] - x[n-1]), a = 1 / (1 + pi w)).  initWithFreq(sr, f) = initWithAngleIncrement(2 f / sr).
struct OnePoleHighPass {
  OnePoleLowPass lp;
  double out = 0;
  void initWithAngleIncrement(double w) { lp.initWithAngleIncrement(w); }
  void feed(double x) { lp.feed(x); out = x - lp.filtered(); }
  double filtered() const { return out; }
};
inline double angle_increment_of_freq(int sample_rate, double freq) { return 2 * freq / sample_rate; } // sound.functions.h:58-62

struct BandPass { // vocoder.cpp:46-64
  OnePoleHighPass hp_;
  OnePoleLowPass lp_;
  void setup(int sample_rate, double freq_from, double freq_to) {
    hp_.initWithAngleIncrement(angle_increment_of_freq(sample_rate, freq_from));
    lp_.initWithAngleIncrement(angle_increment_of_freq(sample_rate, freq_to));
  }
  double feed(double sample) { hp_.feed(sample); lp_.feed(hp_.filtered()); return lp_.filtered(); }
};
struct EnvelopeFollower { // vocoder.cpp:66-79
  OnePoleLowPass lp_;
  void setup(int sample_rate, double freq) { lp_.initWithAngleIncrement(angle_increment_of_freq(sample_rate, freq)); }
  double feed(double sample) { lp_.feed(sample); return lp_.filtered(); }
};
struct BankVocoder { // Modulator<1, double> :310-394 and Carrier<1, double> :667-732 inside the Vocoder compute lambda :763-812
  std::vector<BandPass> mod_band_pass, car_band_pass;
  std::vector<EnvelopeFollower> env_follower;
  std::vector<double> freqs, res;
  void setup(ora_config const &c);
  double feed(double modulator, double carrier) {
    int const sz = static_cast<int>(res.size());
    for (int i = 0; i < sz; ++i) { // Modulator::feed
      double const filtered = mod_band_pass[i].feed(modulator);
      res[i] = env_follower[i].feed(std::abs(filtered));
    }
    double out{}; // Carrier::feed
    for (int i = 0; i < sz; ++i) {
      double const filtered = car_band_pass[i].feed(carrier);
      out += res[i] * filtered;
    }
    return out;
  }
};

} // namespace ora
