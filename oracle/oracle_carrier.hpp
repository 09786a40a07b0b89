/*
 * TEST INFRASTRUCTURE — CPU oracle, part 5: the vocoder carrier synth and the time-domain band-pass vocoder.
 *
 *  (a) VocoderCarrierElement<double> (source/rt.resynth.lib.cpp:21-52) with its initializer (:137-196), driven by the mono
 *      SynthVocoderCarier (:212-221) = the same ImplCRTP as the resynthesis synth (oracle_synth.hpp: SynthT).
 *      Restated from include/audioelement.h: Phased (:1395-1477), FOscillatorAlgo_ (:1651-1676), PulseTrainAlgo_ (:1691-1748),
 *      soundBufferWrapperAlgo (:1507-1583), UnityGainMix (:1882-1982), BaseVolumeAdjusted / VolumeAdjusted (:1110-1265),
 *      Enveloped (:181-291), FreqCtrl_ (:3055-3155), InterpolatedFreq (:2705-2817), and include/sound.functions.h:86-138.
 *      The noise voice reads an L0 sound buffer (getWhiteNoise, absent): it is stepped like the reference's (an index), its samples
 *      are unknown, and a configuration with a non-zero noise volume is refused — with volume 0 its contribution is 0 * sample = 0,
 *      and its constant angleIncrements() (0.5f, include/sound.h:256-258) still takes part in UnityGainMix::angleIncrements().
 *  (b) the band-pass bank the author replaced by FFTs (source/rt.resynth.lib.vocoder.cpp:46-79 BandPass / EnvelopeFollower,
 *      :310-394 Modulator, :667-732 Carrier; orders 1, :737-738).  Its L0 filters are PARITY UNPINNED like every Filter<> on this
 *      path; decisions below.
 */
#pragma once
#include "oracle_synth.hpp"
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
// soundBufferWrapperAlgo<Sound::NOISE> audioelement.h:1507-1583: an index into an L0 buffer whose samples are not available here
struct NoiseStub {
  int index = -1;
  void forgetPastSignals() {}
  void synchronizeAngles(NoiseStub const &o) { index = o.index; }
  void setStartAngle(double) {} // the index it derives needs the buffer's size (L0); irrelevant at volume 0
  void setAngleIncrements(double) {}
  double angleIncrements() const { return 0.5f; } // FGetBuffer<NOISE>::getAngleIncrements, include/sound.h:256-258
  void step() { ++index; }
  double imag() const { return 0.0; } // times a volume that is required to be 0
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
  VolAdj<NoiseStub> noise;
  VolAdj<Phased<SawAlgo>> saw;
  VolAdj<Phased<TriangleAlgo>> triangle;
  VolAdj<Phased<SquareAlgo>> square;
  VolAdj<SineVoice> sine;
  VolAdj<Phased<PulseTrainAlgo_>> pulse;
  template <typename F> void each(F f) { f(noise); f(saw); f(triangle); f(square); f(sine); f(pulse); }
  template <typename F> void each(F f) const { f(noise); f(saw); f(triangle); f(square); f(sine); f(pulse); }
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
