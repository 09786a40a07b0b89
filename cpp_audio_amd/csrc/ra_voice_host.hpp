// Host half of K9 (gen.voice): what a note-on makes of a SoundEngine — the Markov chain's ramp specs and the
// orchestration's schedule — computed once per note with integer / float arithmetic, so that every ramp of every note becomes an
// independent job of the device kernels (ra_voice.hip).  No HIP in this file: ra_voice_expand_note exposes it to CPU tests.
//
// Reference: ImplBase<MODE>::setupAudioElement (include/gen.voice.h:883-1035); SoundEngine::play / create_birds / create_robot /
// create_sweep / initialize_* / do_initialize (include/soundengine.h:497-562, 565-719, 883-1003); orchestrate_algos / playNextSpec
// (:744-803) over the ramps' AHDSREnvelope (include/audioelement.h:702-781, 923-1084).
// cpp.algorithms' MarkovChain, mersenne<> and itp:: are absent: the decisions of DESIGN.md §5 (declared, parity unpinned).
// WIND adds the pink-noise sound buffer its voices and controls read (include/noise.h:82-165, source/sound.cpp:102-132,156-168) and the start
// positions of a ramp's six iterators (include/sound.h:196-200), both over declared draws.
#pragma once
#include "../../include/resynth_amd.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace rav {

inline uint64_t mix64(uint64_t x) { // splitmix64 finaliser: the declared counter hash of this build
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// stands in for mersenne<SEEDED::Yes>() after `.seed(value)` (gen.voice.h:925): draw k of the note
struct Draws {
  uint64_t key;
  uint32_t value, k = 0;
  Draws(uint64_t key_, uint32_t value_) : key(key_), value(value_) {}
  float unit() {
    const uint64_t h = mix64(key ^ mix64(((uint64_t(value) << 32) | k) * 2 + 5));
    ++k;
    return float(uint32_t(h >> 40)) * (1.0f / 16777216.0f);
  }
  float real(float lo, float hi) { return (hi - lo) * unit() + lo; } // std::uniform_real_distribution<float>{lo, hi}
  int coin() { return unit() < 0.5f ? 0 : 1; }                         // std::uniform_int_distribution<>{0, 1}
};

struct Spec { // RampSpecs::Ctrl (soundengine.h:1049-1070): a LogRamp as set up, the ramp's volume, whether a pause may follow
  double from, to, dur, start;
  int32_t interp;
  float vol;
  bool silence;
};
struct Play { // one playNextSpec (soundengine.h:776-803) of the schedule
  int64_t k;      // element step (1-based) in which the ramp is started: its first step is taken in that same element step
  int32_t ramp;   // which of the engine's three ramps
  int32_t spec;
  int32_t T;      // steps until the key release: max(int(.5f + duration) - xfade_len, 0); 0 = the ramp is skipped (audioelement.h:773-776)
};
struct Note {
  std::vector<Spec> specs;
  std::vector<Play> plays;
  int64_t done_step = 0;  // element steps after which every ramp is EnvelopeDone2 and the spec list is exhausted
  float pan = 0.f;
  int32_t A = 0, R = 0;   // attack / release steps of every ramp's envelope
};

inline int ms_to_frames(float ms, int sr) { return int(0.5f + (float(sr) / 1000.f) * ms); } // sound.functions.h:27-33
inline float inc_of(float freq, int sr) { return 2 * freq / sr; }                           // freq_to_angle_increment<float>, :57-62
inline float clamp01(float v) { return v > 1.f ? 1.f : (v < 0.f ? 0.f : v); }

// The engine while its chain runs: everything play() and the node lambdas read or write.
struct Builder {
  const ra_voice_config &c;
  Draws &rnd;
  std::vector<Spec> &specs;
  int sr;
  float base_freq, length, min_exp, max_exp, phase_ratio1, freq_scatter;
  float state_freq = 0.f, state_factor = 0.f;
  float f1 = 0.f, f2 = 0.f, vol1 = 1.f, vol2 = 1.f; // freq1_robot / freq2_robot (also the sweep's ends)
  int interp, freq_interp, xfade_freq, freq_xfade;
  int cur = -1; // RampSpecs::it while building: index of the spec built last
  static constexpr int kMaxSpecs = 30;

  Spec *next_spec() { // get_next_ramp_for_build :1072-1081
    if ((int)specs.size() == kMaxSpecs) return nullptr;
    specs.push_back(Spec{0, 0, 0, 0, RA_ITP_LINEAR, 1.f, true});
    cur = (int)specs.size() - 1;
    return &specs.back();
  }
  float drawn_frames(float len) {
    len *= std::pow(2.f, rnd.real(min_exp, max_exp));
    return float(ms_to_frames(len, sr));
  }
  void tone(double from, double to, float frames, float vol, bool silence) {
    if (Spec *s = next_spec()) *s = Spec{from, to, double(frames), double(phase_ratio1 * frames), interp, vol, silence};
  }
  // SoundEngine::play :497-562
  void play(float fa, float fb) {
    const float frames = drawn_frames(length);
    if (frames <= 0) return;
    const int before = (cur >= 0 && cur < (int)specs.size()) ? cur : -1; // get_current() before the build
    Spec *s = next_spec();
    if (!s) return;
    if (state_freq != fa) {
      const float sc = 1.f + freq_scatter;
      state_factor = rnd.real(1.f / sc, sc);
    }
    state_freq = fb;
    fa *= state_factor;
    fb *= state_factor;
    *s = Spec{double(inc_of(fa, sr)), double(inc_of(fb, sr)), double(frames), 0.0, interp, 1.f, true};
    if (xfade_freq == RA_XFADE_FREQ_NO || before < 0) return;
    double a = specs[before].to;
    const double b = s->from;
    if (xfade_freq != RA_XFADE_FREQ_ALL && a - b == 0.0) return;
    const Spec moved = *s;
    if ((int)specs.size() == kMaxSpecs) { specs.pop_back(); cur = (int)specs.size() - 1; return; } // cancel_last_ramp :1083-1086
    const int at = cur;
    specs.push_back(moved);
    cur = (int)specs.size() - 1;
    if (a == b) a *= 1.00001f;
    specs[at] = Spec{a, b, double(freq_xfade), 0.0, freq_interp, 1.f, true};
  }
  double transposed(double f, int n) const { return f * std::pow(std::pow(2., 1. / 12.), double(n)); } // Midi::transpose_frequency, midi.h:266-269
  // the node lambdas: `enter` = MarkovMove::ENTER_NODE, else LEAVE_NODE
  void visit(int mode, int node, bool enter) {
    if (mode == RA_VOICE_BIRDS) { // create_birds :565-595
      if (node == 1) { if (enter) play(base_freq * 4, base_freq * 3); else play(base_freq * 2, base_freq * 4); }
      else if (node == 2) { if (enter) play(base_freq * 4, base_freq * 3); }
    } else if (mode == RA_VOICE_ROBOTS) { // create_robot :597-695
      if (node == 0 && !enter) {
        const float fr = drawn_frames(length);
        tone(inc_of(f1, sr), inc_of(f1, sr), fr, vol1, false);
        tone(inc_of(f2, sr), inc_of(f2, sr), fr, vol2, true);
      } else if (node == 2 && enter) {
        const float fr = drawn_frames(length);
        tone(inc_of(f2, sr), inc_of(f2, sr), fr, vol2, true);
        for (int up = 2; up <= 4; up += 2) { const double inc = 2 * transposed(f2, up) / sr; tone(inc, inc, fr, vol2, true); }
      } else if (node == 3 && enter) {
        const float fr = drawn_frames(2.f * length);
        tone(inc_of(f2, sr), inc_of(f1, sr), fr, std::min(vol2, vol1), true);
      }
    } else if (mode == RA_VOICE_WIND) { // create_wind :721-735: node 0's lambda takes a spec whichever way the chain moves; the spec's control plays for
      // INT_MAX / 2 samples (ShortTermNoiseAdderCtrl::get_duration_in_samples, :151-154), its long-term iterator takes the engine's interpolation
      if (node == 0) { if (Spec *s = next_spec()) { s->dur = double(INT_MAX / 2); s->interp = interp; } }
    } else { // create_sweep :697-719
      if (node == 0 && !enter) {
        const float fr = drawn_frames(length);
        if (Spec *s = next_spec()) *s = Spec{double(inc_of(f1, sr)), double(inc_of(f2, sr)), double(fr), double(phase_ratio1 * fr), interp, 1.f, true};
      }
    }
  }
};

// float n_slow_steps = sample_rate * std::pow(max_seconds_slow_iter, denorm<SECONDS_SLOW_ITER_*>()), gen.voice.h:623,964-965
inline float wind_slow_steps(int sr, float param) { constexpr float max_seconds_slow_iter = 2.268f; return sr * std::pow(max_seconds_slow_iter, param); }

struct Edge { int from, to; float p; };
inline const Edge *edges_of(int mode, int &n) {
  static const Edge birds[] = {{0, 1, 0.5f}, {1, 0, 0.015f}, {0, 2, 0.5f}, {2, 0, 0.015f}, {2, 1, 0.885f}};
  static const Edge robot[] = {{0, 1, 1.f}, {1, 2, 0.2f}, {2, 1, 0.1f}, {1, 3, 0.2f}, {3, 1, 1.f}};
  static const Edge sweep[] = {{0, 1, 1.f}};
  static const Edge wind[] = {{0, 1, 1.f}, {1, 0, 1.f}};
  if (mode == RA_VOICE_WIND) { n = 2; return wind; }
  if (mode == RA_VOICE_BIRDS) { n = 5; return birds; }
  if (mode == RA_VOICE_ROBOTS) { n = 5; return robot; }
  n = 1;
  return sweep;
}
// MarkovChain::step / step_normalized as decided: the first transition (definition order) whose cumulated probability exceeds r
inline void chain_step(int mode, int &node, float r, bool normalized, bool exec, Builder &b) {
  int ne;
  const Edge *e = edges_of(mode, ne);
  if (normalized) {
    float sum = 0.f;
    for (int i = 0; i < ne; ++i) if (e[i].from == node) sum += e[i].p;
    if (!(sum > 0.f)) return;
    r *= sum;
  }
  float acc = 0.f;
  for (int i = 0; i < ne; ++i) {
    if (e[i].from != node) continue;
    acc += e[i].p;
    if (r < acc) {
      if (exec) { b.visit(mode, node, false); b.visit(mode, e[i].to, true); }
      node = e[i].to;
      return;
    }
  }
}

inline bool config_error(const ra_voice_config &c, const char *&why) {
  why = nullptr;
  if (c.sample_rate < 1000) why = "sample_rate is too small";
  else if (c.n_channels < 1 || c.n_channels > 256) why = "n_channels must be in 1..256";
  else if (c.mode < RA_VOICE_BIRDS || c.mode > RA_VOICE_WIND) why = "mode must be a RA_VOICE_* value";
  else if (c.mode == RA_VOICE_SWEEP && (!(c.low_freq > 0.f) || !(c.high_freq > 0.f))) why = "the sweep's ends must be positive frequencies (LogRamp, audioelement.h:2566-2567)";
  else if (!(c.length_ms > 0.f) || !(c.length_exponent >= 0.f) || c.length_exponent > 7.f) why = "bad length / length exponent";
  else if (ms_to_frames(c.length_ms, c.sample_rate) <= 0) why = "length is too small (soundengine.h:966-970)";
  else if (!(c.xfade_length_seconds >= 0.f) || c.xfade_length_seconds > 10.f) why = "bad xfade length";
  else if (c.loudness_ref_freq_index < 0 || c.loudness_ref_freq_index > 15) why = "loudness_ref_freq_index must be in 0..15";
  else if (!(c.loudness_compensation_amount >= 0.f && c.loudness_compensation_amount <= 1.f)) why = "loudness_compensation_amount must be in 0..1";
  else if (c.interpolation < 0 || c.interpolation >= RA_ITP_END) why = "interpolation must be a ra_itp value";
  else if (c.mode != RA_VOICE_SWEEP) {
    if (c.markov_start_node < 0 || c.markov_start_node > 2) why = "markov_start_node must be in 0..2";
    else if (c.markov_pre_tries < 0 || c.markov_pre_tries > 20 || c.markov_min_path_length < 0 || c.markov_min_path_length > 20 || c.markov_additional_tries < 0 ||
             c.markov_additional_tries > 20)
      why = "the Markov step counts must be in 0..20";
    else if (!(c.articulative_pause_seconds >= 0.f) || c.articulative_pause_seconds > 0.5f) why = "articulative_pause_seconds must be in 0..0.5";
    else if (!(c.freq_scatter >= 0.f) || c.freq_scatter > 1.f) why = "freq_scatter must be in 0..1";
    else if (!(c.length_exponent_scatter >= 0.f) || c.length_exponent_scatter > 1.f) why = "length_exponent_scatter must be in 0..1";
    else if (c.seed_param < 0 || c.seed_param > 257) why = "seed_param must be in 0..257";
    else if (c.mode == RA_VOICE_ROBOTS && (!(c.d1 >= 0.f) || c.d1 > 47.f || !(c.d2 >= 0.f) || c.d2 > 47.f)) why = "d1 / d2 must be in 0..47";
    else if (c.mode == RA_VOICE_ROBOTS && !(c.harmonic_attenuation > 0.f)) why = "harmonic_attenuation must be > 0 (RampSpecs::Ctrl::setVolume, soundengine.h:1054-1057)";
    else if (c.mode == RA_VOICE_BIRDS && (c.xfade_freq < 0 || c.xfade_freq > 2)) why = "xfade_freq must be a RA_XFADE_FREQ_* value";
    else if (c.mode == RA_VOICE_BIRDS && (!(c.freq_transition_seconds >= 0.f) || c.freq_transition_seconds > 0.5f)) why = "freq_transition_seconds must be in 0..0.5";
    else if (c.mode == RA_VOICE_BIRDS && (c.freq_interpolation < 0 || c.freq_interpolation >= RA_ITP_END)) why = "freq_interpolation must be a ra_itp value";
    else if (c.mode == RA_VOICE_BIRDS && c.xfade_freq != RA_XFADE_FREQ_NO && int(c.freq_transition_seconds * c.sample_rate) < 1)
      why = "a frequency transition needs a length of at least one frame (LogRamp::setup, audioelement.h:2518)";
    else if (c.mode == RA_VOICE_WIND) {
      if (c.pink_noise_lp_gain != 0.f)
        why = "pink_noise_lp_gain must be 0: the reference never sets the frequency of that voice's filter (audioelement.h:2079-2086), its output is undefined";
      else if (c.markov_start_node > 1) why = "WIND's chain has two nodes: markov_start_node must be 0 or 1";
      else if (!(c.pink_noise_bp_gain >= 0.f && c.pink_noise_bp_gain <= 1.f) || !(c.pink_noise_br_gain >= 0.f && c.pink_noise_br_gain <= 1.f)) why = "the noise gains must be in 0..1";
      else if (!(c.sine_gain >= 0.f && c.sine_gain <= 1.f)) why = "sine_gain must be in 0..1";
      else if (c.order_filters < 1 || c.order_filters > 258) why = "order_filters must be in 1..258";
      else if (!(c.bp_octave_width_min >= 0.f && c.bp_octave_width_min <= 5.f) || !(c.bp_octave_width_max >= 0.f && c.bp_octave_width_max <= 5.f)) why = "the band widths must be in 0..5 octaves";
      else if (!(c.center_octave_min_long_term >= 0.f && c.center_octave_min_long_term <= 10.f) || !(c.center_octave_max_long_term >= 0.f && c.center_octave_max_long_term <= 10.f))
        why = "the centre octaves must be in 0..10";
      else if (!(c.center_short_term_ratio >= 0.f && c.center_short_term_ratio <= 1.f)) why = "center_short_term_ratio must be in 0..1";
      else if (!(wind_slow_steps(c.sample_rate, c.seconds_slow_iter_long_term) >= 1.f) || !(wind_slow_steps(c.sample_rate, c.seconds_slow_iter_long_term) < 1e9f))
        why = "seconds_slow_iter_long_term: a slow iterator needs at least one step per noise sample (SlowIter, audioelement.h:2862-2877)";
      else if (!(wind_slow_steps(c.sample_rate, c.seconds_slow_iter_short_term) >= 0.f) || !(wind_slow_steps(c.sample_rate, c.seconds_slow_iter_short_term) < 1e9f)) why = "bad seconds_slow_iter_short_term";
    }
  }
  return why != nullptr;
}

inline int xfade_len_of(const ra_voice_config &c) { return int(0.5f + (c.xfade_length_seconds * c.sample_rate)); } // setEnvelopeCharacTime, gen.voice.h:893

// setupAudioElement + initialize_<mode> for one note-on, then the schedule of the orchestration.
inline void expand_note(const ra_voice_config &c, float frequency, uint64_t stream_seed, int32_t seed_value, int64_t noteid, Note &out) {
  out.specs.clear();
  out.plays.clear();
  Draws rnd(stream_seed, (uint32_t)seed_value);
  const int sr = c.sample_rate;
  Builder b{c, rnd, out.specs, sr};
  b.interp = c.interpolation;
  b.length = c.length_ms;
  const float ex = c.length_exponent;
  if (c.mode != RA_VOICE_SWEEP) {
    b.min_exp = ex * (1.f - c.length_exponent_scatter);
    b.max_exp = ex * (1.f + c.length_exponent_scatter);
    b.freq_scatter = c.freq_scatter;
    b.phase_ratio1 = clamp01(c.phase_ratio1);
  } else {
    b.min_exp = b.max_exp = ex;
    b.freq_scatter = 0.f;
    b.phase_ratio1 = 0.f;
  }
  { // engine.setAngleIncrements(freq_to_angle_increment(freq, sr)) -> base_freq (soundengine.h:393-395)
    const double ai = double(inc_of(frequency, sr));
    b.base_freq = float(ai * 0.5 * sr);
  }
  // the pan (gen.voice.h:980-995): the first draw after the reseed; SWEEP does not reseed: the declared hash with salt 3
  if (c.random_pan) {
    if (c.mode == RA_VOICE_SWEEP) {
      const uint64_t h = mix64(stream_seed ^ mix64(uint64_t(noteid) * 2 + 3));
      out.pan = -1.0f + 2.0f * (float(uint32_t(h >> 40)) * (1.0f / 16777216.0f));
    } else out.pan = rnd.real(-1.f, 1.f);
  } else out.pan = c.pan;
  b.xfade_freq = (c.mode == RA_VOICE_BIRDS) ? c.xfade_freq : RA_XFADE_FREQ_NO;
  b.freq_xfade = int(c.freq_transition_seconds * sr);
  b.freq_interp = c.freq_interpolation;
  int start = 0, pre = 0, path = 1, extra = 0;
  if (c.mode == RA_VOICE_SWEEP) { b.f1 = c.low_freq; b.f2 = c.high_freq; }
  else { start = c.markov_start_node; pre = c.markov_pre_tries; path = c.markov_min_path_length; extra = c.markov_additional_tries; }
  if (c.mode == RA_VOICE_ROBOTS) { // initialize_robot :926-961
    const float sc = 1.f + c.freq_scatter;
    const float detune = 0.985f;
    b.f1 = rnd.real(b.base_freq / sc, b.base_freq * sc);
    b.f2 = rnd.real(b.f1 * detune, b.f1 / detune);
    const float att = std::min(0.99f, std::max(0.f, c.harmonic_attenuation)); // set_har_att :827-837
    if (!rnd.coin()) { b.f1 = float(b.transposed(b.f1, int(c.d1))); b.vol1 = std::pow(att, c.d1); }
    else { b.f2 = float(b.transposed(b.f2, int(c.d2))); b.vol2 = std::pow(att, c.d2); }
  }
  // do_initialize :963-1003
  int node = start;
  for (int i = 0; i < pre; ++i) chain_step(c.mode, node, rnd.real(0.f, 1.f), true, false, b);
  for (int i = 0; i < path; ++i) chain_step(c.mode, node, rnd.real(0.f, 1.f), true, true, b);
  for (int i = 0; i < extra; ++i) chain_step(c.mode, node, rnd.real(0.f, 1.f), false, true, b);

  // ---- the schedule: orchestrate_algos / playNextSpec over three ramps whose envelopes are AHDSR{xfade_len, 0, 0, xfade_len, 1} ----
  // A ramp started in element step k with T steps to its release is KeyPressed for T steps, KeyReleased for R, EnvelopeDone1 for 17
  // (audioelement.h:702-756); the orchestration sees it finished from step k + T + R + 17 on (k + 17 when T = 0: skipped at once).
  const int xf = xfade_len_of(c), min_dt = sr / 1000;
  out.A = std::max(xf, min_dt);
  out.R = std::max(xf, min_dt);
  const int pause = (c.mode == RA_VOICE_SWEEP) ? 0 : int(c.articulative_pause_seconds * sr);
  int64_t free_at[3] = {1, 1, 1};
  int64_t due = 1; // the element step in which playNextSpec is first reached for the next spec
  const int ns = (int)out.specs.size();
  for (int i = 0; i <= ns; ++i) {
    int64_t k = std::max(due, std::min(free_at[0], std::min(free_at[1], free_at[2]))); // retried every step until a ramp is finished
    if (i == ns) { out.done_step = k; break; } // get_next_ramp_for_run finds the list exhausted: ramp_specs.done() from now on
    int ramp = 2;
    while (free_at[ramp] > k) --ramp; // get_ramps: the last finished ramp wins (:277-296)
    const Spec &s = out.specs[i];
    // (WIND: .5f + an int — a float sum, soundengine.h:797-798 over :151-154)
    const int T = (c.mode == RA_VOICE_WIND) ? std::max(int(.5f + float(INT_MAX / 2)) - xf, 0) : std::max(int(.5f + s.dur) - xf, 0);
    out.plays.push_back(Play{k, ramp, i, T});
    free_at[ramp] = (T > 0) ? k + T + out.R + 17 : k + 17;
    // the release is seen T steps later (countStepsToRelease() == 0), then the pause if one follows; a skipped ramp is never seen pressed
    due = (T > 0) ? k + T + ((s.silence && pause > 0) ? pause : 0) : k + 1;
  }
  out.done_step = std::max(out.done_step, std::max(free_at[0], std::max(free_at[1], free_at[2])) - 1);
}

// ---- WIND: the pink-noise sound buffer (getPinkNoise, source/sound.cpp:7-10) --------------------------------------------------------------
// soundBuffer<double>(soundId{sr, PINK_NOISE, .05f}): int(sr / .05f) frames (sound.functions.h:45-50) of the Voss-McCartney generator
// (GaussianPinkNoiseAlgo, include/noise.h:82-165: relevantBits(sr / 20) rows of white noise, row i redrawn every 2^i samples and linearly
// interpolated in between, the sample = their mean), the buffer's last tenth cross-faded (sin / cos gains) into samples the same generator
// produced BEFORE the buffer's own (generate_with_smooth_transition, sound.cpp:102-132), then scaled to a peak just below 1
// (normalize_audio, include/sound.h:96-118).  The generator's white draws: draw k = Box-Muller of the counter hash of ("pinknois", k), salt 9.
struct PinkNoise {
  std::vector<double> values;
  float abs_mean = 0.f; // getPinkNoiseAbsMean (sound.cpp:22-43)
};
inline float pink_draw(uint64_t k) {
  const uint64_t h = mix64(0x70696e6b6e6f6973ull ^ mix64(k * 2 + 9));
  const double u1 = (double(h >> 40) + 1.0) * (1.0 / 16777216.0), u2 = double((h >> 16) & 0xffffffull) * (1.0 / 16777216.0);
  return float(std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2));
}
inline void build_pink_noise(int sr, PinkNoise &out) {
  int rows = 0;
  for (unsigned v = unsigned(sr / (10.f * 2.f)); v; v >>= 1) ++rows; // relevantBits
  struct Row { float value = 0.f, prev = 0.f; int at = 0; };
  std::vector<Row> row((size_t)rows);
  uint64_t draws = 0;
  unsigned counter = 0;
  auto redraw = [&](int i) { Row &r = row[(size_t)i]; r.at = int(counter); r.prev = r.value; r.value = pink_draw(draws++); };
  auto redraw_by_counter = [&] { // the row whose period divides the counter: 1 + count_trailing_zeroes, the last row for 0 and beyond
    unsigned i = counter ? 1u + unsigned(__builtin_ctz(counter)) : 33u;
    redraw(int(std::min<unsigned>(i, unsigned(rows - 1))));
  };
  for (int i = rows - 1; i >= 1; --i) { counter = (1u << (rows - 2)) - (1u << (i - 1)); redraw_by_counter(); } // the constructor staggers the rows
  auto next = [&] {
    ++counter;
    redraw(0);
    redraw_by_counter();
    float sum = 0.f;
    for (int i = 0; i < rows; ++i) {
      const Row &r = row[(size_t)i];
      const float w = (1 + int(counter) - r.at) / float(1 << i);
      sum += w * r.value + (1.f - w) * r.prev;
    }
    return sum / size_t(rows);
  };
  const int period = std::max(1, int(float(sr) / .05f));
  const int fade = std::min(10 + period / 10, period);
  std::vector<double> before((size_t)fade);
  for (double &v : before) v = next();
  out.values.resize((size_t)period);
  for (double &v : out.values) v = next();
  for (int i = 0; i < fade; ++i) {
    const float part = (i + 1) / float(fade + 1);
    const float angle = part * float(M_PI_2);
    const float g_before = float(std::sin(double(angle))), g_own = float(std::cos(double(angle)));
    double &v = out.values[out.values.size() - (size_t)fade + (size_t)i];
    v = g_before * before[(size_t)i] + g_own * v;
  }
  float lo = float(out.values[0]), hi = lo;
  for (double v : out.values) { lo = std::min(lo, float(v)); hi = std::max(hi, float(v)); }
  const float scale = (1.f - FLT_EPSILON) / std::max(-lo, hi);
  double sum = 0;
  for (double &v : out.values) { v *= scale; sum += std::abs(v); }
  out.abs_mean = float(sum / double(out.values.size()));
}
// BufferIter::initializeForRun's start position (include/sound.h:196-200) as declared: the counter hash of what the draw is for — the
// stream's key, the note, which spec is being played, which of the ramp's six iterators, initializeForRun (0) or set_sample_rate (1); salt 11
inline int32_t wind_position(uint64_t stream_seed, int64_t noteid, int play, int iter, int which, size_t size) {
  const uint64_t id = ((uint64_t(noteid) * 64 + uint64_t(play)) * 8 + uint64_t(iter)) * 2 + uint64_t(which);
  const uint64_t h = mix64(stream_seed ^ mix64(id * 2 + 11));
  const double u = double(h >> 11) * (1.0 / 9007199254740992.0);
  return int32_t(u * double(float(size - 1)));
}

} // namespace rav
