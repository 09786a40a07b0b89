/*
 * TEST INFRASTRUCTURE — C entry points of the CPU oracle for ctypes (tests/, smoke(), bench.py cpu_baseline).
 * Not part of the product; nothing under cpp_audio_amd/ or include/ links this.
 */
#include "oracle_resynth.hpp"
#include "oracle_voice.hpp"
#include <cstring>
#include <memory>
#include <thread>

using namespace ora;

extern "C" {

void ora_default_config(ora_config *c) { *c = default_config(); }
// 0: explicit log2 / log10 / exp10 / pow (bit-comparable with the product); 1: the platform libm as the reference's text calls it
void ora_set_libm_mode(int on) { libm_mode() = on ? 1 : 0; }

int ora_even_window_size_frames(float seconds, int sr) { return even_window_size_frames(seconds, sr); }
int ora_window_center_stride_frames(float seconds, int sr) { return window_center_stride_frames(seconds, sr); }
int ora_good_stride(float seconds, int sr) { return good_stride(seconds, sr); }
int ora_modulator_window_size(float seconds, int sr) { return modulator_window_size(static_cast<double>(seconds), sr); }
int ora_fft_length_for(int n) { return get_fft_length_for(n, 1); }

// ---- known-answer helpers (rt.resynth.lib.test.cpp) ---------------------------------------------
void ora_note_add_halftones(int note, int octave, long n, int *out_note, int *out_octave) {
  NoteOctave r = NoteOctave{note, octave}.add_halftones(n);
  *out_note = r.note;
  *out_octave = r.octave;
}
int ora_note_dist_halftones(int note, int octave, int note2, int octave2) { return NoteOctave{note, octave}.dist_halftones(NoteOctave{note2, octave2}); }
double ora_midi_get_pitch(int note, int octave) { return Midi().get_pitch(NoteOctave{note, octave}); }
double ora_scale_closest_pitch(int mode, double root_pitch, double pitch) { return getMusicalScale(mode).closest_pitch(root_pitch, pitch); }
double ora_frequency_to_midi_pitch(double f) { auto p = Midi().frequency_to_midi_pitch(f); return p ? *p : NAN; }
double ora_midi_pitch_to_freq(double p) { return Midi().midi_pitch_to_freq(p); }
// the explicit log2 / exp10 of the analysis path (oracle_math.hpp), for the known-answer test against libm
void ora_det_math(const double *x, int64_t n, int which, double *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = which == 0 ? det_log2(x[i]) : which == 1 ? det_log10(x[i]) : which == 2 ? det_exp10(x[i]) : Midi().midi_pitch_to_freq_event(x[i]);
}
float ora_equal_loudness_volume_db(double pitch, int level) { return loudness::equal_loudness_volume_db(pitch, level); }
// the white-noise sound buffer of the carrier's noise voice (oracle_carrier.hpp): returns its length, copies the first min(cap, length) samples
int64_t ora_white_noise(int32_t sample_rate, double *out, int64_t cap) {
  auto const &b = getWhiteNoise(sample_rate);
  for (int64_t i = 0; i < cap && i < static_cast<int64_t>(b.size()); ++i) out[i] = b[i];
  return static_cast<int64_t>(b.size());
}
// the pink-noise sound buffer of gen.voice's WIND (oracle_wind.hpp): the same, and its abs mean
int64_t ora_pink_noise(int32_t sample_rate, double *out, int64_t cap, float *abs_mean) {
  auto const &b = getPinkNoise(sample_rate);
  for (int64_t i = 0; i < cap && i < static_cast<int64_t>(b.values.size()); ++i) out[i] = b.values[static_cast<size_t>(i)];
  if (abs_mean) *abs_mean = b.abs_mean;
  return static_cast<int64_t>(b.values.size());
}
// one playNextSpec's start position of one of a WIND ramp's six iterators (oracle_wind.hpp: the declared draws)
int32_t ora_wind_start_position(uint64_t key, int64_t noteid, int32_t play, int32_t iter, int32_t which, int64_t size) {
  WindDrawCtx c; c.key = key; c.noteid = noteid; c.play = play;
  return wind_start_position(c, iter, which, static_cast<size_t>(size));
}
void ora_stereo(double pan, double *l, double *r) { auto g = stereo(pan); *l = g.gains[0]; *r = g.gains[1]; }
float ora_hash_uniform_pm1(uint64_t seed, uint64_t noteid, uint64_t salt) { return hash_uniform_pm1(seed, noteid, salt); }

double ora_autotune(const ora_config *c, double pitch) {
  Autotune a;
  a.build(c->use_autotune, c->autotune_type, c->autotune_scale_mode, c->autotune_root_note, c->autotune_root_transpose,
          c->autotune_chord_frequencies, c->autotune_bit_chord, c->autotune_factor);
  auto r = a(pitch);
  return r ? *r : NAN;
}
int ora_autotune_allowed(const ora_config *c, double *out, int cap) {
  Autotune a;
  a.build(c->use_autotune, c->autotune_type, c->autotune_scale_mode, c->autotune_root_note, c->autotune_root_transpose,
          c->autotune_chord_frequencies, c->autotune_bit_chord, c->autotune_factor);
  int n = static_cast<int>(a.allowed.size());
  for (int i = 0; i < n && i < cap; ++i) out[i] = a.allowed[i];
  return n;
}

// ---- DSP primitives -------------------------------------------------------------------------------
void ora_half_window(int kind, int n, double *out) {
  std::vector<double> hw;
  make_half_window(kind, n, hw);
  std::memcpy(out, hw.data(), sizeof(double) * n);
}
// out: (L/2+1) interleaved complex
void ora_real_fft(const double *x, int L, double *out) {
  RealFft f(L);
  std::vector<std::complex<double>> X(L / 2 + 1);
  f.forward(x, X.data());
  for (int k = 0; k <= L / 2; ++k) { out[2 * k] = X[k].real(); out[2 * k + 1] = X[k].imag(); }
}
void ora_real_ifft(const double *bins, int L, double *x) {
  RealFft f(L);
  std::vector<std::complex<double>> X(L / 2 + 1);
  for (int k = 0; k <= L / 2; ++k) X[k] = {bins[2 * k], bins[2 * k + 1]};
  f.inverse(X.data(), x);
  for (int i = 0; i < L; ++i) x[i] /= L;
}
// returns fft length; out has L/2 values
int ora_sqmag(const double *x, int n, int window_kind, double *out) {
  SqMagOp op;
  op.window_kind = window_kind;
  op.init_data(n);
  op(x, n);
  std::memcpy(out, op.result.frequencies_sqmag.data(), sizeof(double) * op.result.frequencies_sqmag.size());
  return op.result.fft_length;
}
int ora_peaks(double sr, const double *sqmag, int n_bins, int fft_length, double *freq, double *db, int cap) {
  FrequenciesSqMag s;
  s.frequencies_sqmag.assign(sqmag, sqmag + n_bins);
  s.fft_length = fft_length;
  std::vector<FreqMag> out;
  extract_local_max_freqs_mags(sr, s, out);
  int n = static_cast<int>(out.size());
  for (int i = 0; i < n && i < cap; ++i) { freq[i] = out[i].freq; db[i] = out[i].mag_db; }
  return n;
}

// The stateless half of RtResynth::step on one peak list: returns count of autotuned pitches.
int ora_step_stateless(const ora_config *c, const double *freq, const double *db, int n_peaks, double *pitch, double *vol, float *key, int cap) {
  ora_config cfg = *c;
  RtResynthOffline r(cfg);
  std::vector<FreqMag> fs(n_peaks);
  for (int i = 0; i < n_peaks; ++i) fs[i] = {freq[i], db[i]};
  frequencies_to_pitches(r.midi, fs, r.freqmags_data);
  aggregate_pitches(cfg.nearby_distance_tones, r.freqmags_data, r.pitch_intervals);
  reduce_pitches(cfg.min_volume, r.pitch_intervals, r.reduced_pitches);
  shift_pitches(cfg.pitch_shift_pre_autotune, r.reduced_pitches);
  harmonize_pitches(cfg.pitch_harmonize_pre_autotune, r.pitches_tmp, r.reduced_pitches);
  autotune_pitches(static_cast<double>(cfg.autotune_max_pitch), cfg.autotune_tolerance_pitches, r.autotune, r.reduced_pitches, r.autotuned_pitches);
  shift_pitches(cfg.pitch_shift_post_autotune, r.autotuned_pitches);
  harmonize_pitches(cfg.pitch_harmonize_post_autotune, r.pitches_tmp, r.autotuned_pitches);
  order_pitches_by_perceived_loudness(r.autotuned_pitches, r.loudness_keys, r.order);
  int n = static_cast<int>(r.autotuned_pitches.size());
  for (int i = 0; i < n && i < cap; ++i) { pitch[i] = r.autotuned_pitches[i].midipitch; vol[i] = r.autotuned_pitches[i].volume; key[i] = r.loudness_keys[i]; }
  return n;
}

// ---- the offline job ------------------------------------------------------------------------------
typedef struct ora_event {
  int64_t sample;
  int32_t type;    /* 0 on, 1 off, 2 change, 3 dropped note-on */
  int32_t channel;
  int64_t noteid;
  float frequency, velocity;
} ora_event;

// voice / carrier may be NULL.  out: [n_frames][2] float.  Returns number of events (may exceed log_cap; only log_cap are stored).
int64_t ora_run_offline(const ora_config *c, const float *voice, const float *carrier, int64_t n_frames, float *out, ora_event *log,
                        int64_t log_cap, int32_t *dropped_note_on) {
  RtResynthOffline r(*c);
  r.run(voice, carrier, n_frames, out);
  if (dropped_note_on) *dropped_note_on = r.dropped_note_on;
  int64_t n = static_cast<int64_t>(r.event_log.size());
  for (int64_t i = 0; i < n && i < log_cap; ++i) {
    auto const &e = r.event_log[i];
    log[i] = {e.sample, e.type, e.channel, e.noteid, e.frequency, e.velocity};
  }
  return n;
}

// The same with note events for the vocoder carrier synth (sorted by sample).  carrier_out (may be NULL): the carrier sample the vocoder
// was given at every frame, as a double.  dropped[0] = resynthesis note-ons dropped, dropped[1] = carrier note-ons dropped.
int64_t ora_run_offline_ex(const ora_config *c, const float *voice, const float *carrier, int64_t n_frames, const ora_carrier_event *cev,
                           int64_t n_cev, float *out, double *carrier_out, ora_event *log, int64_t log_cap, int32_t *dropped) {
  RtResynthOffline r(*c);
  r.carrier_events.assign(cev, cev + n_cev);
  std::vector<double> dump;
  if (carrier_out) { dump.reserve(n_frames); r.carrier_dump = &dump; }
  r.run(voice, carrier, n_frames, out);
  if (carrier_out) std::memcpy(carrier_out, dump.data(), sizeof(double) * std::min<size_t>(dump.size(), n_frames));
  if (dropped) { dropped[0] = r.dropped_note_on; dropped[1] = r.carrier_dropped_note_on; }
  int64_t n = static_cast<int64_t>(r.event_log.size());
  for (int64_t i = 0; i < n && i < log_cap; ++i) {
    auto const &e = r.event_log[i];
    log[i] = {e.sample, e.type, e.channel, e.noteid, e.frequency, e.velocity};
  }
  return n;
}

// Per-frame autotuned pitch lists of one stream (the output of the stateless half of step()).
// fire_sample[f] = iteration index at which frame f fired; offsets[f]..offsets[f+1] index pitch/vol.
int64_t ora_run_frames(const ora_config *c, const float *voice, int64_t n_frames, int64_t *fire_sample, int64_t *offsets, int64_t frame_cap,
                       double *pitch, double *vol, int64_t item_cap) {
  RtResynthOffline r(*c);
  std::vector<RtResynthOffline::FrameDump> dump;
  r.frame_dump = &dump;
  std::vector<float> out(static_cast<size_t>(n_frames) * 2);
  r.run(voice, nullptr, n_frames, out.data());
  int64_t nf = static_cast<int64_t>(dump.size());
  int64_t pos = 0;
  for (int64_t f = 0; f < nf && f < frame_cap; ++f) {
    fire_sample[f] = dump[f].fire_sample;
    offsets[f] = pos;
    for (auto const &pv : dump[f].pitches) {
      if (pos < item_cap) { pitch[pos] = pv.midipitch; vol[pos] = pv.volume; }
      ++pos;
    }
    offsets[f + 1] = pos;
  }
  return nf;
}

// Many independent streams over n_threads host threads, streams statically partitioned
// (BASELINE.md §3 CPU-baseline plan).  seed of stream s = c->seed + s.
void ora_run_offline_batch(const ora_config *c, const float *voice, const float *carrier, int32_t n_streams, int64_t n_frames, float *out,
                           int32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  auto work = [&](int t) {
    for (int s = t; s < n_streams; s += n_threads) {
      ora_config cfg = *c;
      cfg.seed = c->seed + static_cast<uint64_t>(s);
      RtResynthOffline r(cfg);
      r.run(voice ? voice + static_cast<size_t>(s) * n_frames : nullptr, carrier ? carrier + static_cast<size_t>(s) * n_frames : nullptr, n_frames,
            out + static_cast<size_t>(s) * n_frames * 2);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
  work(0);
  for (auto &t : th) t.join();
}

// ---- driver (A) -----------------------------------------------------------------------------------
// returns the number of frames the reference loop would write; fills out up to cap.
int64_t ora_resynth_notes(const int32_t *start, const int32_t *end, const double *freq, const double *amp_db, int32_t n_notes, int32_t stride,
                          int32_t sr, double *out, int64_t cap) {
  std::vector<DeducedNote> notes(n_notes);
  for (int i = 0; i < n_notes; ++i) notes[i] = {start[i], end[i], freq[i], amp_db[i]};
  std::vector<double> o;
  resynth_notes(notes, stride, sr, o);
  int64_t n = static_cast<int64_t>(o.size());
  for (int64_t i = 0; i < n && i < cap; ++i) out[i] = o[i];
  return n;
}

// gen.voice (oracle_voice.hpp): Impl_::doProcessing (gen.voice.h:1128-1187) over n_frames for one stream — events reach the
// synth at their frame, then SimpleAudioOutContext::step renders.  ev: sample / type (0 on, 1 off) / noteid / frequency / velocity.
// out [n_frames][2] float.  Returns the number of note-ons dropped.
int32_t ora_voice_run(const ora_voice_params *p, const int64_t *ev_sample, const int32_t *ev_type, const int64_t *ev_noteid, const float *ev_freq,
                      const float *ev_vel, int64_t n_ev, int64_t n_frames, float *out) {
  if (p->mode == VOICE_WIND && (p->pink_noise_lp_gain != 0.f || p->order_filters < 1)) return -1; // the low-pass voice's output is undefined in the reference (oracle_wind.hpp)
  Stepper stepper;
  VoiceSynth synth(p->n_channels);
  synth.phase_policy = 1; // DefaultStartPhase::Zero
  VoiceInitializer init;
  init.p = *p;
  synth.initializer = init;
  synth.initialize(stepper);
  int32_t dropped = 0;
  int64_t e = 0;
  for (int64_t k = 0; k < n_frames; ++k) {
    for (; e < n_ev && ev_sample[e] <= k; ++e) {
      Event evt{ev_type[e] == 0 ? EventType::NoteOn : EventType::NoteOff, ev_noteid[e], ev_freq[e], ev_vel[e], 0};
      if (synth.onEvent(p->sample_rate, evt, stepper) != onEventResult::OK) ++dropped;
    }
    stepper.step(out + 2 * k, 1);
  }
  return dropped;
}

// What the engine makes of consecutive note-ons of one synth (no note-off until the note has played out), for the tests of the product's
// host-side note expansion: per note its ramp specs as the Markov chain built them — spec_d [.][4] = {from, to, duration, start sample},
// spec_i [.][3] = {interpolation, silence follows, 0}, spec_vol [.] — and every playNextSpec as it happened while the element was stepped —
// played [.][3] = {element step (1-based), ramp, spec} —, note_info [n_notes][4] = {specs, plays, steps until the element is finished (every
// ramp EnvelopeDone2 and no spec left), pan as float bits}.  Returns 0, or -1 when a capacity is too small / a note does not end in max_steps.
int32_t ora_voice_trace(const ora_voice_params *p, const float *freqs, int32_t n_notes, int64_t max_steps, double *spec_d, int32_t *spec_i, float *spec_vol,
                        int64_t spec_cap, int64_t *played, int64_t played_cap, int64_t *note_info) {
  // WIND: a spec plays for 2^30 steps and a note-off leaves the channel's ramp unfinished for good (the channel is lost): one channel per note,
  // each note stepped for 2000 samples only; its "steps until finished" is reported as -1, its specs as {0, 0, INT_MAX / 2, 0} + the long-term
  // iterator's interpolation
  bool const wind = p->mode == VOICE_WIND;
  Stepper stepper;
  VoiceSynth synth(wind ? std::max(1, n_notes) : 1);
  synth.phase_policy = 1;
  VoiceInitializer init;
  init.p = *p;
  std::vector<VoicePlayed> trace;
  init.trace = &trace;
  synth.initializer = init;
  synth.initialize(stepper);
  int64_t ns = 0, np = 0;
  float out[2];
  for (int32_t i = 0; i < n_notes; ++i) {
    trace.clear();
    Event on{EventType::NoteOn, i + 1, freqs[i], 1.f, 0};
    if (synth.onEvent(p->sample_rate, on, stepper) != onEventResult::OK) return -1;
    VoiceElement &e = synth.channels[wind ? static_cast<size_t>(i) : 0].elem;
    if (!e.setup_ok) return -1;
    int32_t const n_specs = e.ramp_specs.end;
    if (ns + n_specs > spec_cap) return -1;
    for (int32_t s = 0; s < n_specs; ++s, ++ns) {
      auto const &sp = e.ramp_specs.a[s];
      spec_d[4 * ns + 0] = sp.ctrl.from; spec_d[4 * ns + 1] = sp.ctrl.to; spec_d[4 * ns + 2] = sp.ctrl.duration_in_samples; spec_d[4 * ns + 3] = sp.ctrl.cur_sample;
      spec_i[3 * ns + 0] = sp.ctrl.interp; spec_i[3 * ns + 1] = sp.silenceAfter ? 1 : 0; spec_i[3 * ns + 2] = 0;
      if (wind) {
        spec_d[4 * ns + 0] = spec_d[4 * ns + 1] = spec_d[4 * ns + 3] = 0; spec_d[4 * ns + 2] = sp.wctrl.get_duration_in_samples();
        spec_i[3 * ns + 0] = sp.wctrl.ctrl.ctrl.it.interp;
      }
      spec_vol[ns] = sp.vol;
    }
    int64_t k = 0;
    for (;;) {
      stepper.step(out, 1);
      ++k;
      bool busy = !e.ramp_specs.done();
      for (auto const &r : e.ramps) if (r.env.state != EnvelopeState::EnvelopeDone2) busy = true;
      if (!busy) break;
      if (wind && k >= 2000) { k = -1; break; }
      if (k >= max_steps) return -1;
    }
    if (np + static_cast<int64_t>(trace.size()) > played_cap) return -1;
    for (auto const &t : trace) { played[3 * np] = t.step; played[3 * np + 1] = t.ramp; played[3 * np + 2] = t.spec; ++np; }
    float const pan_gain = static_cast<float>(e.ramps[0].gain_.gains[0]);
    int32_t bits;
    std::memcpy(&bits, &pan_gain, 4);
    note_info[4 * i] = n_specs; note_info[4 * i + 1] = static_cast<int64_t>(trace.size()); note_info[4 * i + 2] = k; note_info[4 * i + 3] = bits;
    Event off{EventType::NoteOff, i + 1, 0.f, 0.f, 0};
    synth.onEvent(p->sample_rate, off, stepper);
    stepper.step(out, 1);
  }
  return 0;
}

// the restated MarkovChain alone: visits of each node over n_steps raw / normalised steps of the BIRDS (which = 0) or ROBOTS (1) graph from
// `start`, driven by the declared draws (tests: visit frequencies against the transition table)
void ora_markov_visits(int32_t which, int32_t start, int32_t normalized, uint64_t seed, int64_t n_steps, int64_t *visits4) {
  VoiceElement e;
  VoiceRng rng;
  rng.key = seed;
  rng.seed(1);
  e.rng = &rng;
  e.sample_rate = 48000;
  auto mc = which == 0 ? e.create_birds() : e.create_robot();
  mc->initialize(start);
  for (int i = 0; i < 4; ++i) visits4[i] = 0;
  for (int64_t i = 0; i < n_steps; ++i) {
    float const r = rng.uniform_real(0.f, 1.f);
    if (normalized) mc->step_normalized<false>(r); else mc->step<false>(r);
    ++visits4[mc->current];
  }
}

// itp::interpolate as decided in oracle_voice.hpp, over arrays (tests: end points, monotony, symmetry)
void ora_itp(int32_t kind, const double *t, int64_t n, double b, double c, double d, double *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = itp_interpolate(kind, t[i], b, c, d);
}


// ---- single objects, for the closed-form checks of tests/test_oracle_closed_forms_cpu.py ------------------------------------------------
// One ResynthElement (with_ctrl = 1: rt.resynth's; 0: driver (A)'s) stepped alone for n samples.  Before step k the operations with
// op_sample == k run in order: 0 setAngleIncrements(v), 1 setVolumeTarget(v), 2 the note-on preparations of ImplCRTP::onEvent and the
// initializer (tryAcquire, forgetPastSignals, setAHDSR, setMaxFilterIncrement(2 / stride), ctrl.setup(stride), stereo(pan = v)),
// 3 setStartAngle(v), 4 onKeyPressed(int(v)), 5 onKeyReleased(int(v)).  Per step: out [n][6] = {imag(0), envelope value, the oscillator's
// angle increment, the volume, the phasor's angle / pi in (-1, 1], RT-active after the step}.
void ora_elem_run(int32_t sample_rate, const int32_t *ahdsr4, float sustain, int32_t stride, int32_t with_ctrl, const int64_t *op_sample, const int32_t *op_type,
                  const double *op_value, int64_t n_ops, int64_t n, double *out) {
  ResynthElement e;
  e.with_ctrl = with_ctrl != 0;
  int64_t o = 0;
  for (int64_t k = 0; k < n; ++k) {
    for (; o < n_ops && op_sample[o] <= k; ++o) {
      double const v = op_value[o];
      switch (op_type[o]) {
        case 0: e.setAngleIncrements(v); break;
        case 1: e.setVolumeTarget(v); break;
        case 2:
          e.env.tryAcquire();
          e.forgetPastSignals();
          e.env.setAHDSR(AHDSR{ahdsr4[0], ahdsr4[1], ahdsr4[2], ahdsr4[3], sustain}, sample_rate);
          e.setMaxFilterIncrement(2. / static_cast<double>(stride));
          e.ctrl.setup(stride);
          e.setStereoGain(stereo(v));
          break;
        case 3: e.setStartAngle(v); break;
        case 4: e.env.onKeyPressed(static_cast<int32_t>(v)); break;
        default: e.env.onKeyReleased(static_cast<int32_t>(v)); break;
      }
    }
    e.step();
    double *r = out + 6 * k;
    r[0] = e.volume ? e.imag(0) : 0.0;
    r[1] = e.env.value();
    r[2] = e.osc.angleIncrements();
    r[3] = e.volume ? *e.volume : 0.0;
    r[4] = std::atan2(e.osc.cur.imag(), e.osc.cur.real()) / M_PI;
    r[5] = e.env.isEnvelopeRTActive() ? 1.0 : 0.0;
  }
}

// LogRamp<double> alone: the increments of n steps after setup(from, to, duration, start, interpolation)
void ora_logramp_run(double from, double to, double duration, double start, int32_t interp, int64_t n, double *out_inc) {
  LogRamp r;
  r.setup(from, to, duration, start, interp);
  for (int64_t k = 0; k < n; ++k) out_inc[k] = r.step();
}

// Limiter<double> as decided in oracle_dsp.hpp, in place over n stereo frames; target [n] receives getTargetCompressionLevel() per frame
void ora_limiter_run(double *lr, int64_t n, double *target) {
  Limiter lim;
  for (int64_t k = 0; k < n; ++k) { lim.feedOneFrame(lr + 2 * k); if (target) target[k] = lim.getTargetCompressionLevel(); }
}

// deduceNotesSlow as decided in oracle_resynth.hpp; returns the number of notes, stores the first cap
int64_t ora_deduce_notes(const float *x, int64_t n, int32_t sr, int32_t half_window_size, int32_t window_kind, int32_t stride, int32_t *start,
                         int32_t *end, double *freq, double *amp_db, int64_t cap) {
  std::vector<DeducedNote> notes;
  deduce_notes(x, n, sr, half_window_size, window_kind, stride, notes);
  int64_t total = static_cast<int64_t>(notes.size());
  for (int64_t i = 0; i < total && i < cap; ++i) { start[i] = notes[i].startFrame; end[i] = notes[i].endFrame; freq[i] = notes[i].frequency; amp_db[i] = notes[i].amplitude; }
  return total;
}

} // extern "C"
