/*
 * TEST INFRASTRUCTURE — CPU oracle, part 4: PeriodicFFT, vocoder, the RtResynth offline job loop
 * (call stack B of SURVEY.md §3) and the main.resynth driver (call stack A).
 * Restates source/rt.resynth.lib.periodicfft.cpp, source/rt.resynth.lib.vocoder.cpp,
 * source/rt.resynth.lib.cpp (:264-382, 1185-1302, 1572-1759) and include/resynth.hpp.
 */
#pragma once
#include "oracle_config.h"
#include "oracle_carrier.hpp"

namespace ora {

inline ora_config default_config() { // lib.cpp:943-999
  ora_config c{};
  c.sample_rate = 48000;
  c.window_size = 4096;
  c.stride = 1024;
  c.n_channels = 64;
  c.window_kind = ORA_WINDOW_GAUSSIAN4;
  c.min_volume = 0.0001f;
  c.nearby_distance_tones = 0.4f;
  c.max_track_pitches = 1.f;
  c.autotune_tolerance_pitches = 100.f;
  c.stereo_spread = 1.f;
  c.env_sustain_level = 1.f;
  c.use_autotune = 0;
  c.autotune_type = ORA_AUTOTUNE_SCALE;
  c.autotune_max_pitch = 150;
  c.autotune_factor = 2;
  c.autotune_scale_mode = ORA_SCALE_MAJOR;
  c.autotune_root_note = 0;
  c.autotune_root_transpose = 0;
  c.autotune_chord_frequencies = ORA_CHORD_HARMONICS;
  c.autotune_bit_chord = 0b10010001;
  c.vocoder_modulator_window_size_seconds = 0.10f;
  c.vocoder_stride_seconds = 0.005f;
  c.vocoder_count_bands = 5;
  c.vocoder_min_freq = 100.f;
  c.vocoder_max_freq = 20000.f;
  c.voice_volume = 0.f;
  c.carrier_volume = 0.1f;
  c.vocoder_volume = 0.f;
  c.analysis_volume = 0.f;
  c.analysis_input_gain = 1.f;
  c.analysis_output_feedback_gain = 0.f;
  c.output_delay_seconds = 1.f;
  c.phase_policy = ORA_PHASE_HASH;
  c.seed = 0;
  c.vocoder_kind = 0;
  c.vocoder_env_follower_cutoff_ratio = 1.f / 20.f;
  c.carrier_n_channels = 0;
  c.vocoder_carrier_square_volume = 1.f;
  c.vocoder_carrier_pulse_width = 0.01f;
  return c;
}

// lib.cpp:484-487 and :1577-1578
inline int even_window_size_frames(float window_size_seconds, int sample_rate) {
  return 2 * std::max(1, sample_rate ? static_cast<int>(0.5 * window_size_seconds * sample_rate) : 0);
}
inline int window_center_stride_frames(float stride_seconds, int sample_rate) {
  return std::max(1, static_cast<int>(0.5f + stride_seconds * sample_rate));
}

// ---- PeriodicFFT<Op> periodicfft.cpp:13-181 ---------------------------------------------------
// Op::operator()(const double*, int n) ; callback(stride, op)
template <typename Op>
struct PeriodicFFT {
  Op op;
  std::vector<double> samples;
  int window_size_ = 0, stride_ = 0, end = 0, ignore_frames = 0;
  std::function<void(int, Op const &)> on_fft_result;
  void setup(int window_size, int stride, std::function<void(int, Op const &)> cb) { // setLambdas :28-53 (sizes constant in a job)
    window_size_ = window_size;
    stride_ = stride;
    on_fft_result = std::move(cb);
    ignore_frames = 0;
    op.init_data(window_size_);
    samples.assign(window_size_, 0.0);
    end = 0;
  }
  void feed(float sample) { // :55-67
    if (ignore_frames > 0) { --ignore_frames; return; }
    samples[end] = sample;
    ++end;
    if (end == window_size_) { end = 0; onFullBuffer(); }
  }
  void onFullBuffer() { // :136-180
    op(samples.data(), window_size_);
    on_fft_result(stride_, op);
    int const overlap = window_size_ - stride_;
    if (overlap >= 0) {
      for (; end < overlap; ++end) samples[end] = samples[end + stride_];
      ignore_frames = 0;
    } else ignore_frames = -overlap;
  }
};

// SqMagFftOperation<Gaussian,double> periodicfft.cpp:251-325
struct SqMagOp {
  int window_kind = WINDOW_GAUSSIAN4;
  std::vector<double> half_window;
  SqMagFft fft;
  FrequenciesSqMag result;
  void init_data(int window_size) { if (static_cast<int>(half_window.size()) != window_size / 2) make_half_window(window_kind, window_size / 2, half_window); }
  void operator()(const double *x, int n) { fft(x, n, half_window, result); }
};

// FftOperation<double> periodicfft.cpp:183-244 (rectangular window, raw bins)
struct RawFftOp {
  RealFft fft;
  std::vector<double> sig;
  std::vector<std::complex<double>> bins; // L/2+1
  int fft_length = 0;
  void init_data(int) {}
  void operator()(const double *x, int n) {
    int L = get_fft_length_for(n, 1);
    if (fft.L != L) fft.init(L);
    fft_length = L;
    sig.assign(L, 0.0);
    std::copy(x, x + n, sig.begin());
    bins.resize(L / 2 + 1);
    fft.forward(sig.data(), bins.data());
  }
};

// ---- vocoder.cpp ------------------------------------------------------------------------------
inline int good_stride(float seconds, int sample_rate) { // :84-92
  int res = std::max(1, static_cast<int>(0.5f + seconds * sample_rate));
  if (0 == res % 2) ++res;
  return res;
}
inline void fill_freqs(float min_freq, float max_freq, int count_bands, std::vector<double> &freqs) { // :15-27
  freqs.clear();
  auto log_min = std::log(min_freq); // float overloads, as in the reference (float members)
  auto log_max = std::log(max_freq);
  for (int i = 0; i <= count_bands; ++i) {
    double const ratio = static_cast<double>(i) / count_bands;
    freqs.push_back(std::exp(log_min + ratio * (log_max - log_min)));
  }
}
inline int modulator_window_size(double window_size_seconds, int sample_rate) { // :111-119
  int c = std::max(1, static_cast<int>(0.5f + sample_rate * window_size_seconds));
  if (1 == c % 2) ++c;
  return std::min(1 << 16, c);
}

// band accumulation of FFTModulator's on_fft_result, :122-163; reused for modulate_bands (decision, Appendix B)
template <typename F>
inline void for_each_band_bin(std::vector<double> const &freqs, int n_bins, double bin_to_Hz, F f) {
  size_t lo = 0, hi = 1, end = freqs.size();
  int band = 0;
  for (int i = 0; i < n_bins; ++i) {
    double const Hz = bin_to_Hz * i;
    if (Hz <= freqs[lo]) continue;
    while (Hz > freqs[hi]) { ++band; ++hi; ++lo; if (hi == end) break; }
    if (hi == end) break;
    f(i, band);
  }
}

inline void BankVocoder::setup(ora_config const &c) { // Modulator::setup_bands / setup :315-354, Carrier::setup :669-692
  fill_freqs(c.vocoder_min_freq, c.vocoder_max_freq, c.vocoder_count_bands, freqs);
  int const sz = static_cast<int>(freqs.size()) - 1;
  mod_band_pass.assign(sz, BandPass{});
  car_band_pass.assign(sz, BandPass{});
  env_follower.assign(sz, EnvelopeFollower{});
  res.assign(sz, 0.0);
  for (int i = 0; i < sz; ++i) {
    mod_band_pass[i].setup(c.sample_rate, freqs[i], freqs[i + 1]);
    car_band_pass[i].setup(c.sample_rate, freqs[i], freqs[i + 1]);
    env_follower[i].setup(c.sample_rate, freqs[i] * c.vocoder_env_follower_cutoff_ratio);
  }
}

struct Vocoder { // Vocoder :734-856 = FFTModulator :100-252 + FFTCarrier :396-551
  int sample_rate = 0, S = 0, n_bands = 0;
  std::vector<double> freqs, bands_amplitudes;
  PeriodicFFT<SqMagOp> mod_fft;
  PeriodicFFT<RawFftOp> car_fft;
  std::vector<double> signal_old, signal_new;
  std::vector<std::complex<double>> tmp_bins;
  RealFft inv;
  int new_signal_idx = 0, sz_half_signal = 0;
  bool have_signal = false;
  int kind = 0;     // 1: Modulator<1, double> / Carrier<1, double> (the members commented out at :846-851) instead of the FFT pair
  BankVocoder bank;

  void setup(ora_config const &c) {
    kind = c.vocoder_kind;
    if (kind == 1) bank.setup(c);
    sample_rate = c.sample_rate;
    n_bands = c.vocoder_count_bands;
    fill_freqs(c.vocoder_min_freq, c.vocoder_max_freq, n_bands, freqs);
    bands_amplitudes.assign(n_bands, 0.0);
    // FFTModulator::setup stores the float seconds in doubles (:186-187)
    S = good_stride(static_cast<float>(static_cast<double>(c.vocoder_stride_seconds)), sample_rate);
    int Wm = modulator_window_size(static_cast<double>(c.vocoder_modulator_window_size_seconds), sample_rate);
    mod_fft.op.window_kind = WINDOW_GAUSSIAN4;
    mod_fft.setup(Wm, S, [this](int, SqMagOp const &op) {
      std::fill(bands_amplitudes.begin(), bands_amplitudes.end(), 0.0);
      auto const &sq = op.result;
      for_each_band_bin(freqs, static_cast<int>(sq.frequencies_sqmag.size()), sq.bin_index_to_Hz(sample_rate),
                        [&](int i, int band) { bands_amplitudes[band] += sq.frequencies_sqmag[i]; });
      for (auto &a : bands_amplitudes) a = std::sqrt(a);
    });
    int Lc = get_fft_length_for(2 * S, 1);
    signal_old.assign(Lc, 0.0);
    signal_new.assign(Lc, 0.0);
    car_fft.setup(2 * S, S, [this](int stride, RawFftOp const &op) { // :420-475
      sz_half_signal = stride;
      int const L = op.fft_length;
      signal_old.swap(signal_new);
      new_signal_idx = 0;
      tmp_bins.assign(op.bins.begin(), op.bins.end());
      // RealFBins_::modulate_bands(sr, amplitudes, freqs, bins): scale each bin by its band's amplitude; bins outside every band -> 0
      std::vector<double> gain(tmp_bins.size(), 0.0);
      for_each_band_bin(freqs, static_cast<int>(tmp_bins.size()), static_cast<double>(sample_rate) / L,
                        [&](int i, int band) { gain[i] = bands_amplitudes[band]; });
      for (size_t i = 0; i < tmp_bins.size(); ++i) tmp_bins[i] *= gain[i];
      if (inv.L != L) inv.init(L);
      inv.inverse(tmp_bins.data(), signal_new.data());
      double const scale = 1.0 / L;
      for (auto &v : signal_new) v *= scale;
      have_signal = true;
    });
    new_signal_idx = 0;
    have_signal = false;
  }
  double feed_carrier(double sample) { // FFTCarrier::feed :501-530
    car_fft.feed(static_cast<float>(sample)); // PeriodicFFT::feed takes a float (:55)
    if (!have_signal) return 0.0;             // reference reads zero-initialised buffers here
    double nm, om;
    equal_gain_xfade(sz_half_signal - 1, new_signal_idx + 1, nm, om);
    double res = signal_new[new_signal_idx] * nm + signal_old[new_signal_idx + sz_half_signal] * om;
    ++new_signal_idx;
    return res;
  }
  // the compute lambda of Vocoder::initialize, :763-812, one frame
  void compute_one(double modulator, double carrier, float vol_mod, float vol_car, float vol_voc, double *buf2) {
    double vocoded;
    if (kind == 1) vocoded = bank.feed(modulator, carrier);
    else {
      mod_fft.feed(static_cast<float>(modulator));
      vocoded = feed_carrier(carrier);
    }
    double const mixed = vol_mod * modulator + vol_car * carrier + vol_voc * vocoded;
    buf2[0] += mixed;
    buf2[1] += mixed;
  }
};

// ---- RtResynth, offline job (lib.cpp:1185-1235) -----------------------------------------------
struct RtResynthOffline {
  ora_config cfg;
  Midi midi;
  Stepper stepper;
  Synth synth;
  Vocoder vocoder;
  Limiter limiter;              // lib.cpp:908
  CarrierSynth vocoder_carrier; // lib.cpp:910
  NoteIdsGenerator vocoder_carrier_noteids;
  std::vector<ora_carrier_event> carrier_events; // sorted by sample; consumed by run()
  size_t carrier_event_pos = 0;
  int carrier_dropped_note_on = 0;
  std::vector<double> *carrier_dump = nullptr; // per frame, the carrier sample the vocoder was given
  bool carrier_from_synth = false;
  Autotune autotune;
  PeriodicFFT<SqMagOp> periodic_fft;
  std::vector<FreqMag> freqmags;
  std::vector<PitchVolume> freqmags_data, reduced_pitches, autotuned_pitches, pitches_tmp;
  std::vector<PitchInterval> pitch_intervals;
  std::vector<float> loudness_keys;
  std::vector<int> order, pitch_changes;
  std::vector<char> continue_playing;
  std::vector<PlayedNote> played_pitches;
  int64_t analysis_frame_idx = 0, next_noteid = 0;
  int dropped_note_on = 0;
  std::vector<EventLogEntry> event_log;
  int64_t now = 0; // index of the output frame the next stepper.step produces
  // per-frame dump of the stateless half of the pipeline (for kernel-level parity tests)
  struct FrameDump { int64_t fire_sample; std::vector<PitchVolume> pitches; };
  std::vector<FrameDump> *frame_dump = nullptr;
  // the current modulator / carrier samples, handed to the vocoder compute (lib.cpp:1397-1418)
  double cur_voice = 0, cur_carrier = 0;
  bool enable_vocoder = true;
  // output -> analysis feedback (lib.cpp:1626-1652): the last published mono sample and the cyclic delay line (L0 `cyclic<float>`: a ring
  // whose cycleEnd() is the oldest element and whose feed() overwrites it)
  float last_published = 0.f;
  std::vector<float> output_delay;
  int delay_pos = 0;

  explicit RtResynthOffline(ora_config const &c) : cfg(c), synth(c.n_channels), vocoder_carrier(std::max(0, c.carrier_n_channels)) {
    synth.phase_policy = c.phase_policy;
    synth.seed = c.seed;
    vocoder_carrier.phase_policy = c.phase_policy;
    vocoder_carrier.seed = c.seed;
    vocoder_carrier.now = &now;
    synth.log = &event_log;
    synth.now = &now;
    autotune.build(c.use_autotune, c.autotune_type, c.autotune_scale_mode, c.autotune_root_note, c.autotune_root_transpose,
                   c.autotune_chord_frequencies, c.autotune_bit_chord, c.autotune_factor);
    // init(): synth.initialize (lib.cpp:1353), then vocoder.initialize (:1377), then init_analysis (:1432)
    synth.initialize(stepper);
    vocoder.setup(c);
    stepper.enqueueOneShot([this]() {
      stepper.registerSimpleCompute([this](double *buf, int frames) {
        if (!enable_vocoder) return true;
        for (int i = 0; i < frames; ++i) {
          // get_modulator_carrier_sample, lib.cpp:1395-1418: realtime mode computes one frame of the carrier synth, the offline job reads
          // its carrier file; here: the file when the job has one, else the synth when it has channels
          double carrier_val = cur_carrier;
          if (carrier_from_synth) { carrier_val = 0.0; vocoder_carrier.compute(&carrier_val, 1); }
          if (carrier_dump) carrier_dump->push_back(carrier_val);
          vocoder.compute_one(cur_voice, carrier_val, cfg.voice_volume, cfg.carrier_volume, cfg.vocoder_volume, buf + 2 * i);
        }
        return true;
      });
    });
    { // updateVocoderCarrierInitializer, lib.cpp:1498-1516 (no note is playing yet: only future notes are concerned)
      CarrierInitializer init;
      init.sample_rate = c.sample_rate;
      init.ahdsr = mkAHDSR(c.sample_rate, c.env_attack_seconds, c.env_hold_seconds, c.env_decay_seconds, c.env_release_seconds, c.env_sustain_level);
      init.noise_volume = c.vocoder_carrier_noise_volume; init.saw_volume = c.vocoder_carrier_saw_volume;
      init.triangle_volume = c.vocoder_carrier_triangle_volume; init.square_volume = c.vocoder_carrier_square_volume;
      init.sine_volume = c.vocoder_carrier_sine_volume; init.pulse_volume = c.vocoder_carrier_pulse_volume;
      init.pulse_width = c.vocoder_carrier_pulse_width;
      vocoder_carrier.initializer = init;
    }
    stepper.post = [this](double *buf, int frames) { // init_post, lib.cpp:1245-1275: the limiter (post "limit"), then the mono publication
      if (cfg.post == 1) for (int i = 0; i < frames; ++i) limiter.feedOneFrame(buf + 2 * i);
      for (int i = 0; i < frames; ++i) last_published = static_cast<float>(buf[2 * i] + buf[2 * i + 1]);
    };
    periodic_fft.op.window_kind = c.window_kind;
    periodic_fft.setup(c.window_size, c.stride, [this](int stride, SqMagOp const &op) {
      extract_local_max_freqs_mags(cfg.sample_rate, op.result, freqmags); // lib.cpp:1592-1595
      step(freqmags, stride);                                             // lib.cpp:1607-1613
    });
  }

  // lib.cpp:1670-1759
  void step(std::vector<FreqMag> const &fs, int future_stride) {
    ++analysis_frame_idx;
    frequencies_to_pitches(midi, fs, freqmags_data);
    aggregate_pitches(cfg.nearby_distance_tones, freqmags_data, pitch_intervals);
    reduce_pitches(cfg.min_volume, pitch_intervals, reduced_pitches);
    shift_pitches(cfg.pitch_shift_pre_autotune, reduced_pitches);
    harmonize_pitches(cfg.pitch_harmonize_pre_autotune, pitches_tmp, reduced_pitches);
    autotune_pitches(static_cast<double>(cfg.autotune_max_pitch), cfg.autotune_tolerance_pitches, autotune, reduced_pitches, autotuned_pitches);
    shift_pitches(cfg.pitch_shift_post_autotune, autotuned_pitches);
    harmonize_pitches(cfg.pitch_harmonize_post_autotune, pitches_tmp, autotuned_pitches);
    if (frame_dump) frame_dump->push_back({now, autotuned_pitches});
    track_pitches(cfg.max_track_pitches, autotuned_pitches, played_pitches, pitch_changes, continue_playing);
    order_pitches_by_perceived_loudness(autotuned_pitches, loudness_keys, order);
    synthesize_sounds(future_stride);
    remove_dead_notes(continue_playing, played_pitches);
    sort_by_current_pitch(played_pitches);
  }

  // lib.cpp:264-382
  void synthesize_sounds(int future_stride) {
    {
      ElementInitializer init{cfg.sample_rate, future_stride, cfg.stereo_spread,
                              mkAHDSR(cfg.sample_rate, cfg.env_attack_seconds, cfg.env_hold_seconds, cfg.env_decay_seconds,
                                      cfg.env_release_seconds, cfg.env_sustain_level),
                              cfg.seed};
      if (!synth.initializer || !(*synth.initializer == init)) { // apply_initializer_to_current_and_future_notes :244-262
        synth.initializer = init;
        stepper.enqueueOneShot([this, init]() {
          for (auto &c : synth.channels)
            if (c.elem.env.isEnvelopeRTActive()) init(c.elem, false, 0);
        });
      }
    }
    for (size_t idx = 0; idx < continue_playing.size(); ++idx) {
      if (!continue_playing[idx]) synth.onEvent(cfg.sample_rate, Event{EventType::NoteOff, played_pitches[idx].noteid, 0.f, 0.f}, stepper);
    }
    for (int idx : order) {
      int const pitch_change = pitch_changes[idx];
      double const new_pitch = autotuned_pitches[idx].midipitch;
      float const new_freq = static_cast<float>(midi.midi_pitch_to_freq_event(new_pitch));
      float const volume = static_cast<float>(cfg.analysis_volume * autotuned_pitches[idx].volume / reduceUnadjustedVolumes);
      if (pitch_change >= 0) {
        PlayedNote &played = played_pitches[pitch_change];
        synth.onEvent(cfg.sample_rate, Event{EventType::NoteChange, played.noteid, new_freq, volume}, stepper);
        played.cur_freq = new_freq;
        played.midi_pitch = new_pitch;
        played.cur_velocity = volume;
      } else {
        if (volume <= 0) continue;
        ++next_noteid;
        auto res = synth.onEvent(cfg.sample_rate, Event{EventType::NoteOn, next_noteid, new_freq, volume}, stepper);
        if (res == onEventResult::OK) played_pitches.push_back({analysis_frame_idx, next_noteid, new_pitch, new_freq, volume});
        else ++dropped_note_on;
      }
    }
  }

  // The do-while of lib.cpp:1204-1234 over n_frames input frames (= max(len voice, len carrier)).
  // Analysis alignment (pubsub.cpp:151-200): the aggregator's start-up consumes input
  // samples 0 and 1, so PeriodicFFT receives inputs 2,3,... ; analysis of iteration k runs before
  // the output step of iteration k.  analysis_output_feedback_gain is 0 (default): no feedback term.
  void run(const float *voice, const float *carrier, int64_t n_frames, float *out /* [n_frames][2] */) {
    carrier_from_synth = !carrier && cfg.carrier_n_channels > 0;
    for (int64_t k = 0; k < n_frames; ++k) {
      now = k;
      // RtResynth::onInputMidiEvent, lib.cpp:1518-1545: events stamped with frame k reach the synth before the step of frame k
      for (; carrier_event_pos < carrier_events.size() && carrier_events[carrier_event_pos].sample <= k; ++carrier_event_pos) {
        ora_carrier_event const &ce = carrier_events[carrier_event_pos];
        if (ce.type == 0) {
          float const f = ce.frequency > 0.f ? ce.frequency : static_cast<float>(midi.midi_pitch_to_freq(static_cast<double>(ce.key)));
          Event e{EventType::NoteOn, vocoder_carrier_noteids.NoteOnId(static_cast<uint64_t>(ce.key)), f, ce.velocity, ce.delay};
          if (vocoder_carrier.onEvent(cfg.sample_rate, e, stepper) != onEventResult::OK) ++carrier_dropped_note_on;
        } else {
          auto id = vocoder_carrier_noteids.NoteOffId(static_cast<uint64_t>(ce.key));
          if (id) vocoder_carrier.onEvent(cfg.sample_rate, Event{EventType::NoteOff, *id, 0.f, 0.f, ce.delay}, stepper);
        }
      }
      float const v = voice ? voice[k] : 0.f;
      cur_voice = v; // read_direct_input: the float sample widened to double (lib.cpp:1073-1079)
      cur_carrier = carrier ? static_cast<double>(carrier[k]) : 0.0;
      if (k >= 2) {
        // analyze_until_input_starvation, lib.cpp:1622-1652.  The aggregator's start-up synchronisation (pubsub.cpp:151-200) swallows
        // input samples 0, 1 and output sample 0; from then on it pairs input sample k with output sample k - 1 (published by the post
        // processor of the previous step as float(L + R), lib.cpp:1262-1272, pubsub.cpp:92).
        float sum = 0.f;
        sum += cfg.analysis_input_gain * v; // :1636
        {
          int const sz = std::max(0, static_cast<int>(0.5f + cfg.output_delay_seconds * cfg.sample_rate)); // :1638-1639
          if (static_cast<int>(output_delay.size()) != sz) { output_delay.assign(sz, 0.f); delay_pos = 0; } // resize zeroes, :1640-1642
          float const o = last_published;
          if (sz) {
            float const value = output_delay[delay_pos]; // *cycleEnd(): the oldest element
            sum += cfg.analysis_output_feedback_gain * value;
            output_delay[delay_pos] = o;                 // feed()
            delay_pos = (delay_pos + 1) % sz;
          } else sum += cfg.analysis_output_feedback_gain * o;
        }
        periodic_fft.feed(sum);
      }
      stepper.step(out + 2 * k, 1);
    }
  }
};

// ---- driver (A): include/resynth.hpp:4-100 -----------------------------------------------------
struct DeducedNote { int32_t startFrame, endFrame; double frequency, amplitude; };

inline void resynth_notes(std::vector<DeducedNote> const &notes, int stride, int sampling_rate, std::vector<double> &out) {
  out.clear();
  std::vector<std::unique_ptr<ResynthElement>> pool;
  auto getAvailableOscillator = [&pool]() {
    for (auto &p : pool)
      if (!p->env.isEnvelopeRTActive()) return p.get();
    pool.push_back(std::make_unique<ResynthElement>());
    pool.back()->with_ctrl = false;
    return pool.back().get();
  };
  auto recordFrame = [&pool, &out]() {
    double res = 0.;
    bool empty = true;
    for (auto const &p : pool) {
      if (!p->env.isEnvelopeRTActive()) continue;
      empty = false;
      p->step();
      res += p->imag_mono();
    }
    out.push_back(res);
    return empty;
  };
  std::vector<DeducedNote> byStart = notes;
  for (auto &n : byStart) { n.startFrame *= stride; n.endFrame *= stride; }
  // std::sort in the reference (:65-70): order of equal start frames is implementation-defined; stable here.
  std::stable_sort(byStart.begin(), byStart.end(), [](DeducedNote const &n, DeducedNote const &m) { return n.startFrame < m.startFrame; });
  int frame = 0;
  for (auto const &note : byStart) {
    for (; frame < note.startFrame; ++frame) recordFrame();
    ResynthElement *ae = getAvailableOscillator();
    ae->forgetPastSignals();
    ae->setAngleIncrements(2 * note.frequency / sampling_rate); // freq_to_angle_increment<double>
    ae->setVolumeTarget(db_to_mag(note.amplitude));
    ae->env.setAHDSR(AHDSR{1000, 0, 1000, 10000, 0.7f}, sampling_rate);
    ae->env.onKeyPressed(0);
    ae->env.onKeyReleased(stride + note.endFrame - note.startFrame);
  }
  while (!recordFrame()) {}
}

// ---- deduceNotesSlow as main.resynth calls it (main.resynth.cpp:39-46) -------------------------------------------------------
// cpp.algorithms code, absent: PARITY UNPINNED.  Certain from the call site: a half window (mirrored into a full one), a hop
// (`window_center_stride`), zero padding factor 1, a "half tone apart in log2 frequency" constant, and the result type
// DeducedNote{startFrame, endFrame (hop units, resynth.hpp:58-61), frequency (Hz; 0 is possible: filtered at main.resynth.cpp:66-68),
// amplitude (dB: :55-65, resynth.hpp:86)}.  Decision (the same as the product's, include/resynth_amd.h: ra_deduce_notes): the rt.resynth
// analysis per frame, then track_pitches' scan links peaks within one semitone into notes; frequency = volume-weighted mean pitch in
// Hz, amplitude = the largest dB, frames = first / last sighting.
inline void deduce_notes(const float *x, int64_t n, int sample_rate, int half_window_size, int window_kind, int stride, std::vector<DeducedNote> &out) {
  out.clear();
  Midi midi;
  SqMagOp op;
  op.window_kind = window_kind;
  int const N = 2 * half_window_size;
  op.init_data(N);
  std::vector<double> win(N);
  std::vector<FreqMag> peaks;
  struct Open { int32_t start, last; double last_pitch, sum_v, sum_pv, max_db; };
  std::vector<Open> open, next;
  auto close = [&](Open const &o) { out.push_back({o.start, o.last, 440.0 * std::pow(2.0, (o.sum_pv / o.sum_v - 69.0) / 12.0), o.max_db}); };
  for (int32_t h = 0; static_cast<int64_t>(h) * stride + N <= n; ++h) {
    for (int i = 0; i < N; ++i) win[i] = static_cast<double>(x[static_cast<int64_t>(h) * stride + i]);
    op(win.data(), N);
    extract_local_max_freqs_mags(sample_rate, op.result, peaks);
    next.clear();
    std::vector<char> cont(open.size(), 0);
    size_t it = 0;
    for (auto const &pk : peaks) {
      auto pitch = midi.frequency_to_midi_pitch_analysis(pk.freq);
      if (!pitch) continue;
      double const v = db_to_mag_analysis(pk.mag_db);
      if (!(v > 0.0)) continue;
      int match = -1;
      for (; it != open.size(); ++it) { // track_pitches, algo.cpp:256-305, with a threshold of one semitone
        if (open[it].last_pitch < *pitch - 1.0) continue;
        if (open[it].last_pitch <= *pitch + 1.0) { match = static_cast<int>(it); ++it; }
        break;
      }
      if (match >= 0) {
        Open o = open[match];
        cont[match] = 1;
        o.last = h; o.last_pitch = *pitch; o.sum_v += v; o.sum_pv += *pitch * v; o.max_db = std::max(o.max_db, pk.mag_db);
        next.push_back(o);
      } else next.push_back({h, h, *pitch, v, *pitch * v, pk.mag_db});
    }
    for (size_t j = 0; j < open.size(); ++j) if (!cont[j]) close(open[j]);
    std::stable_sort(next.begin(), next.end(), [](Open const &a, Open const &b) { return a.last_pitch < b.last_pitch; });
    open.swap(next);
  }
  for (auto const &o : open) close(o);
}

} // namespace ora
