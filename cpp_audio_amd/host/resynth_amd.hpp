// Host-side C++17 mirror of the reference interfaces on the rt.resynth hot path, over the C ABI of include/resynth_amd.h.
//
// The reference (OlivierSohn/cpp.audio) wires this path with templates and two type-erased callables;
// these shims have the same shapes, so the reference's two offline drivers compile against either backend:
//
//   resynth_amd::ComputeShim   <->  SimpleAudioOutContext::ComputeFunc = bool(double* buffer, int nFrames)
//                                   (include/simple_audio_context.h:24-26; registered like gen.crtp.h:258-264, vocoder.cpp:760-815)
//   resynth_amd::InputShim     <->  RecordF = void(const SAMPLE* in, int nFrames)  (include/audio_platforms.h:102-108, lib.cpp:1285-1291)
//   resynth_amd::RtResynthAmd  <->  RtResynth(RtResynthOfflineJob&)  (source/rt.resynth.lib.cpp:1185-1235), same job duck type
//                                   (get_samplerate / read_voice / read_carrier / has_more_* / count_outputs / write_output)
//   resynth_amd::resynth       <->  resynth<T>(notes, stride, sampling_rate, path)  (include/resynth.hpp:4-100)
//   resynth_amd::AmdElement    <->  the element duck type used by resynth.hpp's manual loop
//                                   (step / imag / setAngleIncrements / getAlgo().getVolumeAdjustment().setVolumeTarget /
//                                    editEnvelope().setAHDSR / getEnvelope().isEnvelopeRTActive / onKeyPressed / onKeyReleased /
//                                    forgetPastSignals, resynth.hpp:21,47-48,84-96)
//
// Error convention: RA_EINVAL -> std::invalid_argument, anything else -> std::runtime_error (the reference throws
// std::invalid_argument / std::runtime_error / std::logic_error, lib.cpp:308,337,449-481,1319-1354); computes return bool.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/resynth_amd.h"
#include "wav.hpp"

namespace resynth_amd {

inline int64_t check(int64_t code) {
  if (code >= 0) return code;
  std::string msg = ra_last_error();
  if (code == RA_EINVAL) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

inline ra_config default_config() {
  ra_config c;
  check(ra_default_config(&c));
  return c;
}

// RAII owner of n_streams RtResynth instances on one GPU
class Context {
public:
  Context(const ra_config &cfg, int n_streams, int device = 0) : n_streams_(n_streams) { check(ra_create(&cfg, n_streams, device, &h_)); }
  ~Context() { ra_destroy(h_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  ra_context *get() const { return h_; }
  int n_streams() const { return n_streams_; }
  // voice / carrier: [n_streams][n] (either may be null), out: [n_streams][n][2]
  void process(const float *voice, const float *carrier, int64_t n, float *out) { check(ra_process_host(h_, voice, carrier, n, out)); }
  void reset() { check(ra_reset(h_)); }
  ra_stats stats() const { ra_stats s; check(ra_get_stats(h_, &s)); return s; }

private:
  ra_context *h_ = nullptr;
  int n_streams_;
};

// ---- the two callables of the reference's audio thread --------------------------------------------------------------------
// Holds a raw pointer only (the reference rejects functors that own heap memory, simple_audio_context.h:95-97).
// `stream`: which of the context's streams this callable belongs to — one context serves the RecordF / ComputeFunc pairs of many
// RtResynth instances; per tile every stream is fed before the first compute runs (the order of the reference's audio callback).
struct InputShim {
  ra_context *ctx;
  const float *carrier = nullptr; // optional carrier block delivered together with the voice (job.read_carrier, lib.cpp:1412)
  int stream = 0;
  void operator()(const float *buf, int nFrames) const { check(ra_feed(ctx, stream, buf, carrier, nFrames)); }
};
struct ComputeShim {
  ra_context *ctx;
  int stream = 0;
  // accumulates nFrames (1..16) interleaved stereo doubles, like ImplCRTP::compute + the vocoder compute together
  bool operator()(double *buffer, int nFrames) const { return check(ra_compute(ctx, stream, buffer, nFrames)) != 0; }
};

// ---- RtResynth in offline-job mode ------------------------------------------------------------------------------------------
// Job duck type: rt.resynth.lib.params.cpp:249-386.
class RtResynthAmd {
public:
  template <typename Job>
  RtResynthAmd(Job &j, ra_config cfg, int device = 0, int64_t block_frames = 1 << 20) {
    if (j.count_outputs() != 2) throw std::logic_error("RtResynthAmd: the job must have 2 outputs (lib.cpp:1190)");
    cfg.sample_rate = j.get_samplerate();
    Context ctx(cfg, 1, device);
    std::vector<float> voice, carrier, out;
    // The reference loop (lib.cpp:1204-1234) is a do-while: it always runs one frame, then continues while either input has more.
    bool more = true;
    while (more) {
      voice.clear();
      carrier.clear();
      do {
        float v, c;
        j.read_voice(v);
        j.read_carrier(c);
        voice.push_back(v);
        carrier.push_back(c);
        more = j.has_more_voice() || j.has_more_carrier();
      } while (more && static_cast<int64_t>(voice.size()) < block_frames);
      out.resize(voice.size() * 2);
      ctx.process(voice.data(), carrier.data(), static_cast<int64_t>(voice.size()), out.data());
      j.write_output(out.data(), static_cast<int>(out.size()));
      frames_ += static_cast<int64_t>(voice.size());
    }
    stats_ = ctx.stats();
  }
  int64_t frames() const { return frames_; }
  int countDroppedNoteOns() const { return static_cast<int>(stats_.dropped_note_on); } // lib.cpp:549-551
  ra_stats stats() const { return stats_; }

private:
  int64_t frames_ = 0;
  ra_stats stats_{};
};

// ---- driver (A) -------------------------------------------------------------------------------------------------------------
// DeducedNote duck type: startFrame, endFrame, frequency, amplitude (main.resynth.cpp:39-46, resynth.hpp:58-61,84-96)
template <typename Note>
std::vector<double> render_notes(std::vector<Note> const &notes, int stride, int sampling_rate, int device = 0) {
  std::vector<ra_note> n(notes.size());
  for (size_t i = 0; i < notes.size(); ++i)
    n[i] = {static_cast<int32_t>(notes[i].startFrame), static_cast<int32_t>(notes[i].endFrame), static_cast<double>(notes[i].frequency),
            static_cast<double>(notes[i].amplitude)};
  int64_t T = check(ra_resynth_notes(n.data(), static_cast<int32_t>(n.size()), stride, sampling_rate, device, nullptr, 0));
  std::vector<double> out(static_cast<size_t>(T));
  check(ra_resynth_notes(n.data(), static_cast<int32_t>(n.size()), stride, sampling_rate, device, out.data(), T));
  return out;
}
// same signature as the reference's resynth(): writes a mono IEEE-float64 WAV (resynth.hpp:29-35)
template <typename Note>
void resynth(std::vector<Note> const &notes, int const stride, int const sampling_rate, const std::filesystem::path &path_wav) {
  std::vector<double> out = render_notes(notes, stride, sampling_rate);
  wav::write_f64(path_wav.string(), out.data(), out.size(), 1, sampling_rate);
}

// The body of testDeduceNotes (source/main.resynth.cpp:5-81) without its BMP plot: WAV -> deduceNotesSlow -> drop notes more than
// 60 dB below the loudest one or without a frequency (:53-71) -> resynth() -> mono float64 WAV.  Defaults as in the reference: a
// Gaussian half window of 400 points truncated at 4 sigma, window_center_stride 400 (:6,27).  The reference hands resynth() a literal
// 44100 (its file's rate, :79); here it is the input file's sample rate.  Returns the notes that were rendered.
struct DeducedNote { int32_t startFrame, endFrame; double frequency, amplitude; };
inline std::vector<DeducedNote> deduce_notes(const std::vector<float> &mono, int sample_rate, int half_window_size = 400, int window_center_stride = 400,
                                             int window_kind = RA_WINDOW_GAUSSIAN4, int device = 0) {
  int64_t n = check(ra_deduce_notes(mono.data(), static_cast<int64_t>(mono.size()), sample_rate, half_window_size, window_kind, window_center_stride, device,
                                    nullptr, 0));
  std::vector<ra_note> raw(static_cast<size_t>(n));
  if (n) check(ra_deduce_notes(mono.data(), static_cast<int64_t>(mono.size()), sample_rate, half_window_size, window_kind, window_center_stride, device, raw.data(), n));
  std::vector<DeducedNote> notes(raw.size());
  for (size_t i = 0; i < raw.size(); ++i) notes[i] = {raw[i].start_frame, raw[i].end_frame, raw[i].frequency, raw[i].amplitude_db};
  return notes;
}
inline std::vector<DeducedNote> resynth_wav(const std::filesystem::path &in_wav, const std::filesystem::path &out_wav, int half_window_size = 400,
                                            int window_center_stride = 400, int device = 0) {
  wav::Data d = wav::read(in_wav.string());
  if (d.samples.empty()) throw std::runtime_error("resynth_wav: no channel in '" + in_wav.string() + "'");
  std::vector<float> mono(d.samples[0].begin(), d.samples[0].end()); // deinterlaced.begin(): the first channel (main.resynth.cpp:39)
  std::vector<DeducedNote> notes = deduce_notes(mono, d.sample_rate, half_window_size, window_center_stride, RA_WINDOW_GAUSSIAN4, device);
  std::vector<DeducedNote> filtered;
  if (!notes.empty()) {
    constexpr double maxDbSpan = 60.;
    double max_mag = notes[0].amplitude;
    for (auto const &v : notes) max_mag = std::max(max_mag, v.amplitude);
    for (auto const &v : notes) {
      if (v.amplitude < max_mag - maxDbSpan) continue;
      if (v.frequency == 0) continue;
      filtered.push_back(v);
    }
  }
  resynth(filtered, window_center_stride, d.sample_rate, out_wav);
  return filtered;
}

// ---- element duck type ------------------------------------------------------------------------------------------------------
// Stands in for Enveloped<VolumeAdjusted<SineOscillatorAlgo<double>>, AHDSREnvelope<...>> in resynth.hpp's manual loop.
// The reference drives an element with a burst of setters followed by onKeyPressed(0) / onKeyReleased(delay) and then only
// step()/imag(); the first step() after the burst renders the note's whole life on the GPU, later steps read it back.
struct AHDSR { // audioelement.h:314-335 (aggregate-initialisable like the reference's: attack, itp, hold, decay, itp, release, itp, sustain)
  int32_t attack; int attackItp; int32_t hold; int32_t decay; int decayItp; int32_t release; int releaseItp; float sustain;
};
namespace itp { constexpr int LINEAR = 0; }

class AmdElement {
  struct Env {
    AmdElement *e;
    void setAHDSR(AHDSR const &a, int sample_rate) { e->ahdsr_ = {a.attack, a.hold, a.decay, a.release, a.sustain}; e->sample_rate_ = sample_rate; }
    bool isEnvelopeRTActive() const { return e->pressed_ && (e->pending_ || e->pos_ < e->buf_.size()); }
  };
  struct Vol {
    AmdElement *e;
    Vol &getVolumeAdjustment() { return *this; }
    void setVolumeTarget(double v) { e->volume_ = v; }
  };

public:
  explicit AmdElement(int device = 0) : device_(device) {}
  void forgetPastSignals() {}
  void setAngleIncrements(double a) { inc_ = a; }
  Vol getAlgo() { return Vol{this}; }
  Env editEnvelope() { return Env{this}; }
  Env getEnvelope() const { return Env{const_cast<AmdElement *>(this)}; }
  void onKeyPressed(int32_t delay) {
    if (delay != 0) throw std::invalid_argument("AmdElement: only onKeyPressed(0) is used on this path (resynth.hpp:95)");
    pressed_ = true; pending_ = true; release_delay_ = -1; buf_.clear(); pos_ = 0;
  }
  void onKeyReleased(int32_t delay) { release_delay_ = delay; }
  void step() {
    if (pending_) {
      if (release_delay_ < 1) throw std::logic_error("AmdElement: onKeyReleased(delay >= 1) must follow onKeyPressed (resynth.hpp:96)");
      int64_t T = check(ra_render_note(inc_, volume_, &ahdsr_, release_delay_, 0, sample_rate_, device_, phase_, nullptr, 0));
      buf_.resize(static_cast<size_t>(T));
      check(ra_render_note(inc_, volume_, &ahdsr_, release_delay_, 0, sample_rate_, device_, phase_, buf_.data(), T));
      pending_ = false;
      pos_ = 0;
      cur_ = 0.0;
    }
    if (pos_ < buf_.size()) cur_ = buf_[pos_++]; else cur_ = 0.0;
  }
  double imag() const { return cur_; }

private:
  int device_;
  double inc_ = 0, volume_ = 0, cur_ = 0;
  ra_ahdsr ahdsr_{1000, 0, 1000, 10000, 0.7f};
  int sample_rate_ = 44100;
  int32_t release_delay_ = -1;
  bool pressed_ = false, pending_ = false;
  double phase_[2] = {1.0, 0.0};
  std::vector<double> buf_;
  size_t pos_ = 0;
};

} // namespace resynth_amd
