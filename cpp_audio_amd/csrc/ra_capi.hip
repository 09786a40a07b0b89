// Host side of libresynth_amd.so: context, constant tables, block scheduling, the C ABI of include/resynth_amd.h.
// There is no CPU fallback: every compute entry point needs a HIP device and fails with RA_ENODEVICE otherwise.
#include "ra_internal.h"
#include "ra_voice_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

static thread_local std::string g_last_error;
thread_local RaExtTiming *ra_ext_timing = nullptr;
#define RA_TRY(expr) do { if (int rc_ = (expr)) return rc_; } while (0)
static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }

#define HIP_TRY(expr)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) return fail(RA_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
  } while (0)

struct ra_context {
  ra_config cfg;
  RaParams p;
  RaBuffers b;
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t t = 0;              // frames processed so far (per stream)
  int64_t frames_fired = 0;   // analysis frames fired so far (per stream)
  int64_t last_f_lo = 0, last_nfb = 0, last_n = 0, last_t0 = 0;
  std::vector<void *> allocs;
  // staging for the host-pointer entry points
  float *st_voice = nullptr, *st_carrier = nullptr, *st_out = nullptr;
  double *st_out_d = nullptr;
  int64_t st_cap = 0;
  // ---- the ComputeFunc shim (ra_feed / ra_compute): per stream the input fed but not yet rolled into the window history, and the output
  // rendered but not yet served.  A synth-only context renders AHEAD: between two analysis fires nothing can change the output (events
  // apply where their frame fires, simple_audio_context.h:74-76), so when a compute reaches a fire sample the whole hop up to the next one is
  // rendered by one launch sequence for all streams and the following computes are served from the ring.  The hop's input is still being
  // fed while it is served: its window-history roll is deferred until the hop's input is complete.
  struct Shim {
    std::vector<std::vector<float>> in_voice, in_carrier; // [stream]: samples from frame in_base on
    std::vector<char> carrier_given;                      // [stream]: ra_feed was given a carrier since in_base
    std::vector<std::vector<double>> ring;                // [stream]: interleaved output frames; those from ring_head[stream] on are frames served[stream] .. rendered
    std::vector<size_t> ring_head;                        // [stream]: doubles of ring[stream] already handed out
    int32_t *h_flags = nullptr;                           // pinned: the device error flags, fetched with each block's output
    std::vector<int64_t> served;                          // [stream]: frames handed out by ra_compute
    int64_t in_base = 0, rendered = 0;
    bool deferred = false;                                // the block [def_t0, def_t0 + def_n) has been rendered; its history roll is still owed
    int64_t def_t0 = 0, def_n = 0;
    double *h_out = nullptr; int64_t h_out_cap = 0;       // pinned
    float *h_in = nullptr; int64_t h_in_cap = 0;          // pinned
    bool active = false;
  } shim;
  double *dbg_sqmag = nullptr;
  unsigned long long *dbg_clk = nullptr;
  unsigned long long *dbg_stamp = nullptr; // ra_set_clock_stamps
  // optional per-kernel timing with HIP events on the launch stream (ra_set_profiling)
  bool profiling = false;
  struct Timed { int stage; hipEvent_t a, b; };
  std::vector<Timed> timed;          // launches of the calls not yet resolved
  std::vector<std::pair<hipEvent_t, hipEvent_t>> spare; // recorded event pairs whose launch was timed by hipExtLaunchKernelGGL instead: back to the pool once resolved
  std::vector<hipEvent_t> ev_pool;   // recycled timing events
  double kernel_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // analysis, track, vocoder, propagate, render, history, tracking pre-pass (ST_*)
  int64_t kernel_calls = 0;
  int64_t kernel_launches[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // chunk pipeline: analysis / tracking / propagation / render streams and the events chaining them
  bool pipe_ready = false;
  hipStream_t pipe_stream[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // analysis, tracking, propagation, render, vocoder, carrier synth
  hipEvent_t ev_fork = nullptr, ev_voc = nullptr, ev_car = nullptr, ev_join[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_chunk[3][12] = {};
  // Per-block scratch (pitch lists, commands, hop states) exists twice: block k + 1 may run its analysis / tracking / propagation while
  // block k is still rendering (ra_process_device_async).  A set is reused two blocks later; the events say when its last readers are done.
  struct Scratch {
    int64_t frames_cap = 0, seg_cap = 0;
    double *pl_pitch = nullptr, *pl_vol = nullptr; float *pl_key = nullptr; int32_t *pl_count = nullptr;
    float *pl_nf = nullptr, *pl_nv = nullptr; int32_t *pl_nmc = nullptr, *pl_ord = nullptr;
    RaCmd *cmd = nullptr; int32_t *cmd_non = nullptr;
    double *seg_sd = nullptr; int32_t *seg_si = nullptr;
    double *voc_mix = nullptr; int64_t mix_cap = 0; // [n_streams][n] the vocoder's mix of the block (added by the render)
    double *car_gen_d = nullptr; float *car_gen_f = nullptr; int64_t car_gen_cap = 0; // [n_streams][n] the carrier synth's output for the block (read by the vocoder)
    hipEvent_t ev_track_done = nullptr, ev_prop_done = nullptr, ev_render_done = nullptr, ev_voc_done = nullptr;
    bool used = false;
  } scr[2];
  int parity = 0;
  // vocoder carrier synth: note events not yet consumed by a block, per stream, and the key -> note id book of the reference's
  // NoteIdsGenerator (include/events.h:228-281; one per RtResynth, lib.cpp:911)
  struct CarNoteIds { int64_t next = 0; std::unordered_map<uint64_t, std::deque<int64_t>> ids; };
  std::vector<std::deque<RaCarEv>> car_pending;
  std::vector<CarNoteIds> car_noteids;
  std::vector<int64_t> car_last_sample;
  // host images of the blocks' event lists, in pinned memory: per scratch set one buffer [events][n_streams + 1 offsets] and an event that says
  // when its copies have drained (the host waits for it before refilling the set two blocks later; pageable vectors made the "async" copies
  // host-synchronous on the vocoder queue)
  struct CarStage { char *h = nullptr; size_t cap = 0; size_t n_ev = 0; hipEvent_t done = nullptr; bool used = false; } car_stage[2];
  int64_t car_total_segs = 0;           // slots of the block's segment table (stage_carrier_events)
  bool pipeline_serial = false;         // measurement aid (ra_set_pipeline_serial): same chunks and launches, all on the caller's stream
  bool last_car_gen = false;            // the last block's carrier came from the synth (ra_get_carrier)
  // a block failed after it had started to change stream state (launches issued, histories rolled): the streams are then neither at the
  // block's start nor at its end, and every entry point returns RA_EINTERNAL until ra_reset
  bool poisoned = false;
  bool pending_join = false; // asynchronous blocks are in flight on the internal streams; the caller's stream has not been made to wait yet
  // ra_process_host_pipelined: two copy queues, two sets of chunk-sized staging buffers and the events that order copy-in -> block -> copy-out
  struct HostPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    float *voice[2] = {nullptr, nullptr}, *carrier[2] = {nullptr, nullptr}, *out[2] = {nullptr, nullptr};
    int64_t cap = 0;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out_done[2] = {nullptr, nullptr}, ev_user[2] = {nullptr, nullptr}, ev_pipe[2][6] = {};
  } hp;
};

namespace {

template <typename T>
int dev_alloc(ra_context *c, T **ptr, size_t count, bool zero = true) {
  *ptr = nullptr;
  if (count == 0) count = 1;
  void *p = nullptr;
  HIP_TRY(hipMalloc(&p, count * sizeof(T)));
  if (zero) HIP_TRY(hipMemsetAsync(p, 0, count * sizeof(T), c->stream));
  c->allocs.push_back(p);
  *ptr = static_cast<T *>(p);
  return RA_OK;
}
void dev_free(ra_context *c, void *p) {
  if (!p) return;
  auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
  if (it != c->allocs.end()) c->allocs.erase(it);
  (void)hipFree(p);
}
template <typename T>
int upload(ra_context *c, T **ptr, const std::vector<T> &v) {
  int r = dev_alloc(c, ptr, v.size(), false);
  if (r) return r;
  if (!v.empty()) HIP_TRY(hipMemcpy(*ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return RA_OK;
}

constexpr int kMaxChunks = 12;
int fft_length_for(int n) { int l = 1; while (l < n) l <<= 1; return l; }

// ---- host-side constants (own restatement of the reference's small pure functions; not the oracle) --------------
constexpr int kNumHalftones = 12;
double freq_to_pitch(double f) { return 69.0 + (kNumHalftones / 1.0) * std::log2(f / 440.f); } // midi.h:223-232

const float kLoudFreqs[29] = {20.f, 25.f, 31.5f, 40.f, 50.f, 63.f, 80.f, 100.f, 125.f, 160.f, 200.f, 250.f, 315.f, 400.f, 500.f,
                              630.f, 800.f, 1000.f, 1250.f, 1600.f, 2000.f, 2500.f, 3150.f, 4000.f, 5000.f, 6300.f, 8000.f, 10000.f, 12500.f};
const float kAlpha[29] = {.532f, .506f, .480f, .455f, .432f, .409f, .387f, .367f, .349f, .330f, .315f, .301f, .288f, .276f, .267f,
                          .259f, .253f, .250f, .246f, .244f, .243f, .243f, .243f, .242f, .242f, .245f, .254f, .271f, .301f};
const float kLu[29] = {-31.6f, -27.2f, -23.0f, -19.1f, -15.9f, -13.0f, -10.3f, -8.1f, -6.2f, -4.5f, -3.1f, -2.0f, -1.1f, -0.4f, 0.f,
                       .3f, .5f, 0.f, -2.7f, -4.1f, -1.0f, 1.7f, 2.5f, 1.2f, -2.1f, -7.1f, -11.2f, -10.7f, -3.1f};
const float kTf[29] = {78.5f, 68.7f, 59.5f, 51.1f, 44.0f, 37.5f, 31.5f, 26.5f, 22.1f, 17.9f, 14.4f, 11.4f, 8.6f, 6.2f, 4.4f,
                       3.0f, 2.2f, 2.4f, 3.5f, 1.7f, -1.3f, -4.2f, -6.0f, -5.4f, -1.5f, 6.0f, 12.6f, 13.9f, 12.3f};
// include/loudness.h:188-197
float equal_loudness_volume(int i, float LN) {
  float Af = 4.47e-3f * (std::pow(10.f, .025f * LN) - 1.14f) + std::pow(.4f * std::pow(10.f, ((kTf[i] + kLu[i]) * .1f) - 9.f), kAlpha[i]);
  return 94.f - kLu[i] + (10.f / kAlpha[i]) * std::log(Af) / std::log(10.f);
}

int ms_to_frames(float ms, int sr) { return static_cast<int>(0.5f + (static_cast<float>(sr) / 1000.f) * ms); } // sound.functions.h:29-35

// mkAutotuneFunction tables, lib.cpp:1761-1873
void build_autotune(const ra_config &c, RaParams &p, std::vector<double> &allowed) {
  allowed.clear();
  p.use_autotune = c.use_autotune ? 1 : 0;
  p.autotune_type = c.autotune_type;
  p.autotune_max_pitch = static_cast<double>(c.autotune_max_pitch);
  static const double scales[3][8] = {{0., 2., 4., 5., 7., 9., 11., 12.}, {0., 2., 3., 5., 7., 8., 10., 12.}, {0., 2., 3., 5., 7., 8., 11., 12.}};
  int mode = std::min(std::max(c.autotune_scale_mode, 0), 2);
  for (int i = 0; i < 8; ++i) p.scale_pitches[i] = scales[mode][i];
  for (int i = 0; i < 7; ++i) p.scale_equi[i] = 0.5 * (scales[mode][i] + scales[mode][i + 1]);
  const int Do = 0, La = 9;
  p.scale_root = 69.0 + c.autotune_root_transpose + (c.autotune_root_note - La);
  if (!c.use_autotune) return;
  if (c.autotune_type == RA_AUTOTUNE_CHORD) {
    int offset = c.autotune_root_note - Do;
    if (offset < 0) offset += kNumHalftones;
    offset += c.autotune_root_transpose;
    double const C_pitch = 69.0 + (Do - La) + kNumHalftones;
    double const root = C_pitch + offset;
    if (c.autotune_chord_frequencies == RA_CHORD_SINGLE || c.autotune_chord_frequencies == RA_CHORD_OCTAVES) {
      bool single = c.autotune_chord_frequencies == RA_CHORD_SINGLE;
      for (int octave = single ? 0 : -5; octave <= (single ? 0 : 5); ++octave)
        for (int i = 0; i < 64; ++i)
          if ((c.autotune_bit_chord >> i) & 1) allowed.push_back(root + i + kNumHalftones * octave);
    } else {
      for (int h = 0; h < 36; ++h) {
        double add = static_cast<int>(0.5 + (kNumHalftones / 1.0) * std::log2(static_cast<float>(h + 1))); // midi.h:270-287
        for (int i = 0; i < 64; ++i)
          if ((c.autotune_bit_chord >> i) & 1) { allowed.push_back(root + add + i); allowed.push_back(root - add + i); }
      }
    }
    std::sort(allowed.begin(), allowed.end());
  } else if (c.autotune_type == RA_AUTOTUNE_INTERVALS) {
    int o = c.autotune_root_note - Do;
    if (o < 0) o += kNumHalftones;
    double const offset = o + c.autotune_root_transpose;
    allowed.push_back(offset);
    int const factor = c.autotune_factor;
    if (factor > 0) {
      for (double v = offset - factor; v > 0.; v -= factor) allowed.push_back(v);
      for (double v = offset + factor; v < 151.; v += factor) allowed.push_back(v);
    }
    std::sort(allowed.begin(), allowed.end());
  }
}

int validate(const ra_config &c, int32_t n_streams) {
  if (n_streams < 1) return fail(RA_EINVAL, "n_streams must be >= 1");
  if (c.sample_rate < 1000) return fail(RA_EINVAL, "sample_rate is too small");
  if (c.window_size < 2 || (c.window_size & 1)) return fail(RA_EINVAL, "window_size must be even and >= 2 (getEvenWindowSizeFrames)");
  if (fft_length_for(c.window_size) > RA_MAX_FFT) return fail(RA_EINVAL, "window_size is too large (FFT length > 16384)");
  if (c.stride < 1) return fail(RA_EINVAL, "window_center_stride is too small");
  if (c.n_channels < 1 || c.n_channels > RA_MAX_CHANNELS) return fail(RA_EINVAL, "n_channels must be in 1..256");
  if (c.min_volume < 0) return fail(RA_EINVAL, "min_volume is too small");
  if (c.nearby_distance_tones < 0) return fail(RA_EINVAL, "nearby_distance_tones is too small");
  if (c.max_track_pitches < 0) return fail(RA_EINVAL, "max_track_pitches is too small");
  if (c.max_pitches_per_frame < 0 || c.max_pitches_per_frame > 4096) return fail(RA_EINVAL, "max_pitches_per_frame must be in 0..4096 (0 = sized for the worst case)");
  if (c.vocoder_count_bands < 1 || c.vocoder_count_bands > 50) return fail(RA_EINVAL, "vocoder_count_bands must be in 1..50");
  if (c.window_kind < 0 || c.window_kind > 2) return fail(RA_EINVAL, "unknown window_kind");
  if (c.autotune_factor < 0) return fail(RA_EINVAL, "autotune_factor must be >= 0");
  if (c.analysis_output_feedback_gain != 0.f && c.reference_order > 1) return fail(RA_EINVAL, "analysis_output_feedback_gain needs reference_order 0 or 1");
  if (c.analysis_output_feedback_gain != 0.f && !(c.output_delay_seconds >= 0.f && c.output_delay_seconds <= 60.f)) return fail(RA_EINVAL, "output_delay_seconds must be in 0..60");
  if (c.vocoder_kind != RA_VOCODER_FFT && c.vocoder_kind != RA_VOCODER_BANDPASS) return fail(RA_EINVAL, "unknown vocoder_kind");
  if (c.vocoder_kind == RA_VOCODER_BANDPASS && !(c.vocoder_env_follower_cutoff_ratio > 0.f)) return fail(RA_EINVAL, "vocoder_env_follower_cutoff_ratio must be > 0");
  if (c.post != RA_POST_NONE && c.post != RA_POST_LIMIT) return fail(RA_EINVAL, "unknown post (RA_POST_NONE / RA_POST_LIMIT)");
  if (c.carrier_n_channels < 0 || c.carrier_n_channels > RA_MAX_CHANNELS) return fail(RA_EINVAL, "carrier_n_channels must be in 0..256");
  return RA_OK;
}

// The white-noise sound buffer the carrier's noise voice plays: getWhiteNoise(sample_rate) (source/sound.cpp:3-21) =
// soundBuffer(soundId{sample_rate, Sound::NOISE, noise_duration = .05f}) — the constant is consumed as a frequency (include/sound.h:53-60), so
// the buffer holds freq_to_period_in_samples(.05f, sample_rate) = int(float(sample_rate) / .05f) frames (include/sound.functions.h:45-50;
// 960 000 at 48 kHz) of white_gaussian_noise (sound.cpp:134-147 -> generate :84-100; include/noise.h:3-5), each a float widened to double.
// The draw itself — std::normal_distribution<float> over the OS-seeded mersenne<SEEDED::No>() — is this build's declared policy (DESIGN.md
// §5): sample i = Box-Muller of two 24-bit uniforms from the counter hash of (a fixed key, i), evaluated in double, rounded to float.
uint64_t mix64_noise(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
std::vector<double> build_white_noise(int sample_rate) {
  const float freq_hz = .05f, sr = static_cast<float>(sample_rate);
  const int period = std::max(1, static_cast<int>(sr / freq_hz));
  std::vector<double> v(static_cast<size_t>(period));
  for (size_t i = 0; i < v.size(); ++i) {
    const uint64_t h = mix64_noise(0x77686974656e6f69ull ^ mix64_noise(static_cast<uint64_t>(i) * 2 + 7));
    const double u1 = (static_cast<double>(h >> 40) + 1.0) * (1.0 / 16777216.0);            // (0, 1]
    const double u2 = static_cast<double>((h >> 16) & 0xffffffull) * (1.0 / 16777216.0);     // [0, 1)
    v[i] = static_cast<float>(std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2));
  }
  return v;
}

// twiddle / bit-reversal tables of a real FFT of length L done as an M = L/2 point complex FFT
int upload_fft_tables(ra_context *c, int L, double2 **tw_half, double2 **tw_full, int **bitrev) {
  int M = L / 2, log2M = 0, r;
  while ((1 << log2M) < M) ++log2M;
  std::vector<double2> tw(std::max(M / 2, 1));
  // separate cos and sin calls that cannot be merged into one sincos call (volatile): glibc's sincos differs from its sin / cos by an
  // ulp at a few arguments per table, and whether a compiler merges the pair is its own business (g++ -O2 does); the CPU restatement
  // builds its tables the same way
  auto twiddle = [](double a) { volatile double a_again = a; const double cs = std::cos(a); const double sn = std::sin(a_again); return make_double2(cs, sn); };
  for (int k = 0; k < M / 2; ++k) tw[k] = twiddle(-2.0 * M_PI * k / M);
  if ((r = upload(c, tw_half, tw))) return r;
  // the split's table e^{-2 pi i k / L}, k <= M: the lower half from cos / sin, the upper half DEFINED as its mirror image W^(M - k) = -conj(W^k)
  // (the CPU restatement builds the same table): bins k and M - k then share one twiddle product (ra_real_split_pair)
  std::vector<double2> tf(M + 1);
  for (int k = 0; k <= M / 2; ++k) tf[k] = twiddle(-2.0 * M_PI * k / L);
  for (int k = M / 2 + 1; k <= M; ++k) tf[k] = make_double2(-tf[M - k].x, tf[M - k].y);
  if ((r = upload(c, tw_full, tf))) return r;
  std::vector<int> rev(std::max(M, 1), 0);
  for (int i = 0; i < M; ++i) {
    int x = 0;
    for (int bit = 0; bit < log2M; ++bit) if (i & (1 << bit)) x |= 1 << (log2M - 1 - bit);
    rev[i] = x;
  }
  return upload(c, bitrev, rev);
}

std::vector<double> make_half_window(int kind, int n) {
  std::vector<double> hw(n);
  double sigma = n / 4.0, sum = 0;
  for (int j = 0; j < n; ++j) {
    double d = j + 0.5;
    hw[j] = kind == RA_WINDOW_GAUSSIAN4 ? std::exp(-(d * d) / (2.0 * sigma * sigma)) : kind == RA_WINDOW_HANN ? 0.5 * (1.0 + std::cos(M_PI * d / n)) : 1.0;
    sum += hw[j];
  }
  for (auto &v : hw) v /= sum;
  return hw;
}

// band of every bin: the single pass of FFTModulator's on_fft_result (vocoder.cpp:137-160); -1 = no band
std::vector<int> band_of_bins(const std::vector<double> &freqs, int n_bins, double bin_to_Hz) {
  std::vector<int> band(n_bins, -1);
  size_t lo = 0, hi = 1, end = freqs.size();
  int bi = 0;
  for (int i = 0; i < n_bins; ++i) {
    double const Hz = bin_to_Hz * i;
    if (Hz <= freqs[lo]) continue;
    while (Hz > freqs[hi]) { ++bi; ++hi; ++lo; if (hi == end) break; }
    if (hi == end) break;
    band[i] = bi;
  }
  return band;
}

int build_vocoder_tables(ra_context *c) {
  const ra_config &cfg = c->cfg;
  RaParams &p = c->p;
  RaBuffers &b = c->b;
  int r;
  // good_stride (vocoder.cpp:84-92) and the modulator window lambda (:107-119)
  int S = std::max(1, static_cast<int>(0.5f + cfg.vocoder_stride_seconds * cfg.sample_rate));
  if (0 == S % 2) ++S;
  int Wm = std::max(1, static_cast<int>(0.5f + cfg.sample_rate * static_cast<double>(cfg.vocoder_modulator_window_size_seconds)));
  if (1 == Wm % 2) ++Wm;
  Wm = std::min(1 << 16, Wm);
  p.voc_S = S;
  p.voc_Wm = Wm;
  p.voc_Lm = fft_length_for(Wm);
  p.voc_Mm = p.voc_Lm / 2;
  p.voc_Lc = fft_length_for(2 * S);
  p.voc_Mc = p.voc_Lc / 2;
  p.voc_bands = cfg.vocoder_count_bands;
  if (cfg.vocoder_kind == RA_VOCODER_FFT) {
    // the reference's own limits (vocoder.cpp:94-96): modulator FFT <= 2^16 (the window is clamped to it above, :111-115), stride <= 2^16,
    // carrier FFT <= 2^17.  Transforms that do not fit the 160 KB of LDS run on buffers in HBM (ra_vocoder.hip: BIG).
    if (S > (1 << 16)) return fail(RA_EINVAL, "vocoder stride is too large (max_stride_size = 65536 frames, vocoder.cpp:96)");
    if (p.voc_Lm > (1 << 16) || p.voc_Lc > (1 << 17)) return fail(RA_EINVAL, "vocoder FFT length above the reference's maximum (65536 / 131072)");
  }
  if (!(cfg.vocoder_min_freq > 0.f) || !(cfg.vocoder_max_freq > cfg.vocoder_min_freq)) return fail(RA_EINVAL, "vocoder band limits must satisfy 0 < min < max");
  // SetupParams::fill_freqs, vocoder.cpp:15-27 (float logs, double exp)
  std::vector<double> freqs;
  {
    auto log_min = std::log(cfg.vocoder_min_freq);
    auto log_max = std::log(cfg.vocoder_max_freq);
    for (int i = 0; i <= p.voc_bands; ++i) {
      double const ratio = static_cast<double>(i) / p.voc_bands;
      freqs.push_back(std::exp(log_min + ratio * (log_max - log_min)));
    }
  }
  if (cfg.vocoder_kind == RA_VOCODER_BANDPASS) {
    // BandPass::setup / EnvelopeFollower::setup (vocoder.cpp:48-52,67-69) through Modulator::setup_bands / setup (:315-354) and
    // Carrier::setup (:669-692): initWithFreq(sample_rate, f) = initWithAngleIncrement(2 f / sample_rate), one-pole k = pi w / (1 + pi w)
    std::vector<double> k(3 * 64, 0.0);
    auto coeff = [&](double freq) { const double w = 2 * freq / cfg.sample_rate, a = M_PI * w; return a / (1.0 + a); };
    for (int i = 0; i < p.voc_bands; ++i) {
      k[i] = coeff(freqs[i]);
      k[64 + i] = coeff(freqs[i + 1]);
      k[128 + i] = coeff(freqs[i] * cfg.vocoder_env_follower_cutoff_ratio);
    }
    if ((r = upload(c, &b.bank_k, k))) return r;
    // the ratios of the time-parallel bank's scans: a lane's chunk is 16 samples, so (1 - k)^16 carries a state from one lane to the next, its
    // repeated squares over 2, 4, ... 32 lanes
    std::vector<double> pw(3 * 8 * 64, 0.0);
    for (int t = 0; t < 3; ++t)
      for (int i = 0; i < 64; ++i) {
        double a = 1.0 - k[t * 64 + i];
        for (int q = 0; q < 4; ++q) a *= a; // ^16
        for (int e = 0; e < 8; ++e) { pw[(t * 8 + e) * 64 + i] = a; a *= a; }
      }
    return upload(c, &b.bank_pow, pw);
  }
  if ((r = upload(c, &b.voc_half_window, make_half_window(RA_WINDOW_GAUSSIAN4, Wm / 2)))) return r;
  if ((r = upload_fft_tables(c, p.voc_Lm, &b.vm_tw_half, &b.vm_tw_full, &b.vm_bitrev))) return r;
  if ((r = upload_fft_tables(c, p.voc_Lc, &b.vc_tw_half, &b.vc_tw_full, &b.vc_bitrev))) return r;
  { // modulator: bins 0..Lm/2-1, bands are contiguous runs of bins
    std::vector<int> band = band_of_bins(freqs, p.voc_Mm, static_cast<double>(cfg.sample_rate) / p.voc_Lm);
    std::vector<int> range(2 * p.voc_bands, 0); // [first bin, one past last bin) per band; empty band = [0, 0)
    for (int bi = 0; bi < p.voc_bands; ++bi) {
      int s0 = -1, cnt = 0;
      for (int i = 0; i < p.voc_Mm; ++i) if (band[i] == bi) { if (s0 < 0) s0 = i; ++cnt; }
      if (s0 >= 0) { range[2 * bi] = s0; range[2 * bi + 1] = s0 + cnt; }
    }
    if ((r = upload(c, &b.voc_mod_band_start, range))) return r;
    // the same as a per-bin table for the fused modulator kernel: non-decreasing, -1 below the bands, `bands` above them and for bin M
    std::vector<int> bb(p.voc_Mm + 1, p.voc_bands);
    bool seen = false; // (the bands tile [min, max]: a bin without a band lies below the first one or above the last one)
    for (int i = 0; i < p.voc_Mm; ++i) {
      if (band[i] >= 0) { bb[i] = band[i]; seen = true; }
      else bb[i] = seen ? p.voc_bands : -1;
    }
    if ((r = upload(c, &b.voc_mod_bin_band, bb))) return r;
  }
  if ((r = upload(c, &b.voc_car_band, band_of_bins(freqs, p.voc_Mc + 1, static_cast<double>(cfg.sample_rate) / p.voc_Lc)))) return r;
  { // EqualGainXFade::set(S-1, Sinusoidal).get(idx+1) — decision of DESIGN.md §5: new = (1 - cos(pi i / (len+1))) / 2
    std::vector<double> xf(S);
    for (int idx = 0; idx < S; ++idx) xf[idx] = 0.5 * (1.0 - std::cos(M_PI * (idx + 1) / ((S - 1) + 1.0)));
    if ((r = upload(c, &b.voc_xf_new, xf))) return r;
  }
  return RA_OK;
}

int build_tables(ra_context *c) {
  const ra_config &cfg = c->cfg;
  RaParams &p = c->p;
  std::memset(&p, 0, sizeof(p));
  p.sample_rate = cfg.sample_rate;
  p.N = cfg.window_size;
  p.H = cfg.stride;
  p.L = fft_length_for(p.N);
  p.M = p.L / 2;
  p.log2M = 0;
  while ((1 << p.log2M) < p.M) ++p.log2M;
  p.P = cfg.n_channels;
  p.Ppad = ((p.P + 63) / 64) * 64;
  p.maxr = cfg.max_pitches_per_frame;
  if (p.maxr == 0) {
    // Sized like the reference's growing vectors (lib.cpp:1102-1112) would end up in the worst case: a spectrum of M bins has at most
    // M / 2 local maxima, each at most one list entry (two with a harmonize step: the lists then hold twice maxr).  Bounded by the LDS
    // the analysis (FFT buffer + two lists) and the tracking (40 bytes per entry) workgroups may hold; from FFT length 16384 on that
    // bound is below M / 2 and RA_EOVERFLOW stays possible for noise-like spectra.
    const bool harm = cfg.pitch_harmonize_pre_autotune != 0.f || cfg.pitch_harmonize_post_autotune != 0.f;
    const size_t fft_bytes = (size_t)std::max(p.M, 16) * sizeof(double2), lim = RA_LDS_LIMIT_BYTES; // ra_zsize(M) slots
    const size_t fit_a = fft_bytes + 4096 < lim ? (lim - fft_bytes - 4096) / (2 * sizeof(double2) * (harm ? 2 : 1)) : 1;
    const size_t fit_t = ((size_t)lim - (size_t)p.P * 88 - 64) / 40;
    p.maxr = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(fit_a, fit_t), std::min<size_t>(4096, std::max(16, p.M / 2))));
  }
  p.min_volume = static_cast<double>(cfg.min_volume);
  p.nearby = static_cast<double>(cfg.nearby_distance_tones);
  p.max_track = static_cast<double>(cfg.max_track_pitches);
  p.tolerance = cfg.autotune_tolerance_pitches;
  p.shift_pre = cfg.pitch_shift_pre_autotune;
  p.shift_post = cfg.pitch_shift_post_autotune;
  p.harm_pre = cfg.pitch_harmonize_pre_autotune;
  p.harm_post = cfg.pitch_harmonize_post_autotune;
  p.analysis_volume = cfg.analysis_volume;
  p.input_gain = cfg.analysis_input_gain;
  p.stereo_spread = cfg.stereo_spread;
  p.phase_policy = cfg.phase_policy;
  p.seed = cfg.seed;
  p.half_tone_ratio = std::pow(2., 1.0 / kNumHalftones); // midi.h:238-239
  { // mkAHDSR (lib.cpp:63-82) then setAHDSR (audioelement.h:923-952)
    int a = ms_to_frames(1000.f * cfg.env_attack_seconds, cfg.sample_rate), h = ms_to_frames(1000.f * cfg.env_hold_seconds, cfg.sample_rate);
    int d = ms_to_frames(1000.f * cfg.env_decay_seconds, cfg.sample_rate), r = ms_to_frames(1000.f * cfg.env_release_seconds, cfg.sample_rate);
    int min_dt = cfg.sample_rate / 1000;
    bool hasDecay = cfg.env_sustain_level < 0.999999;
    p.SMinusOne = hasDecay ? std::min(std::max(cfg.env_sustain_level, 0.f), 1.f) - 1.f : 0.f;
    p.envA = std::max(a, min_dt);
    p.envH = std::max(h, 0);
    p.envD = hasDecay ? std::max(d, min_dt) : 0;
    p.envR = std::max(r, min_dt);
  }
  p.vol_mod = cfg.voice_volume;
  p.vol_car = cfg.carrier_volume;
  p.vol_voc = cfg.vocoder_volume;
  p.render_mode = (cfg.reference_order == 2 || cfg.reference_order == 3) ? cfg.reference_order : 0;
  p.render_pack = cfg.reference_order != 4;
  p.fb_gain = cfg.analysis_output_feedback_gain;
  p.fb_sz = std::max(0, static_cast<int>(0.5f + cfg.output_delay_seconds * cfg.sample_rate)); // lib.cpp:1638-1639
  p.voc_enabled = (cfg.voice_volume != 0.f || cfg.carrier_volume != 0.f || cfg.vocoder_volume != 0.f) ? 1 : 0;
  p.post_limit = cfg.post == RA_POST_LIMIT ? 1 : 0;
  p.voc_kind = cfg.vocoder_kind;
  p.car_P = p.voc_enabled ? cfg.carrier_n_channels : 0; // nothing hears the carrier synth while every vocoder volume is 0
  p.car_Ppad = ((p.car_P + 63) / 64) * 64;
  p.car_CH = 4;
  while (p.car_CH < p.car_P) p.car_CH *= 2;
  p.car_serial = cfg.reference_order == 1;
  p.car_vol[0] = cfg.vocoder_carrier_noise_volume; p.car_vol[1] = cfg.vocoder_carrier_saw_volume; p.car_vol[2] = cfg.vocoder_carrier_triangle_volume;
  p.car_vol[3] = cfg.vocoder_carrier_square_volume; p.car_vol[4] = cfg.vocoder_carrier_sine_volume; p.car_vol[5] = cfg.vocoder_carrier_pulse_volume;
  p.car_pulse_width = cfg.vocoder_carrier_pulse_width;

  RaBuffers &b = c->b;
  std::memset(&b, 0, sizeof(b));
  int r;
  // half window: Gaussian truncated at 4 sigma, normalised so the full window sums to 2
  if ((r = upload(c, &b.half_window, make_half_window(cfg.window_kind, p.N / 2)))) return r;
  if ((r = upload_fft_tables(c, p.L, &b.tw_half, &b.tw_full, &b.bitrev))) return r;
  { // include/loudness.h: pitches table (:43-54) and the 60 phon row of elvs (:199-229, lib.cpp:1717)
    std::vector<double> lp(29);
    std::vector<float> elv(29);
    for (int i = 0; i < 29; ++i) { lp[i] = freq_to_pitch(kLoudFreqs[i]); elv[i] = equal_loudness_volume(i, 60.f); }
    if ((r = upload(c, &b.loud_pitches, lp))) return r;
    if ((r = upload(c, &b.loud_elv, elv))) return r;
  }
  {
    std::vector<double> allowed;
    build_autotune(cfg, p, allowed);
    p.n_allowed = static_cast<int>(allowed.size());
    if ((r = upload(c, &b.allowed, allowed))) return r;
  }
  if (p.voc_enabled && (r = build_vocoder_tables(c))) return r;
  if (p.car_P > 0 && cfg.vocoder_carrier_noise_volume != 0.f) { // the noise voice's sound buffer: 7.7 MB at 48 kHz, shared by every stream and channel
    std::vector<double> nz = build_white_noise(cfg.sample_rate);
    b.car_noise_len = static_cast<int32_t>(nz.size());
    if ((r = upload(c, &b.car_noise, nz))) return r;
  }
  return RA_OK;
}

int alloc_state(ra_context *c) {
  RaParams &p = c->p;
  RaBuffers &b = c->b;
  int r;
  size_t S = p.n_streams;
  if ((r = dev_alloc(c, &b.hist, S * p.N))) return r;
  if ((r = dev_alloc(c, &b.track, S))) return r;
  if ((r = dev_alloc(c, &b.played, S * p.P))) return r;
  if ((r = dev_alloc(c, &b.chan_free_at, S * p.P))) return r;
  if ((r = dev_alloc(c, &b.chan_held, S * p.P))) return r;
  if ((r = dev_alloc(c, &b.chan_minch, S * p.P))) return r;
  if ((r = dev_alloc(c, &b.sd, S * SD_COUNT * p.Ppad))) return r;
  if ((r = dev_alloc(c, &b.si, S * SI_COUNT * p.Ppad))) return r;
  if ((r = dev_alloc(c, &b.error_flags, 1))) return r;
  if (p.fb_gain != 0.f) { // the mono output of the fb_sz + N + 2 frames before the block: what the block's first windows look back at
    b.fb_hist_len = (int64_t)p.fb_sz + p.N + 2;
    if ((r = dev_alloc(c, &b.fb_hist, S * b.fb_hist_len))) return r;
  }
  if (p.voc_enabled && p.voc_kind == RA_VOCODER_FFT) {
    if ((r = dev_alloc(c, &b.voc_mod_hist, S * p.voc_Wm))) return r;
    if ((r = dev_alloc(c, &b.voc_car_hist, S * 2 * p.voc_S))) return r;
    if ((r = dev_alloc(c, &b.voc_amp_prev, S * p.voc_bands))) return r;
    if ((r = dev_alloc(c, &b.voc_sig_prev, S * 4 * p.voc_S))) return r;
    // transforms and history rolls that do not fit LDS work on buffers in HBM (ra_vocoder.hip)
    const size_t mod_slots = (size_t)std::max(p.voc_Mm, 16), car_slots = (size_t)std::max(p.voc_Mc, 16) + std::max(p.voc_Mc, 1);
    size_t big = 0;
    if (mod_slots * sizeof(double2) > (size_t)RA_LDS_LIMIT_BYTES) big = std::max(big, mod_slots);
    if (car_slots * sizeof(double2) > (size_t)RA_LDS_LIMIT_BYTES) big = std::max(big, car_slots);
    if (big) {
      b.voc_big_slot = (int64_t)big;
      if ((r = dev_alloc(c, &b.voc_big, ((size_t)RA_VOC_BIG_WORKGROUPS + S) * big, false))) return r;
    }
    const size_t roll_bytes = std::max((size_t)4 * p.voc_S * sizeof(double), (size_t)p.voc_Wm * sizeof(float));
    if (roll_bytes > (size_t)RA_LDS_LIMIT_BYTES) {
      b.voc_roll_slot = (int64_t)((roll_bytes + sizeof(double) - 1) / sizeof(double));
      if ((r = dev_alloc(c, &b.voc_roll_tmp, S * (size_t)b.voc_roll_slot, false))) return r;
    }
  }
  if (p.voc_enabled && p.voc_kind == RA_VOCODER_BANDPASS && (r = dev_alloc(c, &b.bank_state, S * 5 * 64))) return r;
  if (p.post_limit && (r = dev_alloc(c, &b.lim_state, S * 2))) return r;
  if (p.car_P > 0) {
    if ((r = dev_alloc(c, &b.car_sd, S * SD_COUNT * p.car_Ppad))) return r;
    if ((r = dev_alloc(c, &b.car_si, S * SI_COUNT * p.car_Ppad))) return r;
    if ((r = dev_alloc(c, &b.car_ext, S * CX_COUNT * p.car_Ppad))) return r;
    if ((r = dev_alloc(c, &b.car_noteid, S * p.car_Ppad))) return r;
    if ((r = dev_alloc(c, &b.car_dropped, S))) return r;
    if ((r = dev_alloc(c, &b.car_ev_begin, S + 1))) return r;
  }
  return RA_OK;
}

// state right after construction: envelopes Done2 (audioelement.h:358-360), oscillators at angle 0 with increment 0 (:2376-2377, 2438-2441)
int init_state(ra_context *c) {
  RaParams &p = c->p;
  RaBuffers &b = c->b;
  size_t S = p.n_streams;
  HIP_TRY(hipMemsetAsync(b.hist, 0, S * p.N * sizeof(float), c->stream));
  if (b.fb_hist) HIP_TRY(hipMemsetAsync(b.fb_hist, 0, S * b.fb_hist_len * sizeof(float), c->stream));
  HIP_TRY(hipMemsetAsync(b.track, 0, S * sizeof(RaTrackState), c->stream));
  HIP_TRY(hipMemsetAsync(b.played, 0, S * p.P * sizeof(RaPlayed), c->stream));
  HIP_TRY(hipMemsetAsync(b.chan_free_at, 0, S * p.P * sizeof(int64_t), c->stream));
  HIP_TRY(hipMemsetAsync(b.chan_held, 0, S * p.P * sizeof(int32_t), c->stream));
  HIP_TRY(hipMemsetAsync(b.chan_minch, 0, S * p.P * sizeof(int32_t), c->stream));
  HIP_TRY(hipMemsetAsync(b.error_flags, 0, sizeof(int32_t), c->stream));
  if (b.lim_state) { // Limiter{}: compression = target = 1
    std::vector<double> one(S * 2, 1.0);
    HIP_TRY(hipMemcpyAsync(b.lim_state, one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (b.bank_state) HIP_TRY(hipMemsetAsync(b.bank_state, 0, S * 5 * 64 * sizeof(double), c->stream)); // Filter<> starts from rest
  if (p.car_P > 0) { // fresh elements: envelopes Done2, Phased angle 0 / increments 0 / aliasing gain 1, sine phasor 1 (audioelement.h:1473-1476,2438-2441)
    std::vector<double> sd(S * SD_COUNT * p.car_Ppad, 0.0), ext(S * CX_COUNT * p.car_Ppad, 0.0);
    std::vector<int32_t> si(S * SI_COUNT * p.car_Ppad, 0);
    std::vector<int64_t> ids(S * p.car_Ppad, -1);
    for (size_t s = 0; s < S; ++s)
      for (int l = 0; l < p.car_Ppad; ++l) {
        double *d = sd.data() + s * SD_COUNT * p.car_Ppad;
        int32_t *i = si.data() + s * SI_COUNT * p.car_Ppad;
        d[SD_OSC_CRE * p.car_Ppad + l] = 1.0; d[SD_OSC_MRE * p.car_Ppad + l] = 1.0; d[SD_OSC_ALIAS * p.car_Ppad + l] = 1.0;
        i[SI_ENV_STATE * p.car_Ppad + l] = ENV_DONE2; i[SI_ENV_RELDELAY * p.car_Ppad + l] = -1; i[SI_ENV_AHDSTATE * p.car_Ppad + l] = -1;
        ext[s * CX_COUNT * p.car_Ppad + CX_PH_ALIAS * p.car_Ppad + l] = 1.0;
        ext[s * CX_COUNT * p.car_Ppad + CX_NZ_INDEX * p.car_Ppad + l] = -1.0; // soundBufferWrapperAlgo::index, audioelement.h:1581
      }
    HIP_TRY(hipMemcpyAsync(b.car_sd, sd.data(), sd.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(b.car_si, si.data(), si.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(b.car_ext, ext.data(), ext.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(b.car_noteid, ids.data(), ids.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(b.car_dropped, 0, S * sizeof(int32_t), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); // the host images go out of scope
    c->car_pending.assign(S, {});
    c->car_noteids.assign(S, {});
    c->car_last_sample.assign(S, 0);
  }
  if (p.voc_enabled && p.voc_kind == RA_VOCODER_FFT) {
    HIP_TRY(hipMemsetAsync(b.voc_mod_hist, 0, S * p.voc_Wm * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(b.voc_car_hist, 0, S * 2 * p.voc_S * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(b.voc_amp_prev, 0, S * p.voc_bands * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(b.voc_sig_prev, 0, S * 4 * p.voc_S * sizeof(double), c->stream));
  }
  std::vector<double> sd(S * SD_COUNT * p.Ppad, 0.0);
  std::vector<int32_t> si(S * SI_COUNT * p.Ppad, 0);
  for (size_t s = 0; s < S; ++s) {
    double *d = sd.data() + s * SD_COUNT * p.Ppad;
    int32_t *i = si.data() + s * SI_COUNT * p.Ppad;
    for (int l = 0; l < p.Ppad; ++l) {
      d[SD_OSC_CRE * p.Ppad + l] = 1.0;
      d[SD_OSC_MRE * p.Ppad + l] = 1.0;
      d[SD_OSC_ALIAS * p.Ppad + l] = 1.0;
      d[SD_GAIN_L * p.Ppad + l] = 0.0;
      i[SI_ENV_STATE * p.Ppad + l] = ENV_DONE2;
      i[SI_ENV_RELDELAY * p.Ppad + l] = -1;
      i[SI_ENV_AHDSTATE * p.Ppad + l] = -1;
    }
  }
  HIP_TRY(hipMemcpyAsync(b.sd, sd.data(), sd.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(b.si, si.data(), si.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->t = 0;
  c->frames_fired = 0;
  c->last_f_lo = c->last_nfb = c->last_n = c->last_t0 = 0;
  {
    ra_context::Shim &sh = c->shim;
    sh.in_voice.clear(); sh.in_carrier.clear(); sh.carrier_given.clear(); sh.ring.clear(); sh.ring_head.clear(); sh.served.clear();
    sh.in_base = sh.rendered = 0; sh.deferred = false; sh.active = false;
  }
  return RA_OK;
}

int ensure_scratch(ra_context *c, int64_t nfb, int64_t n, int n_chunks) {
  RaParams &p = c->p;
  RaBuffers &b = c->b;
  int r;
  // both sets grow together (on the first block, or for a longer block than ever before): an allocation must not land in the second block
  for (ra_context::Scratch &sc : c->scr) {
    if (nfb > sc.frames_cap) {
      HIP_TRY(hipDeviceSynchronize()); // nothing may still read the old buffers
      dev_free(c, sc.pl_pitch); dev_free(c, sc.pl_vol); dev_free(c, sc.pl_key); dev_free(c, sc.pl_count); dev_free(c, sc.cmd); dev_free(c, sc.cmd_non);
      dev_free(c, sc.pl_nf); dev_free(c, sc.pl_nv); dev_free(c, sc.pl_nmc); dev_free(c, sc.pl_ord);
      // equal-length blocks differ by one frame depending on where they start: leave room, a reallocation synchronises the device
      size_t S = p.n_streams, F = static_cast<size_t>(nfb + 2);
      sc.frames_cap = nfb + 2;
      if ((r = dev_alloc(c, &sc.pl_pitch, S * F * p.maxr, false))) return r;
      if ((r = dev_alloc(c, &sc.pl_vol, S * F * p.maxr, false))) return r;
      if ((r = dev_alloc(c, &sc.pl_key, S * F * p.maxr, false))) return r;
      if ((r = dev_alloc(c, &sc.pl_count, S * F))) return r;
      if ((r = dev_alloc(c, &sc.pl_nf, S * F * p.maxr, false))) return r;
      if ((r = dev_alloc(c, &sc.pl_nv, S * F * p.maxr, false))) return r;
      if ((r = dev_alloc(c, &sc.pl_nmc, S * F * p.maxr, false))) return r;
      if ((r = dev_alloc(c, &sc.pl_ord, S * F * p.maxr, false))) return r;
      if ((r = dev_alloc(c, &sc.cmd, S * F * p.Ppad, false))) return r;
      if ((r = dev_alloc(c, &sc.cmd_non, S * F))) return r;
      HIP_TRY(hipStreamSynchronize(c->stream)); // the zero fills above ran on the caller's stream
    }
    if (c->cfg.reference_order != 1 && nfb + n_chunks > sc.seg_cap) {
      HIP_TRY(hipDeviceSynchronize());
      dev_free(c, sc.seg_sd); dev_free(c, sc.seg_si);
      const int64_t extra = std::max<int64_t>(kMaxChunks, n_chunks);
      size_t S = p.n_streams, G = static_cast<size_t>(nfb + 2 + extra);
      sc.seg_cap = nfb + 2 + extra;
      if ((r = dev_alloc(c, &sc.seg_sd, S * G * SD_HOP_COUNT * p.Ppad, false))) return r;
      if ((r = dev_alloc(c, &sc.seg_si, S * G * SI_HOP_COUNT * p.Ppad, false))) return r;
    }
  }
  ra_context::Scratch &sc = c->scr[c->parity];
  b.frames_cap = sc.frames_cap; b.seg_cap = sc.seg_cap;
  b.pl_pitch = sc.pl_pitch; b.pl_vol = sc.pl_vol; b.pl_key = sc.pl_key; b.pl_count = sc.pl_count;
  b.pl_nf = sc.pl_nf; b.pl_nv = sc.pl_nv; b.pl_nmc = sc.pl_nmc; b.pl_ord = sc.pl_ord;
  b.cmd = sc.cmd; b.cmd_non = sc.cmd_non; b.seg_sd = sc.seg_sd; b.seg_si = sc.seg_si;
  if (p.post_limit && n > b.post_cap) {
    HIP_TRY(hipDeviceSynchronize());
    dev_free(c, b.post_d);
    b.post_cap = n;
    if ((r = dev_alloc(c, &b.post_d, (size_t)p.n_streams * n * 2, false))) return r;
  }
  if (p.fb_gain != 0.f && n > b.fb_block_cap) {
    HIP_TRY(hipDeviceSynchronize());
    dev_free(c, b.fb_block);
    b.fb_block_cap = n;
    if ((r = dev_alloc(c, &b.fb_block, (size_t)p.n_streams * n, false))) return r;
  }
  if (p.voc_enabled) {
    size_t S = p.n_streams;
    for (ra_context::Scratch &sc2 : c->scr) {
      if (n > sc2.mix_cap) {
        HIP_TRY(hipDeviceSynchronize()); // an earlier block may still use the old buffer
        dev_free(c, sc2.voc_mix);
        sc2.mix_cap = n;
        if ((r = dev_alloc(c, &sc2.voc_mix, S * n, false))) return r;
      }
    }
    b.voc_mix = c->scr[c->parity].voc_mix; b.mix_cap = c->scr[c->parity].mix_cap;
    if (p.car_P > 0) { // the generated carrier exists per scratch set: block k + 1's synth writes one while block k's vocoder reads the other
      for (ra_context::Scratch &sc2 : c->scr) {
        if (n > sc2.car_gen_cap) {
          HIP_TRY(hipDeviceSynchronize()); // an earlier block may still use the old buffer
          dev_free(c, sc2.car_gen_d); dev_free(c, sc2.car_gen_f);
          sc2.car_gen_cap = n;
          if ((r = dev_alloc(c, &sc2.car_gen_d, S * n, false))) return r;
          if ((r = dev_alloc(c, &sc2.car_gen_f, S * n, false))) return r;
        }
      }
      b.car_gen_d = c->scr[c->parity].car_gen_d; b.car_gen_f = c->scr[c->parity].car_gen_f; b.car_gen_cap = c->scr[c->parity].car_gen_cap;
    }
    RaFrameRange mod = ra_frames_firing(p.voc_Wm, p.voc_S, c->t, n), car = ra_frames_firing(2 * p.voc_S, p.voc_S, c->t, n);
    if (p.voc_kind != RA_VOCODER_FFT) { mod.count = 0; car.count = 0; }
    if (mod.count > b.voc_mod_cap) {
      HIP_TRY(hipDeviceSynchronize()); // an earlier block may still use the old buffer on the render stream
      dev_free(c, b.voc_amp);
      b.voc_mod_cap = mod.count + 2;
      if ((r = dev_alloc(c, &b.voc_amp, S * b.voc_mod_cap * p.voc_bands, false))) return r;
    }
    if (car.count > b.voc_car_cap) {
      HIP_TRY(hipDeviceSynchronize()); // an earlier block may still use the old buffer on the render stream
      dev_free(c, b.voc_sig);
      b.voc_car_cap = car.count + 2;
      if ((r = dev_alloc(c, &b.voc_sig, S * b.voc_car_cap * 2 * p.voc_S, false))) return r;
    }
  }
  return RA_OK;
}

// frames with fire sample in [t0, t0 + n)
void frames_in_block(const RaParams &p, int64_t t0, int64_t n, int64_t &f_lo, int64_t &nfb) {
  int64_t first = (int64_t)p.N + 1;
  int64_t a = t0 - first;
  f_lo = a <= 0 ? 0 : (a + p.H - 1) / p.H;
  int64_t bnd = t0 + n - 1 - first;
  if (bnd < 0) { nfb = 0; return; }
  int64_t f_hi = bnd / p.H;
  nfb = f_hi >= f_lo ? f_hi - f_lo + 1 : 0;
}

// ---- per-launch timing with HIP events on the stream each kernel is launched on -------------------------------------
enum { ST_ANALYSIS = 0, ST_TRACK, ST_VOCODER, ST_PROPAGATE, ST_RENDER, ST_HISTORY, ST_PREP, ST_COUNT }; // ST_PREP: ra_track_prep_kernel, timed on its own since r04

int resolve_timing(ra_context *c) {
  for (auto &t : c->timed) {
    HIP_TRY(hipEventSynchronize(t.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
    c->kernel_ms[t.stage] += ms;
    ++c->kernel_launches[t.stage];
    c->ev_pool.push_back(t.a);
    c->ev_pool.push_back(t.b);
  }
  c->timed.clear();
  for (auto &sp : c->spare) { c->ev_pool.push_back(sp.first); c->ev_pool.push_back(sp.second); } // (recorded before / after launches that have completed by now)
  c->spare.clear();
  return RA_OK;
}

template <typename F>
int timed_launch(ra_context *c, int stage, hipStream_t st, F &&launch) {
  // a refused launch (grid / LDS / register limits) is reported by the launch that failed, not by whatever consumes its output later
  auto checked = [&]() -> int {
    launch();
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(RA_EHIP, std::string("kernel launch failed: ") + hipGetErrorString(le));
    return RA_OK;
  };
  if (!c->profiling) return checked();
  auto take = [&](hipEvent_t &e) -> int {
    if (c->ev_pool.empty()) { HIP_TRY(hipEventCreate(&e)); }
    else { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
    return RA_OK;
  };
  hipEvent_t ev[2] = {nullptr, nullptr}, xe[2] = {nullptr, nullptr};
  for (auto &e : ev) RA_TRY(take(e));
  // the throughput kernels are timed by the launch itself (RaExtTiming); the pair around the launch stays as the fallback for the launchers
  // that do not take it (reference-order and diagnostic kernels)
  const bool ext = stage == ST_ANALYSIS || stage == ST_RENDER;
  RaExtTiming xt{nullptr, nullptr, false};
  if (ext) { for (auto &e : xe) RA_TRY(take(e)); xt.a = xe[0]; xt.b = xe[1]; ra_ext_timing = &xt; }
  HIP_TRY(hipEventRecord(ev[0], st));
  const int lr = checked();
  ra_ext_timing = nullptr;
  if (lr) { for (auto e : ev) c->ev_pool.push_back(e); if (ext) for (auto e : xe) c->ev_pool.push_back(e); return lr; }
  HIP_TRY(hipEventRecord(ev[1], st));
  if (xt.used) { c->timed.push_back({stage, xe[0], xe[1]}); c->spare.push_back({ev[0], ev[1]}); }
  else { c->timed.push_back({stage, ev[0], ev[1]}); if (ext) for (auto e : xe) c->ev_pool.push_back(e); }
  if (c->timed.size() > 8192) return resolve_timing(c); // bounds the event pool when nobody reads the timers (waits for those launches)
  return RA_OK;
}

// makes the caller's stream wait for everything the internal streams still hold (asynchronous blocks); no host synchronisation
int join_streams(ra_context *c) {
  if (!c->pending_join || !c->pipe_ready) { c->pending_join = false; return RA_OK; }
  for (int i = 0; i < 6; ++i) {
    HIP_TRY(hipEventRecord(c->ev_join[i], c->pipe_stream[i]));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join[i], 0));
  }
  c->pending_join = false;
  return RA_OK;
}

// The carrier synth's note events of the block [t0, t0 + n): taken from the per-stream queues into one list, stream after stream.
// Everything that can fail happens before a queue is touched.
int stage_carrier_events(ra_context *c, int64_t t0, int64_t n) {
  const RaParams &p = c->p;
  RaBuffers &b = c->b;
  const int S = p.n_streams;
  size_t total = 0;
  for (int s = 0; s < S; ++s)
    for (const RaCarEv &e : c->car_pending[s]) { if (e.sample >= t0 + n) break; ++total; }
  if ((int64_t)total > b.car_ev_cap) {
    HIP_TRY(hipDeviceSynchronize()); // an earlier block may still read the old list
    dev_free(c, b.car_ev);
    b.car_ev_cap = std::max<int64_t>(256, 2 * (int64_t)total);
    int r = dev_alloc(c, &b.car_ev, (size_t)b.car_ev_cap, false);
    if (r) { b.car_ev_cap = 0; return r; }
  }
  // the segment table of the time-parallel synth (ra_carrier.hip): per stream one slot per 256 samples, one for the ragged end, one per event
  c->car_total_segs = p.car_serial ? 0 : (int64_t)S * ((n + 255) / 256 + 1) + (int64_t)total;
  if (c->car_total_segs > b.car_seg_cap) {
    HIP_TRY(hipDeviceSynchronize());
    dev_free(c, b.car_seg_info); dev_free(c, b.car_seg_d); dev_free(c, b.car_seg_i);
    b.car_seg_cap = c->car_total_segs + c->car_total_segs / 8;
    int r = dev_alloc(c, &b.car_seg_info, (size_t)b.car_seg_cap, false);
    if (!r) r = dev_alloc(c, &b.car_seg_d, (size_t)b.car_seg_cap * ra_carrier_seg_doubles() * p.car_CH, false);
    if (!r) r = dev_alloc(c, &b.car_seg_i, (size_t)b.car_seg_cap * SI_COUNT * p.car_CH, false);
    if (r) { b.car_seg_cap = 0; return r; }
  }
  ra_context::CarStage &cs = c->car_stage[c->parity];
  if (cs.used) HIP_TRY(hipEventSynchronize(cs.done)); // the copies of the block that used this set two blocks ago
  const size_t need = total * sizeof(RaCarEv) + (size_t)(S + 1) * sizeof(int32_t);
  if (need > cs.cap) {
    if (cs.h) (void)hipHostFree(cs.h);
    cs.h = nullptr; cs.cap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&cs.h), need + need / 2 + 256, hipHostMallocDefault));
    cs.cap = need + need / 2 + 256;
  }
  if (!cs.done) HIP_TRY(hipEventCreateWithFlags(&cs.done, hipEventDisableTiming));
  RaCarEv *flat = reinterpret_cast<RaCarEv *>(cs.h);
  int32_t *begin = reinterpret_cast<int32_t *>(cs.h + total * sizeof(RaCarEv));
  size_t at = 0;
  for (int s = 0; s < S; ++s) {
    begin[s] = (int32_t)at;
    auto &q = c->car_pending[s];
    while (!q.empty() && q.front().sample < t0 + n) { flat[at++] = q.front(); q.pop_front(); }
  }
  begin[S] = (int32_t)at;
  cs.n_ev = at;
  return RA_OK;
}

// The block is cut into time chunks; analysis, tracking, propagation and render of successive chunks run on four HIP
// streams chained by events.  Tracking and propagation are dependency chains over frames that occupy a few hundred
// wavefronts; cutting the block lets the throughput kernels (analysis of later chunks, render of earlier ones) fill the
// rest of the chip meanwhile.  Results do not depend on the cut (each chunk continues the carried state exactly).
int process_block_impl(ra_context *c, const float *d_voice, const float *d_carrier, int64_t n, float *d_out_final, double *d_out_d_caller, bool async, bool &started,
                       bool apply_post, bool defer_history) {
  if (n <= 0) return RA_OK;
  HIP_TRY(hipSetDevice(c->device));
  RaParams &p = c->p;
  const int64_t t0 = c->t;
  int64_t f_lo, nfb;
  frames_in_block(p, t0, n, f_lo, nfb);
  int n_chunks = 1;
  // ~64 frames per chunk (measured on configs[1] with overlapped blocks: 5 chunks 20.96, 6: 20.55, 7: 20.45, 8: 20.61, 9: 20.99, 10: 21.36 ms per step)
  if (c->cfg.reference_order != 1 && nfb >= 64) n_chunks = static_cast<int>(std::min<int64_t>(kMaxChunks, std::max<int64_t>(3, nfb / 64)));
  static const int forced_chunks = std::getenv("RA_PIPELINE_CHUNKS") ? std::atoi(std::getenv("RA_PIPELINE_CHUNKS")) : 0; // tuning aid
  if (forced_chunks > 0 && c->cfg.reference_order != 1 && nfb >= forced_chunks) n_chunks = std::min(forced_chunks, kMaxChunks);
  // blocks that cannot overlap their predecessor: single-chunk / reference-order / feedback blocks (they run on the caller's stream): they
  // first wait for whatever is still in flight.  The vocoder can: all of its work (carrier synth, transforms or filter bank, mix, history
  // roll) is queued on the render stream, in block order, and its single-buffered scratch and mix are only touched there
  int r;
  if (p.fb_gain != 0.f) n_chunks = static_cast<int>(std::min<int64_t>(nfb, nfb / ((int64_t)p.fb_sz / p.H + 1) + 1)) + 2; // pieces (only sizes the hop-state buffer)
  const bool caller_stream_only = n_chunks == 1 || p.fb_gain != 0.f;
  if (caller_stream_only) { if ((r = join_streams(c))) return r; async = false; }
  c->parity ^= 1; // (ensure_scratch binds the set `parity` names; a failure there flips it back: nothing has been launched yet)
  if ((r = ensure_scratch(c, nfb, n, n_chunks))) { c->parity ^= 1; return r; }
  // the vocoder's carrier: the caller's buffer, or — when there is none and the context has a carrier synth — that synth's output
  const bool car_gen = p.car_P > 0 && !d_carrier;
  if (car_gen && (r = stage_carrier_events(c, t0, n))) { c->parity ^= 1; return r; }
  started = true;
  c->last_car_gen = car_gen;
  const float *car_f = car_gen ? c->b.car_gen_f : d_carrier;
  const double *car_d = car_gen ? c->b.car_gen_d : nullptr;
  ra_context::Scratch &sc = c->scr[c->parity];
  c->b.dbg_sqmag = c->dbg_sqmag;
  c->b.dbg_clk = c->dbg_clk;
  c->b.dbg_stamp = c->dbg_stamp;
  const RaBuffers &b = c->b;
  hipStream_t user = c->stream;
  // post "limit": the render leaves the double mix in post_d and the limiter kernel produces the float output (and the mono publication)
  const bool limit = p.post_limit && apply_post && d_out_final;
  float *d_out = limit ? nullptr : d_out_final;
  double *d_out_d = limit ? b.post_d : d_out_d_caller;
  hipError_t car_copy_err = hipSuccess;
  auto launch_carrier = [&](hipStream_t st) { // K6: the carrier synth (when it supplies the carrier)
    ra_context::CarStage &cs = c->car_stage[c->parity];
    if (cs.n_ev) car_copy_err = hipMemcpyAsync(b.car_ev, cs.h, cs.n_ev * sizeof(RaCarEv), hipMemcpyHostToDevice, st);
    if (car_copy_err == hipSuccess)
      car_copy_err = hipMemcpyAsync(b.car_ev_begin, cs.h + cs.n_ev * sizeof(RaCarEv), (size_t)(p.n_streams + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (car_copy_err == hipSuccess) car_copy_err = hipEventRecord(cs.done, st);
    cs.used = true;
    ra_launch_carrier_synth(p, b, t0, n, c->car_total_segs, st);
  };
  auto launch_vocoder = [&](hipStream_t st) { // K6 then K4 / K7 on one queue
    if (car_gen) launch_carrier(st);
    ra_launch_vocoder(p, b, d_voice, car_f, car_d, t0, n, st);
  };

  if (p.fb_gain != 0.f) {
    // Feedback of the output into the analysis input: the analysis of a piece needs the mono output up to fb_sz + 1 frames before its
    // last frame fires, i.e. only output of earlier pieces if a piece holds at most fb_sz / H + 1 analysis frames.  The pieces run one
    // after the other on the caller's stream (no overlap: that is the price of the dependency).
    const int64_t per = (int64_t)p.fb_sz / p.H + 1;
    if (p.voc_enabled) RA_TRY(timed_launch(c, ST_VOCODER, user, [&] { launch_vocoder(user); }));
    // piece 0 holds no frame: it only renders the samples before the block's first frame fires (that frame's window looks back at them)
    int64_t fb0 = 0, fb1 = 0;
    int piece = 0;
    do {
      RaSpan sp;
      sp.t0 = t0; sp.n = n; sp.f_lo = f_lo; sp.fb0 = fb0; sp.nfb = fb1 - fb0; sp.seg0 = fb0 + piece;
      sp.ct0 = (piece == 0) ? t0 : ra_fire_sample(p.N, p.H, f_lo + fb0);
      sp.ct1 = (fb1 == nfb) ? t0 + n : ra_fire_sample(p.N, p.H, f_lo + fb1);
      if (sp.nfb > 0) {
        RA_TRY(timed_launch(c, ST_ANALYSIS, user, [&] { ra_launch_analysis(p, b, d_voice, t0, n, f_lo, sp.fb0, sp.nfb, nfb, user); }));
        RA_TRY(timed_launch(c, ST_PREP, user, [&] { ra_launch_track_prep(p, b, sp.fb0, sp.nfb, user); }));
        RA_TRY(timed_launch(c, ST_TRACK, user, [&] { ra_launch_track(p, b, f_lo, sp.fb0, sp.nfb, user); }));
      }
      if (c->cfg.reference_order == 1) RA_TRY(timed_launch(c, ST_RENDER, user, [&] { ra_launch_synth_serial(p, b, sp, d_out, d_out_d, user); }));
      else {
        RA_TRY(timed_launch(c, ST_PROPAGATE, user, [&] { ra_launch_synth_propagate(p, b, sp, user); }));
        RA_TRY(timed_launch(c, ST_RENDER, user, [&] { ra_launch_synth_render(p, b, sp, d_out, d_out_d, user); }));
      }
      if (limit) RA_TRY(timed_launch(c, ST_HISTORY, user, [&] { ra_launch_limiter(p, b, sp.ct0 - t0, sp.ct1 - t0, n, d_out_final, user); }));
      fb0 = fb1;
      fb1 = std::min(nfb, fb0 + per);
      ++piece;
    } while (fb0 < nfb);
    RA_TRY(timed_launch(c, ST_HISTORY, user, [&] { ra_launch_roll_feedback(p, b, n, user); }));
    n_chunks = 1; // the window history is rolled on the caller's stream below
  } else if (n_chunks == 1) { // everything in order on the caller's stream
    RaSpan sp{t0, n, t0, t0 + n, f_lo, 0, nfb, 0};
    RA_TRY(timed_launch(c, ST_ANALYSIS, user, [&] { ra_launch_analysis(p, b, d_voice, t0, n, f_lo, 0, nfb, nfb, user); }));
    RA_TRY(timed_launch(c, ST_PREP, user, [&] { ra_launch_track_prep(p, b, 0, nfb, user); }));
    RA_TRY(timed_launch(c, ST_TRACK, user, [&] { ra_launch_track(p, b, f_lo, 0, nfb, user); }));
    if (p.voc_enabled) RA_TRY(timed_launch(c, ST_VOCODER, user, [&] { launch_vocoder(user); }));
    if (c->cfg.reference_order == 1) RA_TRY(timed_launch(c, ST_RENDER, user, [&] { ra_launch_synth_serial(p, b, sp, d_out, d_out_d, user); }));
    else {
      RA_TRY(timed_launch(c, ST_PROPAGATE, user, [&] { ra_launch_synth_propagate(p, b, sp, user); }));
      RA_TRY(timed_launch(c, ST_RENDER, user, [&] { ra_launch_synth_render(p, b, sp, d_out, d_out_d, user); }));
    }
    if (limit) RA_TRY(timed_launch(c, ST_HISTORY, user, [&] { ra_launch_limiter(p, b, 0, n, n, d_out_final, user); }));
  } else {
    static const int prio_mode = std::getenv("RA_PIPELINE_PRIO") ? std::atoi(std::getenv("RA_PIPELINE_PRIO")) : 1;
    if (!c->pipe_ready) {
      // The dependency chain (analysis -> tracking -> propagation) must stay ahead of the renderer, which is the long pole and
      // fills whatever the chain leaves free: the chain's queues get dispatch priority over the render queue.
      int prio_least = 0, prio_greatest = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
      for (int i = 0; i < 6; ++i) {
        // RA_PIPELINE_PRIO (tuning aid): 0 = flat, 1 = chain high / render and vocoder low, 2 = tracking and propagation high, the others normal
        const int prio = prio_mode == 0 ? 0 : prio_mode == 2 ? ((i == 1 || i == 2) ? prio_greatest : 0) : (i >= 3 ? prio_least : prio_greatest);
        HIP_TRY(hipStreamCreateWithPriority(&c->pipe_stream[i], hipStreamNonBlocking, prio));
      }
      HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_voc, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_car, hipEventDisableTiming));
      for (int i = 0; i < 6; ++i) HIP_TRY(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
      for (int i = 0; i < kMaxChunks; ++i)
        for (int k = 0; k < 3; ++k) HIP_TRY(hipEventCreateWithFlags(&c->ev_chunk[k][i], hipEventDisableTiming));
      c->pipe_ready = true;
    }
    // every stage has a queue of its own (on a shared queue propagate(i+1) would wait for render(i)); the caller's stream only
    // forks into them and joins them at the end of the block
    hipStream_t sA = c->pipe_stream[0], sT = c->pipe_stream[1], sP = c->pipe_stream[2], sR = c->pipe_stream[3], sV = c->pipe_stream[4], sC = c->pipe_stream[5];
    // streams of one priority share ROCm's 4 hardware queues; with flat priorities the analysis rides on the caller's stream instead so that
    // no two stages end up behind each other on one queue
    if (prio_mode == 0) sA = user;
    static const bool one_queue_env = std::getenv("RA_PIPELINE_SERIAL") != nullptr; // tuning aid: same chunks, no overlap
    const bool one_queue = one_queue_env || c->pipeline_serial;
    if (one_queue) sA = sT = sP = sR = sV = sC = user;
    if (sA == user) { if ((r = join_streams(c))) return r; async = false; } // tuning modes that use the caller's stream cannot overlap blocks
    HIP_TRY(hipEventRecord(c->ev_fork, user));
    for (int i = 0; i < 6; ++i) HIP_TRY(hipStreamWaitEvent(c->pipe_stream[i], c->ev_fork, 0));
    // this scratch set was last used two blocks ago: its writers here wait for its readers there (normally long done)
    if (!sc.ev_track_done) {
      HIP_TRY(hipEventCreateWithFlags(&sc.ev_track_done, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&sc.ev_prop_done, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&sc.ev_render_done, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&sc.ev_voc_done, hipEventDisableTiming));
    }
    if (sc.used) {
      HIP_TRY(hipStreamWaitEvent(sA, sc.ev_track_done, 0));  // analysis rewrites the pitch lists tracking read
      HIP_TRY(hipStreamWaitEvent(sT, sc.ev_prop_done, 0));   // tracking rewrites the commands propagation read
      HIP_TRY(hipStreamWaitEvent(sP, sc.ev_render_done, 0)); // propagation rewrites the hop states the render read
      if (p.voc_enabled) HIP_TRY(hipStreamWaitEvent(sV, sc.ev_render_done, 0)); // the vocoder rewrites this set's mix, which that render added
      if (p.voc_enabled && car_gen) HIP_TRY(hipStreamWaitEvent(sC, sc.ev_voc_done, 0)); // the carrier synth rewrites this set's carrier, which that vocoder read
    }
    if (p.voc_enabled) { // independent of the analysis chain, on a queue of its own: block k + 1's vocoder runs while block k renders (the
      // mix exists per scratch set; the vocoder's own scratch is only touched on this queue, in block order).  The carrier synth — two chains
      // a few hundred waves wide that only FEED the vocoder — has a queue of its own too (r03): block k + 1's walk and render run beside
      // block k's filter bank (the generated carrier exists per scratch set; the synth's segment tables are only touched on its queue)
      if (car_gen && sC != sV) {
        RA_TRY(timed_launch(c, ST_VOCODER, sC, [&] { launch_carrier(sC); }));
        HIP_TRY(hipEventRecord(c->ev_car, sC));
        HIP_TRY(hipStreamWaitEvent(sV, c->ev_car, 0));
        RA_TRY(timed_launch(c, ST_VOCODER, sV, [&] { ra_launch_vocoder(p, b, d_voice, car_f, car_d, t0, n, sV); }));
      } else RA_TRY(timed_launch(c, ST_VOCODER, sV, [&] { launch_vocoder(sV); }));
      HIP_TRY(hipEventRecord(sc.ev_voc_done, sV));
      HIP_TRY(hipEventRecord(c->ev_voc, sV));
      if (sR != sV) HIP_TRY(hipStreamWaitEvent(sR, c->ev_voc, 0));
    }
    // Chunk boundaries (in frames).  The first chunk is short — nothing overlaps its analysis -> tracking -> propagation, so the render
    // stream starts that much sooner — and so are the last two: when the analysis of the block is over, only the chain and the render of
    // the final chunks remain and the chip drains; short final chunks make that drain short.
    int64_t bounds[kMaxChunks + 1];
    {
      static const int tail1 = std::getenv("RA_PIPELINE_TAIL1") ? std::atoi(std::getenv("RA_PIPELINE_TAIL1")) : 28; // tuning aids
      static const int tail2 = std::getenv("RA_PIPELINE_TAIL2") ? std::atoi(std::getenv("RA_PIPELINE_TAIL2")) : 12;
      bounds[0] = 0; bounds[n_chunks] = nfb;
      if (n_chunks >= 5) {
        const int64_t head = std::min<int64_t>(16, nfb / (2 * n_chunks));
        const int64_t t2 = std::min<int64_t>(tail2, nfb / (2 * n_chunks)), t1 = std::min<int64_t>(tail1, nfb / n_chunks);
        const int64_t mid = nfb - head - t1 - t2;
        bounds[1] = head;
        for (int i = 2; i <= n_chunks - 2; ++i) bounds[i] = head + mid * (i - 1) / (n_chunks - 3);
        bounds[n_chunks - 1] = nfb - t2;
      } else {
        for (int i = 1; i < n_chunks; ++i) bounds[i] = nfb * i / n_chunks;
      }
    }
    auto cut = [&](int i) -> int64_t { return bounds[i]; };
    auto span_of = [&](int i) {
      const int64_t fb0 = cut(i), fb1 = (i == n_chunks - 1) ? nfb : cut(i + 1);
      RaSpan sp;
      sp.t0 = t0; sp.n = n; sp.f_lo = f_lo; sp.fb0 = fb0; sp.nfb = fb1 - fb0; sp.seg0 = fb0 + i;
      sp.ct0 = (i == 0) ? t0 : ra_fire_sample(p.N, p.H, f_lo + fb0);
      sp.ct1 = (i == n_chunks - 1) ? t0 + n : ra_fire_sample(p.N, p.H, f_lo + fb1);
      return sp;
    };
    // Queue order.  An analysis workgroup needs four wave slots and a third of a CU's LDS at once, so beside a chip full of single-wave
    // render workgroups it gets few slots and its chunks stretch from 0.6 to ~2.8 ms.  Measured on configs[1] (ms per step):
    //   0  every stage on its own queue, all concurrent                                   23.8 - 24.7   <- default
    //   1  analysis and render share one queue, analysis `lead` chunks ahead (A0 A1 R0 A2 R1 ...)  26.3
    //   2  concurrent queues, but the render waits for the last analysis chunk              25.1
    // i.e. overlapping the two throughput kernels beats giving each the chip to itself, although both then run slower.
    static const int order = std::getenv("RA_PIPELINE_ORDER") ? std::atoi(std::getenv("RA_PIPELINE_ORDER")) : 0;
    static const int lead = std::getenv("RA_PIPELINE_LEAD") ? std::max(1, std::atoi(std::getenv("RA_PIPELINE_LEAD"))) : 2;
    if (order == 1 && !one_queue) sA = sR;
    // chunks per analysis launch: the analysis queue paces the step (it shares every CU with the renderer) and a launch's first and last
    // workgroups run with the chip half empty; two chunks per launch: 14.2 -> 13.4 ms per configs[1] step (1: 14.2, 3: 13.7, 6: 13.5)
    static const int agroup = std::getenv("RA_PIPELINE_AGROUP") ? std::max(1, std::atoi(std::getenv("RA_PIPELINE_AGROUP"))) : 2;
    auto front = [&](int i) -> int { // analysis of chunk i, and the two chains that hang off it
      const RaSpan sp = span_of(i);
      // one analysis launch may cover several chunks (from the second chunk on: the short first one gets the chain going)
      const int a_first = (i == 0) ? 0 : 1 + ((i - 1) / agroup) * agroup, a_last = (i == 0) ? 0 : std::min(n_chunks - 1, a_first + agroup - 1);
      if (i == a_first) {
        const int64_t fb0 = cut(a_first), fb1 = (a_last == n_chunks - 1) ? nfb : cut(a_last + 1);
        RA_TRY(timed_launch(c, ST_ANALYSIS, sA, [&] { ra_launch_analysis(p, b, d_voice, t0, n, f_lo, fb0, fb1 - fb0, nfb, sA); }));
        RA_TRY(timed_launch(c, ST_PREP, sA, [&] { ra_launch_track_prep(p, b, fb0, fb1 - fb0, sA); }));
        HIP_TRY(hipEventRecord(c->ev_chunk[0][a_first], sA));
      }
      HIP_TRY(hipStreamWaitEvent(sT, c->ev_chunk[0][a_first], 0));
      RA_TRY(timed_launch(c, ST_TRACK, sT, [&] { ra_launch_track(p, b, f_lo, sp.fb0, sp.nfb, sT); }));
      HIP_TRY(hipEventRecord(c->ev_chunk[1][i], sT));
      HIP_TRY(hipStreamWaitEvent(sP, c->ev_chunk[1][i], 0));
      RA_TRY(timed_launch(c, ST_PROPAGATE, sP, [&] { ra_launch_synth_propagate(p, b, sp, sP); }));
      HIP_TRY(hipEventRecord(c->ev_chunk[2][i], sP));
      return RA_OK;
    };
    const int ahead = order == 1 ? std::min(lead, n_chunks) : n_chunks;
    for (int i = 0; i < ahead; ++i) if (int fr = front(i)) return fr;
    if (order == 2 && !one_queue) HIP_TRY(hipStreamWaitEvent(sR, c->ev_chunk[0][n_chunks - 1], 0));
    for (int i = 0; i < n_chunks; ++i) {
      if (i + ahead < n_chunks) if (int fr = front(i + ahead)) return fr;
      const RaSpan sp = span_of(i);
      HIP_TRY(hipStreamWaitEvent(sR, c->ev_chunk[2][i], 0));
      RA_TRY(timed_launch(c, ST_RENDER, sR, [&] { ra_launch_synth_render(p, b, sp, d_out, d_out_d, sR); }));
    }
    if (limit) RA_TRY(timed_launch(c, ST_HISTORY, sR, [&] { ra_launch_limiter(p, b, 0, n, n, d_out_final, sR); })); // behind the block's last render
    // the window history for the next block: after this block's last analysis read of it, before the next block's first (same stream)
    RA_TRY(timed_launch(c, ST_HISTORY, sA, [&] { ra_launch_roll_history(p, b, d_voice, n, sA); }));
    HIP_TRY(hipEventRecord(sc.ev_track_done, sT));
    HIP_TRY(hipEventRecord(sc.ev_prop_done, sP));
    HIP_TRY(hipEventRecord(sc.ev_render_done, sR));
    sc.used = true;
    c->pending_join = true;
    if (!async && (r = join_streams(c))) return r;
  }
  // (the shim's render-ahead blocks roll the window history later, when the block's input is complete: shim_roll)
  if (n_chunks == 1 && !defer_history) RA_TRY(timed_launch(c, ST_HISTORY, user, [&] { ra_launch_roll_history(p, b, d_voice, n, user); }));
  if (car_copy_err != hipSuccess) return fail(RA_EHIP, std::string("carrier event upload: ") + hipGetErrorString(car_copy_err));
  HIP_TRY(hipGetLastError());
  if (c->profiling) ++c->kernel_calls;
  c->last_f_lo = f_lo; c->last_nfb = nfb; c->last_n = n; c->last_t0 = t0;
  c->t += n;
  c->frames_fired += nfb;
  return RA_OK;
}

// A block either completes, or fails before it touched any stream state (bad argument, allocation failure: the call can be retried), or
// fails midway: then the context is poisoned until ra_reset (include/resynth_amd.h).
int process_block(ra_context *c, const float *d_voice, const float *d_carrier, int64_t n, float *d_out, double *d_out_d, bool async = false, bool apply_post = true,
                  bool defer_history = false) {
  if (c->poisoned) return fail(RA_EINTERNAL, "an earlier block failed midway: the streams are in an undefined state until ra_reset");
  if (c->shim.active && !defer_history && c->shim.deferred) return fail(RA_EINVAL, "ra_process_* in the middle of a hop served through ra_feed / ra_compute (feed and compute the hop to its end first)");
  bool started = false;
  const int r = process_block_impl(c, d_voice, d_carrier, n, d_out, d_out_d, async, started, apply_post, defer_history);
  if (r && started) c->poisoned = true;
  return r;
}

int ensure_staging(ra_context *c, int64_t n) {
  if (n <= c->st_cap) return RA_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  dev_free(c, c->st_voice); dev_free(c, c->st_carrier); dev_free(c, c->st_out); dev_free(c, c->st_out_d);
  size_t S = c->p.n_streams;
  int r;
  c->st_cap = n;
  if ((r = dev_alloc(c, &c->st_voice, S * n, false))) return r;
  if ((r = dev_alloc(c, &c->st_carrier, S * n, false))) return r;
  if ((r = dev_alloc(c, &c->st_out, S * n * 2, false))) return r;
  if ((r = dev_alloc(c, &c->st_out_d, S * n * 2, false))) return r;
  return RA_OK;
}

} // namespace

extern "C" {

const char *ra_last_error(void) { return g_last_error.c_str(); }

int ra_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ra_default_config(ra_config *c) { // source/rt.resynth.lib.cpp:943-999
  if (!c) return fail(RA_EINVAL, "null config");
  std::memset(c, 0, sizeof(*c));
  c->sample_rate = 48000;
  c->window_size = 4096;
  c->stride = 1024;
  c->n_channels = 64;
  c->window_kind = RA_WINDOW_GAUSSIAN4;
  c->min_volume = 0.0001f;
  c->nearby_distance_tones = 0.4f;
  c->max_track_pitches = 1.f;
  c->autotune_tolerance_pitches = 100.f;
  c->stereo_spread = 1.f;
  c->env_sustain_level = 1.f;
  c->autotune_type = RA_AUTOTUNE_SCALE;
  c->autotune_max_pitch = 150;
  c->autotune_factor = 2;
  c->autotune_scale_mode = RA_SCALE_MAJOR;
  c->autotune_chord_frequencies = RA_CHORD_HARMONICS;
  c->autotune_bit_chord = 0b10010001;
  c->vocoder_modulator_window_size_seconds = 0.10f;
  c->vocoder_stride_seconds = 0.005f;
  c->vocoder_count_bands = 5;
  c->vocoder_min_freq = 100.f;
  c->vocoder_max_freq = 20000.f;
  c->voice_volume = 0.f;
  c->carrier_volume = 0.1f;
  c->vocoder_volume = 0.f;
  c->analysis_volume = 0.f;
  c->analysis_input_gain = 1.f;
  c->phase_policy = RA_PHASE_HASH;
  c->seed = 0;
  c->max_pitches_per_frame = 512;
  c->analysis_output_feedback_gain = 0.f;
  c->output_delay_seconds = 1.f;
  c->vocoder_kind = RA_VOCODER_FFT;
  c->vocoder_env_follower_cutoff_ratio = 1.f / 20.f;
  c->carrier_n_channels = 0;
  c->vocoder_carrier_square_volume = 1.f;
  c->vocoder_carrier_pulse_width = 0.01f;
  return RA_OK;
}

int ra_even_window_size_frames(float window_size_seconds, int sample_rate) { // lib.cpp:484-487
  return 2 * std::max(1, sample_rate ? static_cast<int>(0.5 * window_size_seconds * sample_rate) : 0);
}
int ra_window_center_stride_frames(float stride_seconds, int sample_rate) { // lib.cpp:1577-1578
  return std::max(1, static_cast<int>(0.5f + stride_seconds * sample_rate));
}

int ra_create(const ra_config *cfg, int32_t n_streams, int32_t device, ra_context **out) {
  if (!cfg || !out) return fail(RA_EINVAL, "null argument");
  *out = nullptr;
  int r = validate(*cfg, n_streams);
  if (r) return r;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RA_ENODEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RA_EINVAL, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  ra_context *c = new ra_context();
  c->cfg = *cfg;
  c->device = device;
  r = build_tables(c);
  c->p.n_streams = n_streams;
  // per-workgroup LDS of the kernels whose footprint grows with the configuration: refuse here, with the numbers, rather than at a launch
  if (!r) {
    const size_t lim = RA_LDS_LIMIT_BYTES, la = ra_analysis_lds_bytes(c->p), lt = ra_track_lds_bytes(c->p);
    if (la > lim)
      r = fail(RA_EINVAL, "window_size / max_pitches_per_frame need " + std::to_string(la) + " bytes of LDS per analysis workgroup (limit " +
                              std::to_string(lim) + "): lower max_pitches_per_frame" + ((cfg->pitch_harmonize_pre_autotune != 0.f || cfg->pitch_harmonize_post_autotune != 0.f) ? " (harmonize doubles the list capacity)" : ""));
    else if (lt > lim)
      r = fail(RA_EINVAL, "max_pitches_per_frame / n_channels need " + std::to_string(lt) + " bytes of LDS per tracking workgroup (limit " + std::to_string(lim) + ")");
  }
  if (!r) r = alloc_state(c);
  if (!r) r = init_state(c);
  if (!r && std::getenv("RA_PHASE_CLOCKS")) {
    r = dev_alloc(c, &c->dbg_clk, 128, true);
  }
  if (r) { std::string keep = g_last_error; ra_destroy(c); g_last_error = keep; return r; }
  *out = c;
  return RA_OK;
}

void ra_destroy(ra_context *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize(); // asynchronous blocks may still be running on the internal streams
  if (c->dbg_clk) { // tuning aid: ticks of the 100 MHz s_memtime counter per kernel phase, summed over workgroups
    unsigned long long h[128];
    if (hipMemcpy(h, c->dbg_clk, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      std::fprintf(stderr, "[resynth_amd] phase clocks:");
      for (int i = 0; i < 128; ++i) if (h[i]) std::fprintf(stderr, " [%d]=%llu", i, h[i]);
      std::fprintf(stderr, "\n");
    }
  }
  for (void *p : c->allocs) (void)hipFree(p);
  if (c->shim.h_in) (void)hipHostFree(c->shim.h_in);
  if (c->shim.h_out) (void)hipHostFree(c->shim.h_out);
  if (c->shim.h_flags) (void)hipHostFree(c->shim.h_flags);
  for (auto &cs : c->car_stage) { if (cs.h) (void)hipHostFree(cs.h); if (cs.done) (void)hipEventDestroy(cs.done); }
  for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto &sp : c->spare) { (void)hipEventDestroy(sp.first); (void)hipEventDestroy(sp.second); }
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->pipe_ready) {
    for (int i = 0; i < 6; ++i) { (void)hipStreamSynchronize(c->pipe_stream[i]); (void)hipStreamDestroy(c->pipe_stream[i]); }
    (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_voc); (void)hipEventDestroy(c->ev_car);
    for (auto e : c->ev_join) (void)hipEventDestroy(e);
    for (auto &row : c->ev_chunk) for (auto e : row) if (e) (void)hipEventDestroy(e);
  }
  for (auto &sc : c->scr) for (hipEvent_t e : {sc.ev_track_done, sc.ev_prop_done, sc.ev_render_done, sc.ev_voc_done}) if (e) (void)hipEventDestroy(e);
  if (c->hp.s_in) {
    (void)hipStreamDestroy(c->hp.s_in); (void)hipStreamDestroy(c->hp.s_out);
    for (int q = 0; q < 2; ++q) {
      (void)hipEventDestroy(c->hp.ev_in[q]); (void)hipEventDestroy(c->hp.ev_out_done[q]); (void)hipEventDestroy(c->hp.ev_user[q]);
      for (int i = 0; i < 6; ++i) (void)hipEventDestroy(c->hp.ev_pipe[q][i]);
    }
  }
  delete c;
}

int ra_reset(ra_context *c) {
  if (!c) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  if (c->poisoned) HIP_TRY(hipDeviceSynchronize()); // whatever the failed block still had in flight
  { int jr = join_streams(c); if (jr) return jr; }
  const int r = init_state(c);
  if (!r) c->poisoned = false;
  return r;
}

int ra_set_hip_stream(ra_context *c, void *hip_stream) {
  if (!c) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->stream = static_cast<hipStream_t>(hip_stream);
  return RA_OK;
}

int ra_set_profiling(ra_context *c, int32_t enable) {
  if (!c) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  int r = resolve_timing(c);
  if (r) return r;
  c->profiling = enable != 0;
  for (int i = 0; i < ST_COUNT; ++i) { c->kernel_ms[i] = 0; c->kernel_launches[i] = 0; }
  c->kernel_calls = 0;
  return RA_OK;
}

int ra_set_clock_stamps(ra_context *c, int32_t on) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (on && !RA_CLOCK_STAMPS) return fail(RA_EINVAL, "clock stamps need the diagnostic build (make -C cpp_audio_amd/csrc clk; RESYNTH_AMD_LIB=.../libresynth_amd_clk.so)");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (on && !c->dbg_stamp) RA_TRY(dev_alloc(c, &c->dbg_stamp, 16, true));
  if (c->dbg_stamp) HIP_TRY(hipMemsetAsync(c->dbg_stamp, 0, 16 * sizeof(unsigned long long), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!on && c->dbg_stamp) { dev_free(c, c->dbg_stamp); c->dbg_stamp = nullptr; }
  return RA_OK;
}

int ra_get_clock_stamps(ra_context *c, double *ghz2, int64_t *workgroups2) {
  if (!c || !ghz2 || !workgroups2) return fail(RA_EINVAL, "null argument");
  if (!c->dbg_stamp) return fail(RA_EINVAL, "clock stamps are off (ra_set_clock_stamps)");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  unsigned long long h[16];
  HIP_TRY(hipMemcpy(h, c->dbg_stamp, sizeof(h), hipMemcpyDeviceToHost));
  for (int k = 0; k < 2; ++k) { // slots 0..2: analysis, 4..6: render — shader ticks, 100 MHz ticks, workgroups
    ghz2[k] = h[4 * k + 1] ? 0.1 * (double)h[4 * k] / (double)h[4 * k + 1] : 0.0;
    workgroups2[k] = (int64_t)h[4 * k + 2];
  }
  return RA_OK;
}

int ra_set_pipeline_serial(ra_context *c, int32_t on) {
  if (!c) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  c->pipeline_serial = on != 0;
  return RA_OK;
}

int ra_get_kernel_ms(ra_context *c, double *ms6, int64_t *calls) {
  if (!c || !ms6) return fail(RA_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  int r = resolve_timing(c);
  if (r) return r;
  for (int i = 0; i < 6; ++i) ms6[i] = c->kernel_ms[i];
  if (calls) *calls = c->kernel_calls;
  return RA_OK;
}

int ra_get_kernel_launches(ra_context *c, int64_t *launches6) {
  if (!c || !launches6) return fail(RA_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  int r = resolve_timing(c);
  if (r) return r;
  for (int i = 0; i < 6; ++i) launches6[i] = c->kernel_launches[i];
  return RA_OK;
}

int ra_get_stage_timing(ra_context *c, int32_t cap, double *ms, int64_t *launches, int64_t *calls) {
  if (!c || cap < 0 || (cap > 0 && (!ms || !launches))) return fail(RA_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  int r = resolve_timing(c);
  if (r) return r;
  for (int i = 0; i < ST_COUNT && i < cap; ++i) { ms[i] = c->kernel_ms[i]; launches[i] = c->kernel_launches[i]; }
  if (calls) *calls = c->kernel_calls;
  return ST_COUNT;
}

int ra_process_device(ra_context *c, const float *d_voice, const float *d_carrier, int64_t n_frames, float *d_out) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (n_frames < 0) return fail(RA_EINVAL, "n_frames < 0");
  if (!d_out && n_frames > 0) return fail(RA_EINVAL, "null output");
  return process_block(c, d_voice, d_carrier, n_frames, d_out, nullptr);
}

int ra_process_device_async(ra_context *c, const float *d_voice, const float *d_carrier, int64_t n_frames, float *d_out) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (n_frames < 0) return fail(RA_EINVAL, "n_frames < 0");
  if (!d_out && n_frames > 0) return fail(RA_EINVAL, "null output");
  return process_block(c, d_voice, d_carrier, n_frames, d_out, nullptr, true);
}

int ra_join(ra_context *c) {
  if (!c) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  return join_streams(c);
}

int ra_synchronize(ra_context *c) {
  if (!c) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  int32_t flags = 0;
  HIP_TRY(hipMemcpy(&flags, c->b.error_flags, sizeof(flags), hipMemcpyDeviceToHost));
  if (flags & RA_ERR_OVERFLOW) return fail(RA_EOVERFLOW, "a per-frame pitch list exceeded max_pitches_per_frame");
  if (flags & RA_ERR_BOOKKEEPING) return fail(RA_EINTERNAL, "channel bookkeeping mismatch between tracking and synthesis");
  return RA_OK;
}

int ra_process_host(ra_context *c, const float *h_voice, const float *h_carrier, int64_t n, float *h_out) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (n < 0) return fail(RA_EINVAL, "n_frames < 0");
  if (n == 0) return RA_OK;
  if (!h_out) return fail(RA_EINVAL, "null output");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  int r = ensure_staging(c, n);
  if (r) return r;
  size_t S = c->p.n_streams;
  if (h_voice) HIP_TRY(hipMemcpyAsync(c->st_voice, h_voice, S * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  if (h_carrier) HIP_TRY(hipMemcpyAsync(c->st_carrier, h_carrier, S * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  r = process_block(c, h_voice ? c->st_voice : nullptr, h_carrier ? c->st_carrier : nullptr, n, c->st_out, nullptr);
  if (r) return r;
  HIP_TRY(hipMemcpyAsync(h_out, c->st_out, S * n * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  return ra_synchronize(c);
}

// The same block, cut into chunks of `chunk_frames` that are copied in, processed and copied out on three queues: chunk k + 1's input travels
// and chunk k - 1's output returns while chunk k runs (and, blocks being issued asynchronously, chunk k + 1's analysis overlaps chunk k's render).
int ra_process_host_pipelined(ra_context *c, const float *h_voice, const float *h_carrier, int64_t n, float *h_out, int64_t chunk) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (n < 0 || chunk < 1) return fail(RA_EINVAL, "bad n_frames / chunk_frames");
  if (n == 0) return RA_OK;
  if (!h_out) return fail(RA_EINVAL, "null output");
  if (c->poisoned) return fail(RA_EINTERNAL, "an earlier block failed midway: the streams are in an undefined state until ra_reset");
  if (c->p.fb_gain != 0.f) return ra_process_host(c, h_voice, h_carrier, n, h_out); // (feedback blocks run in sequential pieces anyway)
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  chunk = std::min(chunk, n);
  ra_context::HostPipe &hp = c->hp;
  const size_t S = c->p.n_streams;
  if (!hp.s_in) {
    HIP_TRY(hipStreamCreateWithFlags(&hp.s_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&hp.s_out, hipStreamNonBlocking));
    for (int q = 0; q < 2; ++q) {
      HIP_TRY(hipEventCreateWithFlags(&hp.ev_in[q], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&hp.ev_out_done[q], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&hp.ev_user[q], hipEventDisableTiming));
      for (int i = 0; i < 6; ++i) HIP_TRY(hipEventCreateWithFlags(&hp.ev_pipe[q][i], hipEventDisableTiming));
    }
  }
  if (chunk > hp.cap) {
    HIP_TRY(hipDeviceSynchronize());
    for (int q = 0; q < 2; ++q) { dev_free(c, hp.voice[q]); dev_free(c, hp.carrier[q]); dev_free(c, hp.out[q]); }
    hp.cap = 0;
    for (int q = 0; q < 2; ++q) {
      RA_TRY(dev_alloc(c, &hp.voice[q], S * chunk, false));
      RA_TRY(dev_alloc(c, &hp.carrier[q], S * chunk, false));
      RA_TRY(dev_alloc(c, &hp.out[q], S * chunk * 2, false));
    }
    hp.cap = chunk;
  }
  int k = 0;
  bool used[2] = {false, false};
  for (int64_t t = 0; t < n; t += chunk, ++k) {
    const int q = k & 1;
    const int64_t m = std::min(chunk, n - t);
    // copy-in: the set's previous chunk must have left it (its output copied out = all of its work done)
    if (used[q]) HIP_TRY(hipStreamWaitEvent(hp.s_in, hp.ev_out_done[q], 0));
    if (h_voice) HIP_TRY(hipMemcpy2DAsync(hp.voice[q], (size_t)m * sizeof(float), h_voice + t, (size_t)n * sizeof(float), (size_t)m * sizeof(float), S, hipMemcpyHostToDevice, hp.s_in));
    if (h_carrier) HIP_TRY(hipMemcpy2DAsync(hp.carrier[q], (size_t)m * sizeof(float), h_carrier + t, (size_t)n * sizeof(float), (size_t)m * sizeof(float), S, hipMemcpyHostToDevice, hp.s_in));
    HIP_TRY(hipEventRecord(hp.ev_in[q], hp.s_in));
    HIP_TRY(hipStreamWaitEvent(c->stream, hp.ev_in[q], 0));
    RA_TRY(process_block(c, h_voice ? hp.voice[q] : nullptr, h_carrier ? hp.carrier[q] : nullptr, m, hp.out[q], nullptr, true));
    // copy-out: behind everything the block put on the caller's stream and on the internal queues (without making the caller's stream wait)
    HIP_TRY(hipEventRecord(hp.ev_user[q], c->stream));
    HIP_TRY(hipStreamWaitEvent(hp.s_out, hp.ev_user[q], 0));
    if (c->pipe_ready)
      for (int i = 0; i < 6; ++i) {
        HIP_TRY(hipEventRecord(hp.ev_pipe[q][i], c->pipe_stream[i]));
        HIP_TRY(hipStreamWaitEvent(hp.s_out, hp.ev_pipe[q][i], 0));
      }
    HIP_TRY(hipMemcpy2DAsync(h_out + 2 * t, (size_t)n * 2 * sizeof(float), hp.out[q], (size_t)m * 2 * sizeof(float), (size_t)m * 2 * sizeof(float), S, hipMemcpyDeviceToHost, hp.s_out));
    HIP_TRY(hipEventRecord(hp.ev_out_done[q], hp.s_out));
    used[q] = true;
  }
  HIP_TRY(hipStreamSynchronize(hp.s_out));
  return ra_synchronize(c);
}

void *ra_host_alloc(int64_t bytes) {
  void *p = nullptr;
  if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { fail(RA_EHIP, "ra_host_alloc: hipHostMalloc failed"); return nullptr; }
  return p;
}
void ra_host_free(void *p) { if (p) (void)hipHostFree(p); }

int ra_carrier_events(ra_context *c, const ra_carrier_event *ev, int64_t n) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (n < 0 || (n > 0 && !ev)) return fail(RA_EINVAL, "bad events");
  if (n > 0 && c->cfg.carrier_n_channels <= 0) return fail(RA_EINVAL, "the context has no carrier synth (carrier_n_channels = 0)");
  if (c->p.car_P <= 0) return RA_OK; // every vocoder volume is 0: the carrier synth is inaudible and not run
  for (int64_t i = 0; i < n; ++i) {
    const ra_carrier_event &e = ev[i];
    const std::string at = "carrier event " + std::to_string(i) + ": ";
    if (e.stream < 0 || e.stream >= c->p.n_streams) return fail(RA_EINVAL, at + "stream out of range");
    if (e.type != 0 && e.type != 1) return fail(RA_EINVAL, at + "type must be 0 (NoteOn) or 1 (NoteOff)");
    if (e.delay < 0) return fail(RA_EINVAL, at + "negative delay");
    if (e.sample < c->t) return fail(RA_EINVAL, at + "its frame has already been processed");
    if (e.sample < c->car_last_sample[e.stream]) return fail(RA_EINVAL, at + "samples of a stream must not decrease");
    ra_context::CarNoteIds &g = c->car_noteids[e.stream];
    const uint64_t key = static_cast<uint64_t>(static_cast<uint32_t>(e.key));
    RaCarEv d;
    d.sample = e.sample; d.type = e.type; d.delay = e.delay; d.vel = e.velocity; d.freq = 0.f;
    if (e.type == 0) { // mkNoteOn(NoteOnId(key), midi_pitch_to_freq(key), velocity), lib.cpp:1527-1531
      if (!(e.velocity > 0.f)) return fail(RA_EINVAL, at + "a note-on needs a velocity > 0 (gen.crtp.h:391)");
      d.noteid = ++g.next;
      g.ids[key].push_back(d.noteid);
      d.freq = e.frequency > 0.f ? e.frequency
                                 : static_cast<float>(440.f * std::pow(std::pow(2., 1.0 / kNumHalftones), static_cast<double>(e.key) - 69.0)); // midi.h:258-264
    } else { // NoteOffId: the earliest pending note-on of the key, events.h:255-268
      auto it = g.ids.find(key);
      if (it == g.ids.end()) return fail(RA_EINVAL, at + "note-off for a key with no pending note-on");
      d.noteid = it->second.front();
      if (it->second.size() == 1) g.ids.erase(it); else it->second.pop_front();
    }
    c->car_pending[e.stream].push_back(d);
    c->car_last_sample[e.stream] = e.sample;
  }
  return RA_OK;
}

int ra_get_carrier(ra_context *c, int32_t stream, double *h_carrier, int64_t cap) {
  if (!c || !h_carrier) return fail(RA_EINVAL, "null argument");
  if (stream < 0 || stream >= c->p.n_streams) return fail(RA_EINVAL, "stream out of range");
  if (!c->last_car_gen || !c->b.car_gen_d) return fail(RA_EINVAL, "the last block's carrier did not come from the carrier synth");
  if (cap < c->last_n) return fail(RA_EINVAL, "buffer too small");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(h_carrier, c->b.car_gen_d + static_cast<size_t>(stream) * c->last_n, static_cast<size_t>(c->last_n) * sizeof(double), hipMemcpyDeviceToHost));
  return RA_OK;
}

namespace {
void shim_init(ra_context *c) {
  ra_context::Shim &sh = c->shim;
  if (sh.active) return;
  const int S = c->p.n_streams;
  sh.in_voice.assign(S, {}); sh.in_carrier.assign(S, {}); sh.carrier_given.assign(S, 0); sh.ring.assign(S, {}); sh.ring_head.assign(S, 0); sh.served.assign(S, c->t);
  sh.in_base = sh.rendered = c->t;
  sh.deferred = false;
  sh.active = true;
}
int shim_pinned(ra_context *c, int64_t floats_in, int64_t doubles_out) {
  ra_context::Shim &sh = c->shim;
  if (floats_in > sh.h_in_cap) {
    if (sh.h_in) (void)hipHostFree(sh.h_in);
    sh.h_in = nullptr; sh.h_in_cap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sh.h_in), sizeof(float) * (size_t)(floats_in + floats_in / 2), hipHostMallocDefault));
    sh.h_in_cap = floats_in + floats_in / 2;
  }
  if (doubles_out > sh.h_out_cap) {
    if (sh.h_out) (void)hipHostFree(sh.h_out);
    sh.h_out = nullptr; sh.h_out_cap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sh.h_out), sizeof(double) * (size_t)(doubles_out + doubles_out / 2), hipHostMallocDefault));
    sh.h_out_cap = doubles_out + doubles_out / 2;
  }
  return RA_OK;
}
// the history roll a render-ahead block owes, once every stream has fed the block's input
int shim_roll(ra_context *c) {
  ra_context::Shim &sh = c->shim;
  if (!sh.deferred) return RA_OK;
  const int S = c->p.n_streams;
  const int64_t n = sh.def_n;
  for (int s = 0; s < S; ++s)
    if ((int64_t)sh.in_voice[s].size() < n) return fail(RA_EINVAL, "ra_compute: stream " + std::to_string(s) + " has not been fed up to the frames the others are computing");
  RA_TRY(ensure_staging(c, n));
  RA_TRY(shim_pinned(c, 3 * (int64_t)S * n, 0));
  // (the roll's image has its own third of the pinned buffer and of the device staging — st_out is idle on this path — so nothing waits for it)
  float *hr = sh.h_in + 2 * (size_t)S * n;
  for (int s = 0; s < S; ++s) std::memcpy(hr + (size_t)s * n, sh.in_voice[s].data(), sizeof(float) * (size_t)n);
  HIP_TRY(hipMemcpyAsync(c->st_out, hr, sizeof(float) * (size_t)S * n, hipMemcpyHostToDevice, c->stream));
  ra_launch_roll_history(c->p, c->b, c->st_out, n, c->stream);
  HIP_TRY(hipGetLastError());
  for (int s = 0; s < S; ++s) {
    sh.in_voice[s].erase(sh.in_voice[s].begin(), sh.in_voice[s].begin() + n);
    sh.in_carrier[s].erase(sh.in_carrier[s].begin(), sh.in_carrier[s].begin() + std::min<size_t>(n, sh.in_carrier[s].size()));
    sh.carrier_given[s] = 0;
  }
  sh.in_base += n;
  sh.deferred = false;
  return RA_OK;
}
// renders the next block for every stream: at least up to frame `upto` (exclusive) of the caller's compute
int shim_render(ra_context *c, int64_t upto) {
  ra_context::Shim &sh = c->shim;
  const RaParams &p = c->p;
  const int S = p.n_streams;
  RA_TRY(shim_roll(c));
  const int64_t R = sh.rendered;
  const bool ahead = !p.voc_enabled && p.fb_gain == 0.f; // nothing but the analysis' events reaches the output
  int64_t X, need_in;
  if (ahead) {
    const int64_t k0 = ra_fire_sample(p.N, p.H, 0);
    if (R < k0) { X = k0; need_in = 0; }                        // before the first frame fires: silence up to it
    else {                                                       // R is a fire sample: up to the next one; the frame needs input sample R itself
      const int64_t f = (R - k0) / p.H;
      X = ra_fire_sample(p.N, p.H, f + 1);
      need_in = (ra_fire_sample(p.N, p.H, f) == R) ? 1 : 0;
      if (ra_fire_sample(p.N, p.H, f) != R) { X = upto; need_in = X - R; } // (not aligned: a context that was driven by ra_process_* before)
    }
    X = std::min<int64_t>(X, R + ((int64_t)1 << 20));
  } else { X = upto; need_in = X - R; }                          // a vocoder or a feedback loop needs every input sample: block by block as requested
  const int64_t n = X - R;
  for (int s = 0; s < S; ++s)
    if ((int64_t)sh.in_voice[s].size() < need_in) return fail(RA_EINVAL, "ra_compute: stream " + std::to_string(s) + " has not been fed up to the frames the others are computing");
  RA_TRY(ensure_staging(c, n));
  RA_TRY(shim_pinned(c, 3 * (int64_t)S * n, 2 * (int64_t)S * n));
  bool any_carrier = false, all_carrier = true;
  for (int s = 0; s < S; ++s) { any_carrier |= sh.carrier_given[s] != 0; all_carrier &= sh.carrier_given[s] != 0; }
  const bool synth_carrier = p.car_P > 0 && !any_carrier; // no carrier was fed: the carrier synth supplies it (ra_carrier_events)
  if (p.car_P > 0 && any_carrier && !all_carrier) return fail(RA_EINVAL, "ra_feed: a block's carrier must be given for every stream or for none (none = the carrier synth plays)");
  float *hv = sh.h_in, *hc = sh.h_in + (size_t)S * n;
  for (int s = 0; s < S; ++s) {
    const int64_t have = std::min<int64_t>(n, (int64_t)sh.in_voice[s].size());
    std::memcpy(hv + (size_t)s * n, sh.in_voice[s].data(), sizeof(float) * (size_t)have);
    std::memset(hv + (size_t)s * n + have, 0, sizeof(float) * (size_t)(n - have));
    const int64_t hcn = std::min<int64_t>(n, (int64_t)sh.in_carrier[s].size());
    std::memcpy(hc + (size_t)s * n, sh.in_carrier[s].data(), sizeof(float) * (size_t)hcn);
    std::memset(hc + (size_t)s * n + hcn, 0, sizeof(float) * (size_t)(n - hcn));
  }
  HIP_TRY(hipMemcpyAsync(c->st_voice, hv, sizeof(float) * (size_t)S * n, hipMemcpyHostToDevice, c->stream));
  if (!synth_carrier) HIP_TRY(hipMemcpyAsync(c->st_carrier, hc, sizeof(float) * (size_t)S * n, hipMemcpyHostToDevice, c->stream));
  const bool defer = ahead && need_in < n;
  // a compute's contribution: not post-processed (the post-processors run in SimpleAudioOutContext::step, after every compute)
  RA_TRY(process_block(c, c->st_voice, synth_carrier ? nullptr : c->st_carrier, n, nullptr, c->st_out_d, false, false, defer));
  HIP_TRY(hipMemcpyAsync(sh.h_out, c->st_out_d, sizeof(double) * 2 * (size_t)S * n, hipMemcpyDeviceToHost, c->stream));
  if (!sh.h_flags) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sh.h_flags), 64, hipHostMallocDefault));
  HIP_TRY(hipMemcpyAsync(sh.h_flags, c->b.error_flags, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (*sh.h_flags & RA_ERR_OVERFLOW) return fail(RA_EOVERFLOW, "a per-frame pitch list exceeded max_pitches_per_frame");
  if (*sh.h_flags & RA_ERR_BOOKKEEPING) return fail(RA_EINTERNAL, "channel bookkeeping mismatch between tracking and synthesis");
  for (int s = 0; s < S; ++s) {
    std::vector<double> &ring = sh.ring[s];
    ring.erase(ring.begin(), ring.begin() + (ptrdiff_t)sh.ring_head[s]); // (what is left is at most one tile: the streams are served in step)
    sh.ring_head[s] = 0;
    ring.insert(ring.end(), sh.h_out + 2 * (size_t)s * n, sh.h_out + 2 * (size_t)(s + 1) * n);
  }
  sh.rendered = X;
  if (defer) { sh.deferred = true; sh.def_t0 = R; sh.def_n = n; }
  else {
    for (int s = 0; s < S; ++s) {
      sh.in_voice[s].erase(sh.in_voice[s].begin(), sh.in_voice[s].begin() + std::min<size_t>(n, sh.in_voice[s].size()));
      sh.in_carrier[s].erase(sh.in_carrier[s].begin(), sh.in_carrier[s].begin() + std::min<size_t>(n, sh.in_carrier[s].size()));
      sh.carrier_given[s] = 0;
    }
    sh.in_base = X;
  }
  return RA_OK;
}
} // namespace

int ra_feed(ra_context *c, int32_t stream, const float *h_voice, const float *h_carrier, int32_t n) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (stream < 0 || stream >= c->p.n_streams) return fail(RA_EINVAL, "ra_feed: stream out of range");
  if (n < 0) return fail(RA_EINVAL, "n_frames < 0");
  if (c->poisoned) return fail(RA_EINTERNAL, "an earlier block failed midway: the streams are in an undefined state until ra_reset");
  shim_init(c);
  ra_context::Shim &sh = c->shim;
  if (h_carrier && n > 0) {
    // (a carrier given from some frame of a block on: the frames before count as silence, as with the reference's queue)
    sh.in_carrier[stream].resize(sh.in_voice[stream].size(), 0.f);
    sh.carrier_given[stream] = 1;
  }
  if (h_voice) sh.in_voice[stream].insert(sh.in_voice[stream].end(), h_voice, h_voice + n);
  else sh.in_voice[stream].insert(sh.in_voice[stream].end(), (size_t)n, 0.f);
  if (h_carrier) sh.in_carrier[stream].insert(sh.in_carrier[stream].end(), h_carrier, h_carrier + n);
  return RA_OK;
}

int ra_compute(ra_context *c, int32_t stream, double *h_buffer, int32_t n) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (stream < 0 || stream >= c->p.n_streams) return fail(RA_EINVAL, "ra_compute: stream out of range");
  if (n < 1 || n > RA_N_FRAMES_PER_BUFFER) return fail(RA_EINVAL, "nFrames must be in 1..16 (max_n_frames_per_loop)");
  if (!h_buffer) return fail(RA_EINVAL, "null buffer");
  if (c->p.post_limit && c->p.fb_gain != 0.f)
    return fail(RA_EINVAL, "ra_compute with post = limit and an output feedback: the loop feeds back what was PLAYED, i.e. the limited signal, which only the caller of the "
                           "computes has (SimpleAudioOutContext::step runs the post-processors after them, lib.cpp:1245-1275) — use ra_process_*");
  shim_init(c);
  ra_context::Shim &sh = c->shim;
  const int64_t t = sh.served[stream];
  if (sh.in_base + (int64_t)sh.in_voice[stream].size() < t + n) return fail(RA_EINVAL, "ra_compute: fewer frames fed than requested");
  if (sh.rendered < t + n) { // (only a compute that has to render touches the device: the others — all but one per hop and context — are served from the host ring)
    HIP_TRY(hipSetDevice(c->device));
    while (sh.rendered < t + n) RA_TRY(shim_render(c, t + n));
  }
  const double *ring = sh.ring[stream].data() + sh.ring_head[stream];
  for (int i = 0; i < 2 * n; ++i) h_buffer[i] += ring[i]; // the reference computes accumulate (gen.crtp.h:364-369)
  sh.ring_head[stream] += 2 * (size_t)n;
  sh.served[stream] = t + n;
  return 1;
}

int ra_get_stats(ra_context *c, ra_stats *st) {
  if (!c || !st) return fail(RA_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::vector<RaTrackState> ts(c->p.n_streams);
  HIP_TRY(hipMemcpy(ts.data(), c->b.track, ts.size() * sizeof(RaTrackState), hipMemcpyDeviceToHost));
  std::memset(st, 0, sizeof(*st));
  for (auto &x : ts) st->dropped_note_on += x.dropped_note_on;
  st->frames_analysed = c->frames_fired * c->p.n_streams;
  st->samples_rendered = c->t;
  HIP_TRY(hipMemcpy(&st->error_flags, c->b.error_flags, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (c->p.car_P > 0) {
    std::vector<int32_t> dr(c->p.n_streams);
    HIP_TRY(hipMemcpy(dr.data(), c->b.car_dropped, dr.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int32_t d : dr) st->carrier_dropped_note_on += d;
  }
  return RA_OK;
}

int64_t ra_get_events(ra_context *c, int32_t stream, ra_event *h_events, int64_t cap) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (stream < 0 || stream >= c->p.n_streams) return fail(RA_EINVAL, "stream out of range");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  const RaParams &p = c->p;
  int64_t nfb = c->last_nfb;
  if (nfb == 0) return 0;
  std::vector<RaCmd> cmds(static_cast<size_t>(nfb) * p.Ppad);
  HIP_TRY(hipMemcpy(cmds.data(), c->b.cmd + static_cast<size_t>(stream) * c->b.frames_cap * p.Ppad, cmds.size() * sizeof(RaCmd), hipMemcpyDeviceToHost));
  int64_t count = 0;
  std::vector<std::pair<int, int>> order;
  for (int64_t fb = 0; fb < nfb; ++fb) {
    order.clear();
    for (int l = 0; l < p.P; ++l) {
      const RaCmd &m = cmds[fb * p.Ppad + l];
      if ((m.type & 0xff) != RA_CMD_NONE) order.push_back({m.ord, l});
    }
    std::sort(order.begin(), order.end());
    for (auto &o : order) {
      const RaCmd &m = cmds[fb * p.Ppad + o.second];
      if (count < cap && h_events) {
        ra_event e;
        e.sample = ra_fire_sample(p.N, p.H, c->last_f_lo + fb);
        e.type = (m.type & 0xff) == RA_CMD_ON ? 0 : (m.type & 0xff) == RA_CMD_OFF ? 1 : 2;
        e.channel = o.second;
        e.noteid = m.noteid;
        e.frequency = m.freq;
        e.velocity = m.vel;
        h_events[count] = e;
      }
      ++count;
    }
  }
  return count;
}

int64_t ra_get_frame_pitches(ra_context *c, int32_t stream, int64_t *h_fire, int64_t *h_offsets, int64_t frame_cap, double *h_pitch, double *h_vol,
                             int64_t item_cap) {
  if (!c) return fail(RA_EINVAL, "null context");
  if (stream < 0 || stream >= c->p.n_streams) return fail(RA_EINVAL, "stream out of range");
  HIP_TRY(hipSetDevice(c->device));
  { int jr = join_streams(c); if (jr) return jr; }
  HIP_TRY(hipStreamSynchronize(c->stream));
  const RaParams &p = c->p;
  int64_t nfb = c->last_nfb;
  if (nfb == 0) return 0;
  std::vector<int32_t> counts(nfb);
  HIP_TRY(hipMemcpy(counts.data(), c->b.pl_count + static_cast<size_t>(stream) * c->b.frames_cap, nfb * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<double> pit(static_cast<size_t>(nfb) * p.maxr), vol(static_cast<size_t>(nfb) * p.maxr);
  size_t base = static_cast<size_t>(stream) * c->b.frames_cap * p.maxr;
  HIP_TRY(hipMemcpy(pit.data(), c->b.pl_pitch + base, pit.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(vol.data(), c->b.pl_vol + base, vol.size() * sizeof(double), hipMemcpyDeviceToHost));
  int64_t pos = 0;
  for (int64_t fb = 0; fb < nfb && fb < frame_cap; ++fb) {
    h_fire[fb] = ra_fire_sample(p.N, p.H, c->last_f_lo + fb);
    h_offsets[fb] = pos;
    for (int i = 0; i < counts[fb]; ++i) {
      if (pos < item_cap) { h_pitch[pos] = pit[fb * p.maxr + i]; h_vol[pos] = vol[fb * p.maxr + i]; }
      ++pos;
    }
    h_offsets[fb + 1] = pos;
  }
  return nfb;
}

int ra_sqmag(const float *h_windows, int32_t n_windows, int32_t window_size, int32_t window_kind, int32_t device, double *h_sqmag, int32_t *fft_length) {
  if (!h_windows || !h_sqmag || n_windows < 1) return fail(RA_EINVAL, "bad argument");
  ra_config cfg;
  ra_default_config(&cfg);
  cfg.window_size = window_size;
  cfg.stride = window_size;
  cfg.window_kind = window_kind;
  cfg.carrier_volume = 0.f; // analysis only
  ra_context *c = nullptr;
  int r = ra_create(&cfg, 1, device, &c);
  if (r) return r;
  int M = c->p.M;
  if (fft_length) *fft_length = c->p.L;
  int64_t n = 2 + static_cast<int64_t>(n_windows) * window_size; // PeriodicFFT starts at input sample 2
  std::vector<float> in(n, 0.f), out(static_cast<size_t>(n) * 2);
  std::memcpy(in.data() + 2, h_windows, sizeof(float) * n_windows * window_size);
  double *dbg = nullptr;
  r = dev_alloc(c, &dbg, static_cast<size_t>(n_windows) * M);
  if (!r) {
    c->dbg_sqmag = dbg;
    r = ra_process_host(c, in.data(), nullptr, n, out.data());
  }
  if (!r) {
    if (c->last_nfb != n_windows) r = fail(RA_EINTERNAL, "ra_sqmag: unexpected frame count");
    else if (hipMemcpy(h_sqmag, dbg, sizeof(double) * n_windows * M, hipMemcpyDeviceToHost) != hipSuccess) r = fail(RA_EHIP, "copy back failed");
  }
  std::string keep = g_last_error;
  ra_destroy(c);
  g_last_error = keep;
  return r;
}

// deduceNotesSlow as main.resynth uses it (main.resynth.cpp:39-46): see include/resynth_amd.h for what is built and why.
namespace {
// linking, with track_pitches' scan (algo.cpp:256-305): open notes and peaks both ascend in pitch; one linker per signal, fed frame by frame
struct NoteLinker {
  struct Open { int32_t start, last; double last_pitch, sum_v, sum_pv, max_db; };
  std::vector<Open> open, next;
  std::vector<char> cont;
  std::vector<ra_note> notes;
  void close(const Open &o) {
    const double mean_pitch = o.sum_pv / o.sum_v;
    notes.push_back({o.start, o.last, 440.0 * std::pow(2.0, (mean_pitch - 69.0) / 12.0), o.max_db});
  }
  void frame(int64_t h, const double *pitch, const double *vol, int count) {
    next.clear();
    size_t it = 0;
    cont.assign(open.size(), 0);
    for (int i = 0; i < count; ++i) {
      const double pt = pitch[i], v = vol[i];
      if (!(v > 0.0)) continue;
      const double db = 20.0 * std::log10(v);
      int match = -1;
      for (; it != open.size(); ++it) {
        if (open[it].last_pitch < pt - 1.0) continue;
        if (open[it].last_pitch <= pt + 1.0) { match = static_cast<int>(it); ++it; }
        break;
      }
      if (match >= 0) {
        Open o = open[match];
        cont[match] = 1;
        o.last = static_cast<int32_t>(h); o.last_pitch = pt; o.sum_v += v; o.sum_pv += pt * v; o.max_db = std::max(o.max_db, db);
        next.push_back(o);
      } else next.push_back({static_cast<int32_t>(h), static_cast<int32_t>(h), pt, v, pt * v, db});
    }
    for (size_t j = 0; j < open.size(); ++j) if (!cont[j]) close(open[j]);
    std::stable_sort(next.begin(), next.end(), [](const Open &a, const Open &b) { return a.last_pitch < b.last_pitch; }); // sort_by_current_pitch, algo.cpp:434-439
    open.swap(next);
  }
  void finish() { for (const Open &o : open) close(o); open.clear(); }
};
} // namespace

// n_signals signals of n frames each through ONE context: the analysis of all of them is one launch sequence per block (blocks of at most
// 2^18 frames bound the buffers), the frame lists of a block come back in three strided copies and every signal's linker runs on a host thread.
int64_t ra_deduce_notes_batch(const float *h_signals, int32_t n_signals, int64_t n, int32_t sample_rate, int32_t half_window_size, int32_t window_kind, int32_t stride,
                              int32_t device, ra_note *h_notes, int64_t cap, int64_t *h_offsets) {
  if (!h_signals || n < 0 || n_signals < 1) return fail(RA_EINVAL, "bad signals");
  if (half_window_size < 1 || stride < 1) return fail(RA_EINVAL, "bad half window size / stride");
  if (cap < 0 || (cap > 0 && !h_notes)) return fail(RA_EINVAL, "bad notes buffer");
  ra_config cfg;
  ra_default_config(&cfg);
  cfg.sample_rate = sample_rate;
  cfg.window_size = 2 * half_window_size;
  cfg.stride = stride;
  cfg.window_kind = window_kind;
  cfg.n_channels = 2;
  cfg.min_volume = 0.f;            // every peak is kept ...
  cfg.nearby_distance_tones = 0.f; // ... as an interval of its own: the frame lists are the peaks (pitch, 10^(dB/20)) in ascending pitch
  cfg.analysis_volume = 0.f;       // nothing is synthesised
  cfg.carrier_volume = 0.f;
  cfg.max_pitches_per_frame = std::min(4096, std::max(64, fft_length_for(cfg.window_size) / 4));
  ra_context *c = nullptr;
  int r = ra_create(&cfg, n_signals, device, &c);
  if (r) return r;
  const size_t S = static_cast<size_t>(n_signals);
  // PeriodicFFT swallows input samples 0 and 1 in the offline job: two leading zeros put frame h on signal[h * stride ...)
  // block length: at most 2^18 frames, and few enough analysis frames that a block's lists (16 B per item and signal on the host) stay near 1 GB
  const int64_t total = n + 2;
  const int64_t frames_budget = std::max<int64_t>(8, (int64_t(1) << 30) / (int64_t(n_signals) * c->p.maxr * 16));
  const int64_t B = std::min<int64_t>(total, std::min<int64_t>(int64_t(1) << 18, frames_budget * stride + cfg.window_size));
  float *d_in = nullptr, *d_out = nullptr;
  std::vector<float> stage(S * static_cast<size_t>(B));
  std::vector<int32_t> counts;
  std::vector<double> pitch, vol;
  std::vector<NoteLinker> linkers(S);
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_in), S * B * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_out), S * B * 2 * sizeof(float)));
    for (int64_t t = 0; t < total; t += B) {
      const int64_t len = std::min(B, total - t);
      for (size_t s = 0; s < S; ++s) {
        float *dst = stage.data() + s * len;
        const float *src = h_signals + s * static_cast<size_t>(n);
        for (int64_t i = 0; i < len; ++i) { const int64_t a = t + i - 2; dst[i] = a >= 0 ? src[a] : 0.f; }
      }
      HIP_TRY(hipMemcpy(d_in, stage.data(), S * len * sizeof(float), hipMemcpyHostToDevice));
      RA_TRY(ra_process_device(c, d_in, nullptr, len, d_out));
      RA_TRY(ra_synchronize(c));
      const int64_t nfb = c->last_nfb, f_lo = c->last_f_lo;
      if (nfb == 0) continue;
      const size_t maxr = c->p.maxr, row = static_cast<size_t>(nfb) * maxr;
      counts.resize(S * nfb); pitch.resize(S * row); vol.resize(S * row);
      HIP_TRY(hipMemcpy2D(counts.data(), nfb * sizeof(int32_t), c->b.pl_count, c->b.frames_cap * sizeof(int32_t), nfb * sizeof(int32_t), S, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy2D(pitch.data(), row * sizeof(double), c->b.pl_pitch, c->b.frames_cap * maxr * sizeof(double), row * sizeof(double), S, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy2D(vol.data(), row * sizeof(double), c->b.pl_vol, c->b.frames_cap * maxr * sizeof(double), row * sizeof(double), S, hipMemcpyDeviceToHost));
      const int threads = static_cast<int>(std::min<size_t>(S, std::max(1u, std::thread::hardware_concurrency() / 2)));
      auto work = [&](int tix) {
        for (size_t s = tix; s < S; s += threads)
          for (int64_t fb = 0; fb < nfb; ++fb)
            linkers[s].frame(f_lo + fb, pitch.data() + s * row + fb * maxr, vol.data() + s * row + fb * maxr, counts[s * nfb + fb]);
      };
      if (threads == 1) work(0);
      else {
        std::vector<std::thread> th;
        for (int tix = 0; tix < threads; ++tix) th.emplace_back(work, tix);
        for (auto &x : th) x.join();
      }
    }
    return RA_OK;
  };
  r = body();
  std::string keep = g_last_error;
  (void)hipFree(d_in); (void)hipFree(d_out);
  ra_destroy(c);
  g_last_error = keep;
  if (r) return r;
  int64_t at = 0;
  for (size_t s = 0; s < S; ++s) {
    linkers[s].finish();
    if (h_offsets) h_offsets[s] = at;
    for (const ra_note &nt : linkers[s].notes) { if (at < cap) h_notes[at] = nt; ++at; }
  }
  if (h_offsets) h_offsets[S] = at;
  return at;
}

int64_t ra_deduce_notes(const float *h_signal, int64_t n, int32_t sample_rate, int32_t half_window_size, int32_t window_kind, int32_t stride,
                        int32_t device, ra_note *h_notes, int64_t cap) {
  if (!h_signal) return fail(RA_EINVAL, "bad signal");
  return ra_deduce_notes_batch(h_signal, 1, n, sample_rate, half_window_size, window_kind, stride, device, h_notes, cap, nullptr);
}

// One note on one driver-(A) element, starting from a given oscillator phasor.  This is what the element duck-type shim
// (cpp_audio_amd/host/resynth_amd.hpp: AmdElement) renders when the reference's manual loop first steps an element after
// forgetPastSignals / setAngleIncrements / setVolumeTarget / setAHDSR / onKeyPressed(0) / onKeyReleased(delay)
// (include/resynth.hpp:83-96).  Returns the number of frames during which the element is RT-active.
int64_t ra_render_note(double angle_increment, double volume, const ra_ahdsr *env, int32_t release_delay, int32_t next_min_change_before,
                       int32_t sample_rate, int32_t device, double *phase_re_im, double *h_out, int64_t cap) {
  if (!env || !phase_re_im) return fail(RA_EINVAL, "null argument");
  if (release_delay < 1 || sample_rate < 1000) return fail(RA_EINVAL, "bad release delay / sampling rate");
  (void)next_min_change_before; // forgetPastSignals latches the previous note's value, but every stage re-latches before using it
  RaEnvParams ep;
  { // setAHDSR, audioelement.h:923-952
    int min_dt = sample_rate / 1000;
    bool hasDecay = env->sustain < 0.999999;
    ep.SMinusOne = hasDecay ? std::min(std::max(env->sustain, 0.f), 1.f) - 1.f : 0.f;
    ep.A = std::max(env->attack, min_dt);
    ep.H = std::max(env->hold, 0);
    ep.D = hasDecay ? std::max(env->decay, min_dt) : 0;
    ep.R = std::max(env->release, min_dt);
  }
  double period = (angle_increment == 0.) ? 1. : 2. / std::fabs(angle_increment);
  int next_min_change = static_cast<int>(0.5f + 2.5 * period);
  int r_eff = std::max(next_min_change, ep.R);
  int64_t T = static_cast<int64_t>(release_delay) + r_eff + 17; // steps 0 .. delay + r_eff + 16 run with the element RT-active
  if (!h_out) return T;
  if (cap < T) return fail(RA_EINVAL, "output buffer too small");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RA_ENODEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RA_EINVAL, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  RaNoteEv e;
  e.start = 0; e.inc = angle_increment; e.vol = volume; e.delay = release_delay; e.pad = 0;
  int begin[2] = {0, 1};
  double2 ph = make_double2(phase_re_im[0], phase_re_im[1]);
  RaNoteEv *d_ev = nullptr; int *d_begin = nullptr; double *d_partial = nullptr, *d_out = nullptr; double2 *d_ph = nullptr;
  auto cleanup = [&]() { (void)hipFree(d_ev); (void)hipFree(d_begin); (void)hipFree(d_partial); (void)hipFree(d_out); (void)hipFree(d_ph); };
  hipError_t er = hipMalloc(&d_ev, sizeof(RaNoteEv));
  if (er == hipSuccess) er = hipMalloc(&d_begin, sizeof(begin));
  if (er == hipSuccess) er = hipMalloc(&d_partial, static_cast<size_t>(T) * sizeof(double));
  if (er == hipSuccess) er = hipMalloc(&d_out, static_cast<size_t>(T) * sizeof(double));
  if (er == hipSuccess) er = hipMalloc(&d_ph, sizeof(double2));
  if (er == hipSuccess) er = hipMemcpy(d_ev, &e, sizeof(e), hipMemcpyHostToDevice);
  if (er == hipSuccess) er = hipMemcpy(d_begin, begin, sizeof(begin), hipMemcpyHostToDevice);
  if (er == hipSuccess) er = hipMemcpy(d_ph, &ph, sizeof(ph), hipMemcpyHostToDevice);
  if (er == hipSuccess) { ra_launch_notes(d_ev, d_begin, 1, ep, T, d_partial, d_out, d_ph, nullptr); er = hipGetLastError(); }
  if (er == hipSuccess) er = hipMemcpy(h_out, d_out, static_cast<size_t>(T) * sizeof(double), hipMemcpyDeviceToHost);
  if (er == hipSuccess) er = hipMemcpy(&ph, d_ph, sizeof(ph), hipMemcpyDeviceToHost);
  cleanup();
  if (er != hipSuccess) return fail(RA_EHIP, std::string("ra_render_note: ") + hipGetErrorString(er));
  phase_re_im[0] = ph.x; phase_re_im[1] = ph.y;
  return T;
}

int64_t ra_resynth_notes(const ra_note *notes, int32_t n_notes, int32_t stride, int32_t sr, int32_t device, double *h_out, int64_t cap) {
  if (n_notes < 0 || (n_notes > 0 && !notes)) return fail(RA_EINVAL, "bad notes");
  if (stride < 1 || sr < 1000) return fail(RA_EINVAL, "bad stride / sampling rate");
  // setAHDSR(AHDSR{1000, LINEAR, 0, 1000, LINEAR, 10000, LINEAR, 0.7f}, sampling_rate), resynth.hpp:87-93
  RaEnvParams ep;
  {
    int min_dt = sr / 1000;
    ep.A = std::max(1000, min_dt);
    ep.H = 0;
    ep.D = std::max(1000, min_dt);
    ep.R = std::max(10000, min_dt);
    ep.SMinusOne = std::min(std::max(0.7f, 0.f), 1.f) - 1.f;
  }
  // unstride and order by start frame (resynth.hpp:56-70; equal starts keep their input order)
  struct N { int64_t start, end; double freq, amp; };
  std::vector<N> byStart(n_notes);
  for (int i = 0; i < n_notes; ++i) {
    if (notes[i].end_frame < notes[i].start_frame || notes[i].start_frame < 0) return fail(RA_EINVAL, "note with endFrame < startFrame or negative start");
    byStart[i] = {static_cast<int64_t>(notes[i].start_frame) * stride, static_cast<int64_t>(notes[i].end_frame) * stride, notes[i].frequency,
                  notes[i].amplitude_db};
  }
  std::stable_sort(byStart.begin(), byStart.end(), [](const N &a, const N &b) { return a.start < b.start; });
  // replay getAvailableOscillator (resynth.hpp:19-27): an element sounds until the step in which it reaches EnvelopeDone2
  std::vector<int64_t> slot_end;                 // last step index at which the slot's element is RT-active
  std::vector<std::vector<RaNoteEv>> per_slot;
  int64_t max_end = -1, last_start = 0;
  for (const N &nt : byStart) {
    double inc = 2 * nt.freq / sr;               // freq_to_angle_increment<double>, sound.functions.h:58-62
    double period = (inc == 0.) ? 1. : 2. / std::fabs(inc);
    int next_min_change = static_cast<int>(0.5f + 2.5 * period); // Enveloped::setAngleIncrements, audioelement.h:218-225
    int64_t delay = stride + nt.end - nt.start;
    if (delay > 0x7fffffff) return fail(RA_EINVAL, "note too long");
    int r_eff = std::max(next_min_change, ep.R);
    int64_t end_step = nt.start + delay + r_eff + 16;
    size_t slot = 0;
    for (; slot < slot_end.size(); ++slot) if (slot_end[slot] < nt.start) break;
    if (slot == slot_end.size()) { slot_end.push_back(0); per_slot.emplace_back(); }
    slot_end[slot] = end_step;
    RaNoteEv e;
    e.start = nt.start; e.inc = inc; e.vol = std::pow(10.0, nt.amp / 20.0); // DbToMag, resynth.hpp:86
    e.delay = static_cast<int32_t>(delay); e.pad = 0;
    per_slot[slot].push_back(e);
    max_end = std::max(max_end, end_step);
    last_start = nt.start;
  }
  // frames the loops of resynth.hpp:77-79,99 write: up to and including the first frame that finds no active element
  int64_t T = n_notes == 0 ? 1 : std::max(last_start, max_end + 1) + 1;
  if (!h_out) return T;
  if (cap < T) return fail(RA_EINVAL, "output buffer too small");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RA_ENODEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RA_EINVAL, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  std::vector<RaNoteEv> flat;
  std::vector<int> begin(per_slot.size() + 1, 0);
  for (size_t s = 0; s < per_slot.size(); ++s) { begin[s] = static_cast<int>(flat.size()); flat.insert(flat.end(), per_slot[s].begin(), per_slot[s].end()); }
  begin[per_slot.size()] = static_cast<int>(flat.size());
  int n_slots = static_cast<int>(per_slot.size());
  int groups = std::max(1, (n_slots + 63) / 64);
  RaNoteEv *d_ev = nullptr; int *d_begin = nullptr; double *d_partial = nullptr, *d_out = nullptr;
  auto cleanup = [&]() { (void)hipFree(d_ev); (void)hipFree(d_begin); (void)hipFree(d_partial); (void)hipFree(d_out); };
  hipError_t e = hipMalloc(&d_ev, std::max<size_t>(1, flat.size()) * sizeof(RaNoteEv));
  if (e == hipSuccess) e = hipMalloc(&d_begin, begin.size() * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&d_partial, static_cast<size_t>(groups) * T * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&d_out, static_cast<size_t>(T) * sizeof(double));
  if (e == hipSuccess && !flat.empty()) e = hipMemcpy(d_ev, flat.data(), flat.size() * sizeof(RaNoteEv), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_begin, begin.data(), begin.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    ra_launch_notes(d_ev, d_begin, n_slots, ep, T, d_partial, d_out, nullptr, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(h_out, d_out, static_cast<size_t>(T) * sizeof(double), hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(RA_EHIP, std::string("ra_resynth_notes: ") + hipGetErrorString(e));
  return T;
}

// ---- gen.voice: SWEEP, BIRDS, ROBOTS (K9) ----------------------------------------------------------------------------------------------
// The host's share of a block (ra_voice_host.hpp does the per-note work): ImplCRTP::onEvent's channel bookkeeping over
// SoundEngine::tryAcquire (soundengine.h:301-317, 470-480), the notes' ramps cut into the block's jobs, the jobs of a stream packed into
// slots whose jobs never overlap, the state slots the two kernels exchange.  Everything a block uploads goes through one pinned staging
// buffer of two (an event says when its copies have drained), so ra_voice_process_device_async returns without waiting for the device.
// A handful of host threads that share a block's per-stream work (streams are independent synths): run(n, f) calls f(i) for i in 0 .. n - 1.
// The partition is static — thread w always gets the same contiguous streams — so that what a stream allocates in one block is freed by
// the same thread in a later one (with a dynamic hand-out the frees crossed threads and the allocator's arena locks ate the gain).
struct VoicePool {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  std::function<void(int)> job;
  int n = 0, generation = 0, busy = 0;
  bool quit = false;
  explicit VoicePool(int threads) {
    for (int i = 1; i < threads; ++i) workers.emplace_back([this, i] { loop(i); });
  }
  ~VoicePool() {
    { std::lock_guard<std::mutex> g(m); quit = true; }
    cv_go.notify_all();
    for (auto &w : workers) w.join();
  }
  void share(int w) { const int T = (int)workers.size() + 1; for (int i = (int)((long long)n * w / T), e = (int)((long long)n * (w + 1) / T); i < e; ++i) job(i); }
  void loop(int w) {
    int seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> g(m); cv_go.wait(g, [&] { return quit || generation != seen; }); if (quit) return; seen = generation; }
      share(w);
      { std::lock_guard<std::mutex> g(m); if (--busy == 0) cv_done.notify_one(); }
    }
  }
  void run(int count, std::function<void(int)> f) {
    if (workers.empty() || count < 8) { for (int i = 0; i < count; ++i) f(i); return; }
    { std::lock_guard<std::mutex> g(m); job = std::move(f); n = count; busy = (int)workers.size(); ++generation; }
    cv_go.notify_all();
    share(0);
    std::unique_lock<std::mutex> g(m);
    cv_done.wait(g, [&] { return busy == 0; });
  }
};

struct ra_voice {
  ra_voice_config cfg;
  RaVoiceParams p;
  int device = 0, n_streams = 0;
  int64_t t = 0;
  hipStream_t stream = nullptr;
  float *out = nullptr;
  int64_t out_cap = 0;
  // a channel is free iff no note is on and every ramp of its engine is finished; a note-off while a ramp is busy leaves that ramp
  // unfinished for good (nothing steps the element any more): the channel is lost (isEnvelopeFinished_internal, soundengine.h:470-480)
  struct Chan { bool odd = false, lost = false; int64_t noteid = -1, t_on = 0; int live = -1; std::vector<std::pair<int64_t, int64_t>> busy; /* element steps [k0, k1) */ };
  struct LiveRamp { rav::Spec spec; int64_t k; int32_t T; int32_t carry; int32_t ramp = 0, play = 0; /* which of the engine's three ramps; which spec */ };
  struct Live { int32_t chan; bool done; int64_t t_on, noteid, limit; double fL, fR; int32_t A, R; std::vector<LiveRamp> ramps; };
  struct JobRec { RaVoiceJob j; int32_t end, live, ramp; };
  struct Synth { // one stream: everything a block's host work touches for it (so that streams can be handed to different threads)
    std::vector<ra_voice_event> pending; // queued events from `head` on (consumed by index: a worker thread never frees what the caller's thread allocated)
    size_t head = 0;
    int64_t last_sample = 0;
    int32_t seed_counter = 0;          // `thread_local int seed` of setupAudioElement (gen.voice.h:910): one per stream here
    std::vector<Chan> chans;           // [P]
    std::vector<Live> live;            // notes that still have samples to render
    int64_t dropped = 0;
    rav::Note scratch;
    // per block
    std::vector<JobRec> jobs;
    std::vector<int32_t> slot_end;
    int32_t n_states = 0, n_work = 0, slots_used = 0;
    int64_t job_base = 0, state_base = 0, work_base = 0;
  };
  std::vector<Synth> synths;
  std::unique_ptr<VoicePool> pool;
  // device side
  static constexpr int kTile = 256;
  // The state pre-pass (and the block's copies) run on a queue of their own: block k + 1's pre-pass — dependency chains a few hundred
  // waves wide — beside block k's render.  Everything a render reads exists twice (the render of block k - 1 must be over before block
  // k + 1's copies land: ev_render_done), the tile-boundary states three times (block k + 1 reads block k's and writes its own while the
  // render of block k still reads block k - 1's and block k's).
  hipStream_t pre_stream = nullptr;
  hipEvent_t ev_pre_done[2] = {nullptr, nullptr}, ev_render_done[2] = {nullptr, nullptr};
  bool set_used[2] = {false, false};
  RaVoiceJob *d_jobs[2] = {nullptr, nullptr}; int32_t *d_order[2] = {nullptr, nullptr}, *d_slot_begin[2] = {nullptr, nullptr};
  int64_t jobs_cap[2] = {0, 0}, slots_cap[2] = {0, 0}, order_cap[2] = {0, 0};
  RaVoiceState *d_states[3] = {nullptr, nullptr, nullptr};
  RaVoiceSeg *d_seg = nullptr; // (the pre-pass's own scratch: only ever touched on its queue, in block order)
  int64_t seg_cap = 0;
  int64_t states_cap[3] = {0, 0, 0};
  int parity = 0;     // which of the two sets this block uses
  int state_idx = 0;  // which of the three state arrays this block writes (the one before it holds the carry-in)
  struct Staging { void *h = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; } staging[2];
  int stage_parity = 0;
  // timing
  struct Timed { hipEvent_t a, b, b2, c; }; // pre-pass a -> b on its queue, render b2 -> c on the caller's
  std::vector<Timed> timed;
  std::vector<hipEvent_t> ev_pool;
  bool profiling = false;
  double ms[2] = {0, 0};
  int64_t launches[2] = {0, 0};
  double host_ms[6] = {0, 0, 0, 0, 0, 0}; // RA_VOICE_PROFILE: jobs, prefix, staging wait, emit, grow + copies, launches
  int64_t host_blocks = 0;
  // a block failed after its events had been consumed and channel / live-note state changed (a size limit, a staging or device allocation, a
  // HIP call before the launches): the object is then neither at the block's start nor at its end; every later block returns RA_EINTERNAL
  bool poisoned = false;
  // ---- WIND (ra_wind.cuh): the pink-noise buffer, the playing ramps' states (a slot per (stream, channel, ramp): a channel is taken again only
  // when its ramps are finished), the block's jobs, the chunk's scratch ----
  RaWindParams wp;
  double *d_pink = nullptr, *d_wfstates = nullptr, *d_wscratch = nullptr;
  RaWindState *d_wstates = nullptr;
  RaWindJob *d_wjobs = nullptr;
  int32_t *d_wstream_begin = nullptr;
  int64_t wjobs_cap = 0, wsb_cap = 0, wscratch_cap = 0;
  hipStream_t wind_stream = nullptr;                 // the controls' queue
  hipEvent_t ev_wctl[2] = {nullptr, nullptr}, ev_wchain[2] = {nullptr, nullptr}, ev_wblock = nullptr;
};

namespace {
struct VoiceProgram { const char *name; int start, pre, path, extra; int itp; float scat, length, ex, ex_scat, pause, xfade; int xfade_freq; float freq_xfade; float d1, d2, att, low, high; };
// ImplBase::getPrograms (gen.voice.h:700-761): make_bird(start, pre, path, extra, itp, freq_scat, length, exponent, exponent scatter, pause, xfade, xfade_freq,
// freq_xfade, freq_itp) — whose FREQ_TRANSITION_INTERPOLATION is written from `itp`, not `freq_itp` (:607-612), kept —, make_robot(start, pre, path, extra,
// pause, itp, freq_scat, length, exponent, exponent scatter, xfade, d1, d2, attenuation), make_sweep(itp, length, exponent, xfade, low, high)
const VoiceProgram kBirds[] = {
    {"Standard & Cute bird", 0, 0, 1, 0, RA_ITP_EASE_INOUT_CIRC, 0.f, 93.f, 2.f, .5f, 0.02267f, 0.0295f, RA_XFADE_FREQ_NO, 0.14f, 0, 0, 0, 0, 0},
    {"Scat bird", 0, 0, 3, 17, RA_ITP_EASE_INOUT_CIRC, 0.015f, 10.f, 2.f, .5f, 0.0445f, 0.0177f, RA_XFADE_FREQ_NONTRIVIAL, 0.363f, 0, 0, 0, 0, 0},
    {"Rhythmic bird", 1, 0, 3, 11, RA_ITP_EASE_INOUT_CIRC, 0.f, 19.8f, 2.f, 0.f, 0.03188f, 0.01138f, RA_XFADE_FREQ_ALL, 0.0182f, 0, 0, 0, 0, 0},
    {"Slow bird", 0, 2, 1, 0, RA_ITP_EASE_IN_EXPO, 0.f, 73.7f, 2.f, .5f, 0.02267f, 0.0295f, RA_XFADE_FREQ_NO, 0.14f, 0, 0, 0, 0, 0},
    {"BiTone bird", 1, 0, 2, 0, RA_ITP_EASE_IN_EXPO, .414f, 78.6f, 2.f, .5f, 0.09755f, 0.0295f, RA_XFADE_FREQ_NO, 0.14f, 0, 0, 0, 0, 0},
    {"Happy bird 1", 1, 0, 4, 0, RA_ITP_EASE_IN_EXPO, .414f, 78.6f, 2.f, .5f, 0.1326f, 0.0454f, RA_XFADE_FREQ_NO, 0.14f, 0, 0, 0, 0, 0},
    {"Happy bird 2", 1, 0, 4, 0, RA_ITP_EASE_IN_EXPO, .414f, 63.9f, 1.19f, 1.f, 0.1326f, 0.0454f, RA_XFADE_FREQ_NO, 0.14f, 0, 0, 0, 0, 0},
    {"Laughing bird", 1, 0, 2, 0, RA_ITP_EASE_IN_EXPO, .414f, 78.6f, 2.f, .5f, 0.2193f, 0.0295f, RA_XFADE_FREQ_ALL, 0.0725f, 0, 0, 0, 0, 0},
    {"Talkative bird", 0, 0, 6, 0, RA_ITP_EASE_INOUT_CIRC, 0.12f, 93.3f, 2.f, .5f, 0.152f, 0.05f, RA_XFADE_FREQ_NONTRIVIAL, 0.0998f, 0, 0, 0, 0, 0}};
const VoiceProgram kRobots[] = {
    {"R2D2", 0, 0, 1, 1, RA_ITP_LINEAR, 0.f, 19.8f, 2.1f, 0.39f, 0.0835f, 0.0053f, 0, 0.f, 6, 12, .98f, 0, 0},
    {"Communication", 0, 0, 16, 14, RA_ITP_EASE_INOUT_CIRC, 0.f, 10.f, 1.89f, 1.f, 0.03188f, 0.0053f, 0, 0.f, 6, 10, .98f, 0, 0}};
const VoiceProgram kSweeps[] = {
    {"Sweep 1", 0, 0, 0, 0, RA_ITP_LINEAR, 0.f, 73.f, 5.f, 0.f, 0.f, 0.0109f, 0, 0.f, 0, 0, 0, 0.0009f, 20000.f},
    {"Fullrange", 0, 0, 0, 0, RA_ITP_LINEAR, 0.f, 500.f, 5.f, 0.f, 0.f, 0.0109f, 0, 0.f, 0, 0, 0, 0.000226f, 20000.f}};

// ImplBase::getPrograms for WIND (gen.voice.h:762-806): make_noise_wind(order, width {min, max}, centre {min, max}, seconds per noise sample),
// make_sine_wind(centre, short-term ratio, seconds long, seconds short), make_mixed_wind = make_noise_wind + the sine at 0.01 (:625-697)
struct WindProgram { const char *name; int kind; int order; float w0, w1, c0, c1, sec_long, sec_short, short_ratio; }; // kind 0 noise, 1 sine, 2 mixed
const WindProgram kWinds[] = {
    {"Medium wind in trees", 0, 1, 0.f, 0.f, 1.f, 8.f, 2.268f, 0, 0},
    {"Steady wind", 0, 4, 1.3f, 1.3f, 5.2f, 5.5f, 0.09f, 0, 0},
    {"Strong wind", 0, 4, 3.8f, 3.8f, 1.f, 8.f, 2.268f, 0, 0},
    {"Vinyl cracks", 0, 89, 3.45f, 5.f, 8.1f, 8.1f, 0.000748f, 0, 0},
    {"Small animal eating", 0, 61, 0.f, 5.f, 8.1f, 8.1f, 0.000227f, 0, 0},
    {"Heavy rain in a car", 0, 33, 3.45f, 5.f, 8.1f, 8.1f, 0.000227f, 0, 0},
    {"Light rain in a car", 0, 89, 3.45f, 5.f, 8.1f, 8.1f, 0.000227f, 0, 0},
    {"Heavy rain", 0, 13, 5.f, 5.f, 7.8f, 8.f, 0.000283f, 0, 0},
    {"Light rain", 0, 13, 3.45f, 3.45f, 8.0f, 8.3f, 0.000227f, 0, 0},
    {"Bubbles", 0, 129, 2.45f, 3.25f, 4.8f, 8.3f, 0.0229f, 0, 0},
    {"Earth rumbling", 0, 30, 1.95f, 5.f, 2.5f, 3.2f, 0.1589f, 0, 0},
    {"Sine wind", 1, 1, 0.f, 0.f, 4.6f, 6.8f, 2.268f, 0.0005f, 0.2f},
    {"Kettle whistle pure", 1, 1, 0.f, 0.f, 7.5f, 7.7f, 0.0005f, 0.0005f, 0.f},
    {"Kettle whistle mixed", 2, 7, .9f, .9f, 7.5f, 7.7f, 0.00716f, 0, 0}};
// a parameter as the program array holds it and setupAudioElement reads it back: normalize<>() then denorm<>() (include/normalization.h:8-15)
float through_params(float v, float m, float M) { const float x = (v - m) / (M - m); return m + x * (M - m); }

void voice_common_defaults(ra_voice_config *c) { // make_common, gen.voice.h:474-531 + include/loudness_init_values.h
  std::memset(c, 0, sizeof(*c));
  c->sample_rate = 48000;
  c->n_channels = 16;
  c->sine_gain = 1.f;
  c->loudness_level = 30.f; c->loudness_compensation_amount = 1.f; c->loudness_ref_freq_index = 5;
  c->pan = 0.f;
  c->low_freq = 100.f; c->high_freq = 20000.f;
  c->freq_interpolation = RA_ITP_EASE_IN_EXPO;
  c->order_filters = 1; c->center_octave_min_long_term = 1.f; c->center_octave_max_long_term = 8.f; c->seconds_slow_iter_long_term = 1.f; // make_common :499-505
}
int pow2_ceil_log2(int v) { int k = 0; while ((1 << k) < v) ++k; return k; }
} // namespace

int ra_voice_default_config(ra_voice_config *c) {
  if (!c) return fail(RA_EINVAL, "null config");
  voice_common_defaults(c);
  c->mode = RA_VOICE_SWEEP;
  c->interpolation = RA_ITP_LINEAR;
  c->length_ms = 73.f; c->length_exponent = 5.f; c->xfade_length_seconds = 0.0109f; // make_sweep(LINEAR, 73, 5, 0.0109, ...), gen.voice.h:752-754
  c->random_pan = 0;
  return RA_OK;
}

int ra_voice_program(int32_t mode, int32_t index, ra_voice_config *c, const char **name) {
  if (!c) return fail(RA_EINVAL, "null config");
  const VoiceProgram *tab = nullptr;
  int count = 0;
  if (mode == RA_VOICE_BIRDS) { tab = kBirds; count = 9; }
  else if (mode == RA_VOICE_ROBOTS) { tab = kRobots; count = 2; }
  else if (mode == RA_VOICE_SWEEP) { tab = kSweeps; count = 2; }
  else if (mode == RA_VOICE_WIND) {
    const int n_wind = (int)(sizeof(kWinds) / sizeof(kWinds[0]));
    if (index < 0 || index >= n_wind) return fail(RA_EINVAL, "ra_voice_program: no such program");
    const WindProgram &g = kWinds[index];
    voice_common_defaults(c);
    c->mode = mode;
    c->random_pan = 1; c->seed_param = 0;
    // make_common(0, 0, 6, 0, 0, itp, 0.12f, 93.3f, 2.f, .5f, 0.0499f, ...), gen.voice.h:629,654,679
    c->markov_start_node = 0; c->markov_pre_tries = 0; c->markov_min_path_length = 6; c->markov_additional_tries = 0;
    c->interpolation = (g.kind == 1) ? RA_ITP_LINEAR : RA_ITP_PROPORTIONAL_VALUE_DERIVATIVE;
    c->freq_scatter = 0.12f; c->length_ms = 93.3f; c->length_exponent = 2.f; c->length_exponent_scatter = .5f; c->xfade_length_seconds = 0.0499f;
    c->order_filters = g.order;
    c->pink_noise_lp_gain = 0.f; c->pink_noise_br_gain = 0.f;
    c->pink_noise_bp_gain = (g.kind == 1) ? 0.f : 1.f;
    c->sine_gain = (g.kind == 0) ? 0.f : (g.kind == 1 ? 0.1f : 0.01f);
    c->loudness_compensation_amount = (g.kind == 2) ? 0.f : 1.f; // (:664, :690; make_noise_wind keeps make_common's 1)
    c->bp_octave_width_min = through_params(g.w0, 0.f, 5.f); c->bp_octave_width_max = through_params(g.w1, 0.f, 5.f);
    c->center_octave_min_long_term = through_params(g.c0, 0.f, 10.f); c->center_octave_max_long_term = through_params(g.c1, 0.f, 10.f);
    constexpr float max_seconds_slow_iter = 2.268f;
    c->seconds_slow_iter_long_term = std::log(g.sec_long) / std::log(max_seconds_slow_iter);
    c->seconds_slow_iter_short_term = (g.kind == 1) ? std::log(g.sec_short) / std::log(max_seconds_slow_iter) : 0.f; // (make_common's 0.f otherwise)
    c->center_short_term_ratio = g.short_ratio;
    if (name) *name = g.name;
    return RA_OK;
  }
  else return fail(RA_EINVAL, "ra_voice_program: mode must be a RA_VOICE_* value");
  if (index < 0 || index >= count) return fail(RA_EINVAL, "ra_voice_program: no such program");
  const VoiceProgram &g = tab[index];
  voice_common_defaults(c);
  c->mode = mode;
  c->random_pan = 1; // make_common stores RANDOM_PAN = !true = 0, which setupAudioElement reads as "random" (gen.voice.h:511,983)
  c->seed_param = 0; // SEED 0: another seed per note
  c->interpolation = g.itp;
  c->length_ms = g.length; c->length_exponent = g.ex; c->xfade_length_seconds = g.xfade;
  if (mode == RA_VOICE_SWEEP) { c->low_freq = g.low; c->high_freq = g.high; }
  else {
    c->markov_start_node = g.start; c->markov_pre_tries = g.pre; c->markov_min_path_length = g.path; c->markov_additional_tries = g.extra;
    c->articulative_pause_seconds = g.pause; c->freq_scatter = g.scat; c->length_exponent_scatter = g.ex_scat;
  }
  if (mode == RA_VOICE_BIRDS) { c->xfade_freq = g.xfade_freq; c->freq_transition_seconds = g.freq_xfade; c->freq_interpolation = g.itp; }
  if (mode == RA_VOICE_ROBOTS) { c->d1 = g.d1; c->d2 = g.d2; c->harmonic_attenuation = g.att; }
  if (name) *name = g.name;
  return RA_OK;
}

// host-only: the pink-noise sound buffer WIND reads and a ramp's declared start positions (ra_voice_host.hpp), for CPU tests
int64_t ra_voice_pink_noise(int32_t sample_rate, double *h_out, int64_t cap, float *abs_mean) {
  if (sample_rate < 1000) return fail(RA_EINVAL, "sample_rate is too small");
  rav::PinkNoise pn;
  rav::build_pink_noise(sample_rate, pn);
  for (int64_t i = 0; h_out && i < cap && i < (int64_t)pn.values.size(); ++i) h_out[i] = pn.values[(size_t)i];
  if (abs_mean) *abs_mean = pn.abs_mean;
  return (int64_t)pn.values.size();
}
int32_t ra_voice_wind_start_position(uint64_t stream_seed, int64_t noteid, int32_t play, int32_t iterator, int32_t which, int64_t buffer_length) {
  if (noteid < 0 || play < 0 || play >= 30 || iterator < 0 || iterator >= 6 || which < 0 || which > 1 || buffer_length < 2) return fail(RA_EINVAL, "bad argument");
  return rav::wind_position(stream_seed, noteid, play, iterator, which, (size_t)buffer_length);
}

int ra_voice_expand_note(const ra_voice_config *cfg, float frequency, uint64_t stream_seed, int32_t seed_value, int64_t noteid, double *spec_d, int32_t *spec_i, float *spec_vol,
                         int64_t spec_cap, int64_t *played, int64_t played_cap, int64_t *info) {
  if (!cfg) return fail(RA_EINVAL, "null config");
  const char *why = nullptr;
  if (rav::config_error(*cfg, why)) return fail(RA_EINVAL, why);
  if (!(frequency > 0.f)) return fail(RA_EINVAL, "a note-on needs a frequency > 0 (soundengine.h:972-975)");
  rav::Note note;
  rav::expand_note(*cfg, frequency, stream_seed, seed_value, noteid, note);
  const int64_t ns = (int64_t)note.specs.size(), np = (int64_t)note.plays.size();
  if (info) {
    double pan = std::min(1., std::max(-1., (double)note.pan));
    const float gl = (float)std::cos(0.25 * M_PI * (pan + 1.));
    int32_t bits;
    std::memcpy(&bits, &gl, 4);
    info[0] = ns; info[1] = np; info[2] = note.done_step; info[3] = bits;
  }
  for (int64_t i = 0; i < ns && i < spec_cap; ++i) {
    const rav::Spec &s = note.specs[i];
    if (spec_d) { spec_d[4 * i] = s.from; spec_d[4 * i + 1] = s.to; spec_d[4 * i + 2] = s.dur; spec_d[4 * i + 3] = s.start; }
    if (spec_i) { spec_i[3 * i] = s.interp; spec_i[3 * i + 1] = s.silence ? 1 : 0; spec_i[3 * i + 2] = 0; }
    if (spec_vol) spec_vol[i] = s.vol;
  }
  for (int64_t i = 0; i < np && i < played_cap; ++i)
    if (played) { played[4 * i] = note.plays[i].k; played[4 * i + 1] = note.plays[i].ramp; played[4 * i + 2] = note.plays[i].spec; played[4 * i + 3] = note.plays[i].T; }
  if ((spec_d || spec_i || spec_vol) && ns > spec_cap) return fail(RA_EOVERFLOW, "ra_voice_expand_note: spec capacity too small");
  if (played && np > played_cap) return fail(RA_EOVERFLOW, "ra_voice_expand_note: play capacity too small");
  return RA_OK;
}

namespace {
// setupAudioElement's WIND branches (gen.voice.h:936-978 over SetFilterWidths / ConfigureFilters / SetGains / SetSlowParams, :334-408): what is the same
// for every ramp of the object; the pink-noise buffer goes to the device, the ramps' state slots are allocated once
int voice_wind_setup(ra_voice *v) {
  const ra_voice_config &c = v->cfg;
  rav::PinkNoise pink;
  rav::build_pink_noise(c.sample_rate, pink);
  RaWindParams &w = v->wp;
  std::memset(&w, 0, sizeof(w));
  w.order = c.order_filters;
  w.pink_len = (int)pink.values.size();
  { // soundBufferWrapperAlgo::setStartAngle(0) (audioelement.h:1544-1557): the index before the first step
    int index = static_cast<int>(((0. + 1.) * (double)pink.values.size() * 0.5) + 0.5);
    w.pink_start = std::max(index, 0) - 1;
  }
  w.g_bp = c.pink_noise_bp_gain; w.g_br = c.pink_noise_br_gain; w.g_sine = c.sine_gain;
  w.width_min = std::min(c.bp_octave_width_min, c.bp_octave_width_max); w.width_max = std::max(c.bp_octave_width_min, c.bp_octave_width_max);
  { // octaveRangeToAngleIncrementsRange (:854-879) -> SoundEngineFreqCtrl::setAngleIncrementsRange (soundengine.h:56-63)
    constexpr float lowest_f = 10.f;
    float fm = lowest_f * std::pow(2.f, c.center_octave_min_long_term), fM = lowest_f * std::pow(2.f, c.center_octave_max_long_term);
    if (fm > fM) std::swap(fm, fM);
    const float ra_min = rav::inc_of(fm, c.sample_rate), ra_max = rav::inc_of(fM, c.sample_rate);
    w.log_inc_min = static_cast<float>(std::log(static_cast<double>(ra_min)));
    w.log_inc_max = static_cast<float>(std::log(static_cast<double>(ra_max)));
    w.inv_approx_range = 1.f / (2.f * pink.abs_mean);
  }
  const float n_slow_steps = rav::wind_slow_steps(c.sample_rate, c.seconds_slow_iter_long_term);
  const float n_slow_steps_short = rav::wind_slow_steps(c.sample_rate, c.seconds_slow_iter_short_term);
  w.n_long = static_cast<int>(n_slow_steps);                                   // SetSlowParams::set(ctrl, int, int n_slow_steps_long, float)
  w.n_band = static_cast<int32_t>(static_cast<unsigned int>(n_slow_steps));    // ConfigureFilters::configure(mix, unsigned int n_slow_steps, ra)
  w.short_rate = static_cast<float>(static_cast<int>(n_slow_steps_short));     // set_short_term_noise_rate(float) fed the int
  w.short_amplitude = c.center_short_term_ratio;
  w.interp_long = c.interpolation;
  HIP_TRY(hipStreamCreateWithFlags(&v->wind_stream, hipStreamNonBlocking));
  for (auto *e : {&v->ev_wctl[0], &v->ev_wctl[1], &v->ev_wchain[0], &v->ev_wchain[1], &v->ev_wblock}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  const size_t slots = (size_t)v->n_streams * (size_t)c.n_channels * 3;
  if (v->n_streams > 65535 || slots > (size_t)INT32_MAX) return fail(RA_EINVAL, "WIND: at most 65535 streams per object (the mix kernel's grid)");
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&v->d_pink), sizeof(double) * pink.values.size()));
  HIP_TRY(hipMemcpy(v->d_pink, pink.values.data(), sizeof(double) * pink.values.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&v->d_wstates), sizeof(RaWindState) * slots));
  HIP_TRY(hipMemset(v->d_wstates, 0, sizeof(RaWindState) * slots));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&v->d_wfstates), sizeof(double) * slots * 4 * (size_t)w.order));
  HIP_TRY(hipMemset(v->d_wfstates, 0, sizeof(double) * slots * 4 * (size_t)w.order));
  return RA_OK;
}
} // namespace

int ra_voice_create(const ra_voice_config *cfg, int32_t n_streams, int32_t device, ra_voice **out) {
  if (!cfg || !out) return fail(RA_EINVAL, "null argument");
  *out = nullptr;
  const ra_voice_config &c = *cfg;
  if (n_streams < 1) return fail(RA_EINVAL, "n_streams must be >= 1");
  const char *why = nullptr;
  if (rav::config_error(c, why)) return fail(RA_EINVAL, why);
  if (c.mode == RA_VOICE_SWEEP) { // the one spec of a sweep is known now: refuse here what playNextSpec asserts (soundengine.h:797-799)
    float len = c.length_ms;
    len *= std::pow(2.f, c.length_exponent);
    const float n_frames = static_cast<float>(rav::ms_to_frames(len, c.sample_rate));
    if (!(n_frames > 0.f)) return fail(RA_EINVAL, "length is too small");
    if (static_cast<int>(.5f + n_frames) - rav::xfade_len_of(c) < 0) return fail(RA_EINVAL, "the sweep is shorter than its cross-fade (soundengine.h:797-799)");
  }
  RaVoiceParams p;
  std::memset(&p, 0, sizeof(p));
  p.sample_rate = c.sample_rate;
  p.sine_gain = c.sine_gain;
  p.log_ratio = c.loudness_compensation_amount;
  p.db_to_log2 = c.loudness_compensation_amount / 20.f * 3.32192809488736f;
  { // include/loudness.h:199-229
    int li = static_cast<int>(c.loudness_level * .1f) - 2;
    li = std::max(0, li); li = std::min(8, li);
    for (int i = 0; i < 29; ++i) { p.loud_freqs[i] = kLoudFreqs[i]; p.loud_elv[i] = equal_loudness_volume(i, (li + 2) * 10.f); }
    p.max_db = p.loud_elv[c.loudness_ref_freq_index];
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RA_ENODEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RA_EINVAL, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  ra_voice *v = new ra_voice();
  v->cfg = c; v->p = p; v->device = device; v->n_streams = n_streams;
  v->synths.resize(n_streams);
  for (auto &sy : v->synths) sy.chans.assign(c.n_channels, {});
  {
    int threads = std::min(8, std::max(1, (int)std::thread::hardware_concurrency() / 2));
    if (const char *e = std::getenv("RA_VOICE_THREADS")) threads = std::max(1, std::min(64, std::atoi(e)));
    v->pool.reset(new VoicePool(threads));
  }
  v->profiling = std::getenv("RA_VOICE_PROFILE") != nullptr;
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&v->staging[i].done, hipEventDisableTiming);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&v->ev_pre_done[i], hipEventDisableTiming);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&v->ev_render_done[i], hipEventDisableTiming);
  if (e == hipSuccess) { // above the caller's stream in dispatch priority: beside a render whose grid fills the chip an equal-priority queue waits for the grid's tail
    int prio_least = 0, prio_greatest = 0;
    e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&v->pre_stream, hipStreamNonBlocking, prio_greatest);
  }
  if (e != hipSuccess) { ra_voice_destroy(v); return fail(RA_EHIP, std::string("ra_voice_create: ") + hipGetErrorString(e)); }
  if (c.mode == RA_VOICE_WIND) {
    if (int r = voice_wind_setup(v)) { ra_voice_destroy(v); return r; }
  }
  *out = v;
  return RA_OK;
}

void ra_voice_destroy(ra_voice *v) {
  if (!v) return;
  if (std::getenv("RA_VOICE_PROFILE") && v->host_blocks)
    std::fprintf(stderr, "ra_voice host ms per block (%lld blocks): jobs %.3f prefix %.3f staging-wait %.3f emit %.3f grow+copies %.3f launches %.3f\n", (long long)v->host_blocks,
                 v->host_ms[0] / v->host_blocks, v->host_ms[1] / v->host_blocks, v->host_ms[2] / v->host_blocks, v->host_ms[3] / v->host_blocks, v->host_ms[4] / v->host_blocks,
                 v->host_ms[5] / v->host_blocks);
  (void)hipSetDevice(v->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(v->out);
  for (int i = 0; i < 2; ++i) {
    (void)hipFree(v->d_jobs[i]); (void)hipFree(v->d_order[i]); (void)hipFree(v->d_slot_begin[i]);
    if (v->ev_pre_done[i]) (void)hipEventDestroy(v->ev_pre_done[i]);
    if (v->ev_render_done[i]) (void)hipEventDestroy(v->ev_render_done[i]);
  }
  for (int i = 0; i < 3; ++i) (void)hipFree(v->d_states[i]);
  (void)hipFree(v->d_seg);
  if (v->wind_stream) (void)hipStreamDestroy(v->wind_stream);
  for (auto e : {v->ev_wctl[0], v->ev_wctl[1], v->ev_wchain[0], v->ev_wchain[1], v->ev_wblock}) if (e) (void)hipEventDestroy(e);
  (void)hipFree(v->d_pink); (void)hipFree(v->d_wfstates); (void)hipFree(v->d_wscratch); (void)hipFree(v->d_wstates); (void)hipFree(v->d_wjobs); (void)hipFree(v->d_wstream_begin);
  if (v->pre_stream) (void)hipStreamDestroy(v->pre_stream);
  for (auto &s : v->staging) { if (s.h) (void)hipHostFree(s.h); if (s.done) (void)hipEventDestroy(s.done); }
  for (auto &t : v->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); (void)hipEventDestroy(t.b2); (void)hipEventDestroy(t.c); }
  for (auto &e : v->ev_pool) (void)hipEventDestroy(e);
  delete v;
}

int ra_voice_set_hip_stream(ra_voice *v, void *hip_stream) {
  if (!v) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(v->device));
  HIP_TRY(hipStreamSynchronize(v->pre_stream));
  HIP_TRY(hipStreamSynchronize(v->stream)); // what was enqueued on the old stream is finished before anything follows on the new one
  v->stream = static_cast<hipStream_t>(hip_stream);
  return RA_OK;
}

int ra_voice_synchronize(ra_voice *v) {
  if (!v) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(v->device));
  HIP_TRY(hipStreamSynchronize(v->pre_stream));
  if (v->wind_stream) HIP_TRY(hipStreamSynchronize(v->wind_stream));
  HIP_TRY(hipStreamSynchronize(v->stream));
  return RA_OK;
}

int ra_voice_events(ra_voice *v, const ra_voice_event *ev, int64_t n) {
  if (!v) return fail(RA_EINVAL, "null context");
  if (n < 0 || (n > 0 && !ev)) return fail(RA_EINVAL, "bad events");
  for (int64_t i = 0; i < n; ++i) {
    const ra_voice_event &e = ev[i];
    const std::string at = "voice event " + std::to_string(i) + ": ";
    if (e.stream < 0 || e.stream >= v->n_streams) return fail(RA_EINVAL, at + "stream out of range");
    if (e.type != 0 && e.type != 1) return fail(RA_EINVAL, at + "type must be 0 (NoteOn) or 1 (NoteOff)");
    if (e.sample < v->t) return fail(RA_EINVAL, at + "its frame has already been processed");
    if (e.sample < v->synths[e.stream].last_sample) return fail(RA_EINVAL, at + "samples of a stream must not decrease");
    if (e.type == 0 && !(e.velocity > 0.f)) return fail(RA_EINVAL, at + "a note-on needs a velocity > 0 (gen.crtp.h:391)");
    if (e.type == 0 && !(e.frequency > 0.f)) return fail(RA_EINVAL, at + "a note-on needs a frequency > 0 (soundengine.h:972-975)");
    ra_voice::Synth &sy = v->synths[e.stream];
    if (sy.head == sy.pending.size()) { sy.pending.clear(); sy.head = 0; }
    else if (sy.head > 4096 && sy.head * 2 > sy.pending.size()) { sy.pending.erase(sy.pending.begin(), sy.pending.begin() + (ptrdiff_t)sy.head); sy.head = 0; }
    sy.pending.push_back(e);
    sy.last_sample = e.sample;
  }
  return RA_OK;
}

namespace {
int voice_grow_bytes(void **ptr, int64_t &cap, int64_t need, size_t elem, hipStream_t st, hipStream_t st2) { // device buffers only grow; both queues are drained before one is replaced
  if (need <= cap) return RA_OK;
  HIP_TRY(hipStreamSynchronize(st2));
  HIP_TRY(hipStreamSynchronize(st));
  (void)hipFree(*ptr);
  *ptr = nullptr;
  cap = std::max<int64_t>(1024, need + need / 2);
  HIP_TRY(hipMalloc(ptr, elem * (size_t)cap));
  return RA_OK;
}
#define voice_grow(ptr, cap, need, v) voice_grow_bytes(reinterpret_cast<void **>(ptr), cap, need, sizeof(**(ptr)), (v)->stream, (v)->pre_stream)

// the block's events of one stream: ImplCRTP::onEvent (gen.crtp.h:389-449, 541-588) with the engine's acquisition rule
void voice_stream_events(ra_voice *v, int s, int64_t t0, int64_t t1) {
  ra_voice::Synth &sy = v->synths[s];
  const int P = v->cfg.n_channels;
  ra_voice::Chan *ch = sy.chans.data();
  while (sy.head < sy.pending.size() && sy.pending[sy.head].sample < t1) {
    const ra_voice_event e = sy.pending[sy.head++];
    const int64_t te = std::max(e.sample, t0); // (events are not accepted before the frames already processed)
    if (e.type == 0) { // NoteOn: the first channel whose engine is finished
      int sel = -1;
      for (int c = 0; c < P; ++c) if (!ch[c].odd && !ch[c].lost) { sel = c; break; }
      if (sel < 0) { ++sy.dropped; continue; }
      // setupAudioElement: the reseed (gen.voice.h:910-925), the chain, the schedule
      int32_t seed_value = 0;
      if (v->cfg.mode != RA_VOICE_SWEEP) {
        int32_t &sc = sy.seed_counter;
        if (v->cfg.seed_param) sc = v->cfg.seed_param;
        else if (sc == 257) sc = 1;
        else ++sc;
        seed_value = sc;
      }
      rav::Note &note = sy.scratch;
      rav::expand_note(v->cfg, e.frequency, v->cfg.seed + (uint64_t)s, seed_value, e.noteid, note);
      ra_voice::Chan &c = ch[sel];
      c.odd = true; c.noteid = e.noteid; c.t_on = te;
      c.busy.clear();
      ra_voice::Live l;
      l.chan = sel; l.done = false; l.t_on = te; l.noteid = e.noteid; l.limit = INT64_MAX; l.A = note.A; l.R = note.R;
      {
        const double o_vol = ((static_cast<double>(0.1f) * 1.) / static_cast<double>(0.1f)) * (double)e.velocity; // Element::baseVolume * velocity (gen.crtp.h:425; audioelement.h:1279,2370)
        double pan = (double)note.pan;
        pan = std::min(pan, 1.); pan = std::max(pan, -1.);
        const double ang = (0.25 * M_PI) * (pan + 1.); // stereo(), include/stereo.h:18-27
        l.fL = o_vol * std::cos(ang); l.fR = o_vol * std::sin(ang);
      }
      for (const rav::Play &pl : note.plays) {
        c.busy.emplace_back(pl.k, pl.T > 0 ? pl.k + pl.T + note.R + 16 : pl.k + 16);
        if (pl.T > 0 && pl.T + note.R - 1 > 0) l.ramps.push_back({note.specs[pl.spec], pl.k, pl.T, -1, pl.ramp, pl.spec});
      }
      if (!l.ramps.empty()) { c.live = (int)sy.live.size(); sy.live.push_back(std::move(l)); }
      else c.live = -1;
    } else { // NoteOff: the first channel that last played this note id; it reacts only while its note is on
      int sel = -1;
      for (int c = 0; c < P; ++c) if (ch[c].noteid == e.noteid) { sel = c; break; }
      if (sel < 0 || !ch[sel].odd) continue;
      const int64_t k = te - ch[sel].t_on; // steps the element took
      ch[sel].odd = false;
      for (const auto &b : ch[sel].busy) if (k >= b.first && k < b.second) { ch[sel].lost = true; break; }
      if (ch[sel].live >= 0) { ra_voice::Live &l = sy.live[ch[sel].live]; l.limit = std::min(l.limit, k); }
      ch[sel].live = -1;
    }
  }
}

// one stream's share of a block, first half: events, the live notes' ramps cut into this block's jobs (state slots numbered within the
// stream), the jobs packed into slots — a stream's jobs in begin order, each into the lowest slot that is free again (greedy interval colouring)
void voice_stream_jobs(ra_voice *v, int s, int64_t t0, int64_t t1) {
  ra_voice::Synth &sy = v->synths[s];
  const int L = ra_voice::kTile;
  voice_stream_events(v, s, t0, t1);
  sy.jobs.clear();
  sy.n_states = 0; sy.n_work = 0;
  for (size_t li = 0; li < sy.live.size(); ++li) {
    ra_voice::Live &l = sy.live[li];
    bool more = false;
    for (size_t ri = 0; ri < l.ramps.size(); ++ri) {
      ra_voice::LiveRamp &r = l.ramps[ri];
      const int64_t k_last = std::min(r.k + r.T + l.R - 2, l.limit); // the ramp's last sounding element step (1-based), cut by the note-off
      const int64_t fa = std::max(l.t_on + r.k - 1, t0), fz = std::min(l.t_on + k_last - 1, t1 - 1);
      const bool goes_on = l.t_on + k_last - 1 >= t1;
      more |= goes_on;
      if (fa > fz) continue;
      ra_voice::JobRec jr;
      RaVoiceJob &j = jr.j;
      j.from = r.spec.from; j.to = r.spec.to; j.dur = r.spec.dur; j.cur0 = r.spec.start;
      j.vol = (double)r.spec.vol; j.fL = l.fL; j.fR = l.fR;
      j.C = (j.to == j.from) ? (double)1.f : -std::log(j.from / j.to) / (j.to - j.from);
      j.l2r = (r.spec.interp == RA_ITP_PROPORTIONAL_VALUE_DERIVATIVE) ? std::log2(j.to / j.from) : 0.0;
      j.begin = (int32_t)(fa - t0); j.n = (int32_t)(fz - fa + 1);
      j.j0 = (int32_t)(fa - l.t_on - r.k + 2);
      j.T = r.T; j.A = l.A; j.R = l.R; j.interp = r.spec.interp;
      j.state_in = (j.j0 == 1) ? -1 : r.carry;
      j.stream = s; j.slot = 0;
      const int32_t tiles = (j.begin + j.n - 1) / L - j.begin / L;
      j.state_tiles = sy.n_states;
      sy.n_states += tiles;
      j.state_out = goes_on ? sy.n_states++ : -1;
      if (tiles > 0 || goes_on) ++sy.n_work;
      jr.end = j.begin + j.n; jr.live = (int32_t)li; jr.ramp = (int32_t)ri;
      sy.jobs.push_back(jr);
    }
    l.done = !more;
  }
  std::sort(sy.jobs.begin(), sy.jobs.end(), [](const ra_voice::JobRec &a, const ra_voice::JobRec &b) { return a.j.begin < b.j.begin; });
  sy.slot_end.clear();
  for (auto &jr : sy.jobs) {
    int sl = -1;
    for (size_t k = 0; k < sy.slot_end.size(); ++k) if (sy.slot_end[k] <= jr.j.begin) { sl = (int)k; break; }
    if (sl < 0) { sl = (int)sy.slot_end.size(); sy.slot_end.push_back(0); }
    sy.slot_end[sl] = jr.end;
    jr.j.slot = sl;
  }
  sy.slots_used = (int32_t)sy.slot_end.size();
}

// retire the notes that have been rendered to their end (a channel stays taken until its note-off)
void voice_retire(ra_voice::Synth &sy) {
  size_t keep = 0;
  for (size_t li = 0; li < sy.live.size(); ++li) {
    ra_voice::Live &l = sy.live[li];
    ra_voice::Chan &c = sy.chans[l.chan];
    const bool mine = c.live == (int)li;
    if (!l.done) { if (mine) c.live = (int)keep; if (keep != li) sy.live[keep] = std::move(l); ++keep; }
    else if (mine) c.live = -1;
  }
  sy.live.resize(keep);
}

// second half, once the streams' offsets are known: the state slots become block-wide indices (and are remembered by the ramps that go
// on), the jobs go to the staging image sorted by (slot, begin), finished notes are retired
void voice_stream_emit(ra_voice *v, int s, int g, RaVoiceJob *hj, int32_t *hs, int32_t *ho) {
  ra_voice::Synth &sy = v->synths[s];
  const int L = ra_voice::kTile;
  int32_t fill[1024];
  for (int k = 0; k < g; ++k) fill[k] = 0;
  for (auto &jr : sy.jobs) ++fill[jr.j.slot];
  int64_t o = sy.job_base;
  for (int k = 0; k < g; ++k) { hs[(size_t)s * g + k] = (int32_t)o; const int c = fill[k]; fill[k] = (int32_t)o; o += c; }
  int64_t w = sy.work_base;
  for (auto &jr : sy.jobs) { // (begin order within a slot is the order of `jobs`)
    RaVoiceJob &j = jr.j;
    j.state_tiles += (int32_t)sy.state_base;
    if (j.state_out >= 0) j.state_out += (int32_t)sy.state_base;
    sy.live[jr.live].ramps[jr.ramp].carry = j.state_out;
    const int at = fill[j.slot]++;
    hj[at] = j;
    if (j.state_out >= 0 || (j.begin + j.n - 1) / L > j.begin / L) ho[w++] = at;
  }
  voice_retire(sy);
}

// WIND: the stream's jobs as the wind kernels want them (ra_wind.cuh) — the ramp's state slot and the declared start positions of its six iterators
void voice_wind_emit(ra_voice *v, int s, RaWindJob *hj, int32_t *hs) {
  ra_voice::Synth &sy = v->synths[s];
  const int P = v->cfg.n_channels;
  const uint64_t key = v->cfg.seed + (uint64_t)s;
  hs[s] = (int32_t)sy.job_base;
  int64_t at = sy.job_base;
  for (auto &jr : sy.jobs) {
    const ra_voice::Live &l = sy.live[jr.live];
    const ra_voice::LiveRamp &r = l.ramps[jr.ramp];
    RaWindJob w;
    std::memset(&w, 0, sizeof(w));
    w.vol = jr.j.vol; w.fL = jr.j.fL; w.fR = jr.j.fR;
    w.begin = jr.j.begin; w.n = jr.j.n; w.j0 = jr.j.j0; w.T = jr.j.T; w.A = jr.j.A; w.R = jr.j.R;
    w.stream = s;
    w.slot = (int32_t)((((int64_t)s * P + l.chan) * 3) + r.ramp);
    if (w.j0 == 1)
      for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 2; ++k) w.pos[i][k] = rav::wind_position(key, l.noteid, r.play, i, k, (size_t)v->wp.pink_len);
    hj[at++] = w;
  }
  voice_retire(sy);
}

void *voice_stage(ra_voice *v, size_t bytes, int &err) { // the pinned staging buffer of this block (waits for the block before last to have drained it)
  ra_voice::Staging &sg = v->staging[v->stage_parity];
  err = RA_OK;
  if (sg.used) { if (hipEventSynchronize(sg.done) != hipSuccess) { err = fail(RA_EHIP, "hipEventSynchronize(staging)"); return nullptr; } }
  if (bytes > sg.cap) {
    if (sg.h) (void)hipHostFree(sg.h);
    sg.h = nullptr;
    sg.cap = bytes + bytes / 2 + 4096;
    if (hipHostMalloc(&sg.h, sg.cap, hipHostMallocDefault) != hipSuccess) { sg.cap = 0; err = fail(RA_EHIP, "hipHostMalloc(staging)"); return nullptr; }
  }
  return sg.h;
}

// A WIND block after the streams' events and jobs: the jobs go up through the staging buffer, then controls -> filter chains -> mix per chunk of
// the block; the controls on a queue of the object's own, one chunk ahead of the chains and the mix on the caller's stream.  The chunk length keeps the scratch
// (up to 13 doubles per job and frame) near 1.5 GiB.
int voice_wind_block(ra_voice *v, int64_t n, float *d_out) {
  const int S = v->n_streams;
  int64_t n_jobs = 0;
  for (auto &sy : v->synths) { sy.job_base = n_jobs; n_jobs += (int64_t)sy.jobs.size(); }
  if (n_jobs > INT32_MAX - 2) return fail(RA_EINVAL, "the block has too many ramps: process it in shorter pieces");
  const size_t bytes_jobs = sizeof(RaWindJob) * (size_t)n_jobs, bytes_sb = sizeof(int32_t) * (size_t)(S + 1);
  static_assert(sizeof(RaWindJob) % 8 == 0, "the stream offsets follow the jobs in the staging image");
  int err = RA_OK;
  char *h = static_cast<char *>(voice_stage(v, bytes_jobs + bytes_sb + 64, err));
  if (err) return err;
  RaWindJob *hj = reinterpret_cast<RaWindJob *>(h);
  int32_t *hs = reinterpret_cast<int32_t *>(h + bytes_jobs);
  v->pool->run(S, [v, hj, hs](int s) { voice_wind_emit(v, s, hj, hs); });
  hs[S] = (int32_t)n_jobs;
  hipStream_t st = v->stream;
  RA_TRY(voice_grow(&v->d_wjobs, v->wjobs_cap, n_jobs + 1, v));
  RA_TRY(voice_grow(&v->d_wstream_begin, v->wsb_cap, (int64_t)S + 1, v));
  if (n_jobs) HIP_TRY(hipMemcpyAsync(v->d_wjobs, hj, bytes_jobs, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(v->d_wstream_begin, hs, bytes_sb, hipMemcpyHostToDevice, st));
  {
    ra_voice::Staging &sg = v->staging[v->stage_parity];
    HIP_TRY(hipEventRecord(sg.done, st));
    sg.used = true;
    v->stage_parity ^= 1;
  }
  const RaWindParams &w = v->wp;
  // scratch per job and frame: what the controls write (two coefficients per band, the sine's output) exists twice — the controls of chunk k + 1 run on their
  // own queue beside the chains of chunk k, a latency-bound kernel of one wave per job beside an issue-bound one —, the chains' outputs once
  const int n_ctl = (w.g_bp != 0.f ? 2 : 0) + (w.g_br != 0.f ? 2 : 0) + (w.g_sine != 0.f ? 1 : 0), n_out = (w.g_bp != 0.f ? 1 : 0) + (w.g_br != 0.f ? 2 : 0);
  const int arrays = std::max(1, 2 * n_ctl + n_out);
  const int64_t jobs1 = std::max<int64_t>(1, n_jobs);
  int64_t ch = (int64_t(3) << 29) / (jobs1 * arrays * 8); // ~1.5 GiB
  ch = std::max<int64_t>(4096, ch);
  if (const char *e = std::getenv("RA_WIND_CHUNK")) ch = std::max(64, std::atoi(e)); // (tests: a ramp's state across chunk boundaries)
  ch = std::min<int64_t>(ch, n);
  RA_TRY(voice_grow(&v->d_wscratch, v->wscratch_cap, jobs1 * arrays * ch, v));
  RaWindScratch sc2[2];
  {
    double *q = v->d_wscratch;
    const size_t each = (size_t)jobs1 * (size_t)ch;
    auto take = [&] { double *r = q; q += each; return r; };
    double *p_bp = nullptr, *p_lp = nullptr, *p_hp = nullptr;
    if (w.g_bp != 0.f) p_bp = take();
    if (w.g_br != 0.f) { p_lp = take(); p_hp = take(); }
    for (auto &sc : sc2) {
      std::memset(&sc, 0, sizeof(sc));
      sc.stride = ch;
      sc.p_bp = p_bp; sc.p_br_lp = p_lp; sc.p_br_hp = p_hp;
      if (w.g_bp != 0.f) { sc.k_bp_lp = take(); sc.k_bp_hp = take(); }
      if (w.g_br != 0.f) { sc.k_br_lp = take(); sc.k_br_hp = take(); }
      if (w.g_sine != 0.f) sc.p_sine = take();
    }
  }
  ra_voice::Timed tm{nullptr, nullptr, nullptr, nullptr};
  if (v->profiling) {
    hipEvent_t *evs[4] = {&tm.a, &tm.b, &tm.b2, &tm.c};
    for (auto *pe : evs) {
      if (!v->ev_pool.empty()) { *pe = v->ev_pool.back(); v->ev_pool.pop_back(); }
      else HIP_TRY(hipEventCreate(pe));
    }
    HIP_TRY(hipEventRecord(tm.a, st)); HIP_TRY(hipEventRecord(tm.b, st)); HIP_TRY(hipEventRecord(tm.b2, st)); // (reported as the render: the controls' queue runs inside this interval)
  }
  // the controls' queue starts behind the job table's upload and behind whatever the caller's stream still does with the scratch (the last block's chains)
  // ... when the chains run as scans (every chain within one wave): beside the wave pipelines of longer chains — issue-bound at 0.69 of the nominal rate —
  // the controls only compete for the same issue slots (measured, order 129: 71 -> 80 ms per step with the second queue): one queue then
  const bool scans = (w.g_bp == 0.f || 2 * w.order <= 64) && (w.g_br == 0.f || w.order <= 64);
  static const bool one_queue = std::getenv("RA_WIND_ONE_QUEUE") != nullptr; // tuning aid
  hipStream_t sc_q = (scans && !one_queue) ? v->wind_stream : st;
  HIP_TRY(hipEventRecord(v->ev_wblock, st));
  HIP_TRY(hipStreamWaitEvent(sc_q, v->ev_wblock, 0));
  int k = 0;
  for (int64_t c0 = 0; c0 < n; c0 += ch, ++k) {
    const int clen = (int)std::min<int64_t>(ch, n - c0), q = k & 1;
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(sc_q, v->ev_wchain[q], 0)); // chunk k - 2's chains have read this half
    ra_wind_launch_controls(v->p, w, v->d_wjobs, (int)n_jobs, v->d_pink, v->d_wstates, sc2[q], (int)c0, clen, sc_q);
    HIP_TRY(hipEventRecord(v->ev_wctl[q], sc_q));
    HIP_TRY(hipStreamWaitEvent(st, v->ev_wctl[q], 0));
    ra_wind_launch_chains(w, v->d_wjobs, (int)n_jobs, v->d_wstream_begin, S, v->d_pink, v->d_wfstates, sc2[q], (int)c0, clen, n, d_out, st);
    HIP_TRY(hipEventRecord(v->ev_wchain[q], st));
  }
  HIP_TRY(hipGetLastError());
  if (v->profiling) { HIP_TRY(hipEventRecord(tm.c, st)); v->timed.push_back(tm); ++v->launches[0]; ++v->launches[1]; }
  ++v->host_blocks;
  v->t += n;
  return RA_OK;
}

int voice_block_impl(ra_voice *v, int64_t n, float *d_out, bool &started) {
  HIP_TRY(hipSetDevice(v->device));
  if (n > INT32_MAX / 4) return fail(RA_EINVAL, "a block must have fewer than 2^29 frames");
  started = true; // from here on the streams' queues, channels and live notes are being advanced
  const int S = v->n_streams, L = ra_voice::kTile;
  const int64_t t0 = v->t, t1 = v->t + n;
  auto clk = [] { return std::chrono::steady_clock::now(); };
  auto lap = [v](std::chrono::steady_clock::time_point &a, int i) { const auto b = std::chrono::steady_clock::now(); v->host_ms[i] += std::chrono::duration<double, std::milli>(b - a).count(); a = b; };
  auto tp = clk();
  v->pool->run(S, [v, t0, t1](int s) { voice_stream_jobs(v, s, t0, t1); });
  lap(tp, 0);
  if (v->cfg.mode == RA_VOICE_WIND) return voice_wind_block(v, n, d_out);
  int64_t n_jobs = 0, n_states = 0, n_work = 0;
  int g_needed = 1;
  for (auto &sy : v->synths) {
    sy.job_base = n_jobs; sy.state_base = n_states; sy.work_base = n_work;
    n_jobs += (int64_t)sy.jobs.size(); n_states += sy.n_states; n_work += sy.n_work;
    g_needed = std::max(g_needed, (int)sy.slots_used);
  }
  if (n_states > INT32_MAX - 2 || n_jobs > INT32_MAX - 2) return fail(RA_EINVAL, "the block has too many ramps: process it in shorter pieces");
  const int g_log2 = pow2_ceil_log2(g_needed), g = 1 << g_log2;
  if (g > 1024) return fail(RA_EINTERNAL, "more than 1024 overlapping ramps in a stream"); // (not reachable: <= 256 channels of three ramps each)
  // ---- staging image: jobs sorted by (stream, slot, begin), the slots' offsets, the jobs the pre-pass works on ----
  const int64_t n_slots = (int64_t)S * g + 1;
  const size_t bytes_jobs = sizeof(RaVoiceJob) * (size_t)n_jobs, bytes_slots = sizeof(int32_t) * (size_t)n_slots, bytes_order = sizeof(int32_t) * (size_t)n_work;
  int err = RA_OK;
  lap(tp, 1);
  char *h = static_cast<char *>(voice_stage(v, bytes_jobs + bytes_slots + bytes_order + 64, err));
  if (err) return err;
  lap(tp, 2);
  RaVoiceJob *hj = reinterpret_cast<RaVoiceJob *>(h);
  int32_t *hs = reinterpret_cast<int32_t *>(h + bytes_jobs), *ho = hs + n_slots;
  v->pool->run(S, [v, g, hj, hs, ho](int s) { voice_stream_emit(v, s, g, hj, hs, ho); });
  hs[(size_t)S * g] = (int32_t)n_jobs;
  lap(tp, 3);
  // ---- device buffers, copies, launches ----
  static_assert(sizeof(RaVoiceJob) == 120, "layout of the job record");
  const int bs = v->parity, cur = v->state_idx, prev = (cur + 2) % 3;
  hipStream_t sP = v->pre_stream, sR = v->stream;
  // (a buffer that must grow drains both queues first: its twin and the third state array grow with it, so that blocks of a steady size stop
  // growing anything after the first one; the array holding the carry-in keeps its contents and its size)
  for (int q : {bs, bs ^ 1}) {
    RA_TRY(voice_grow(&v->d_jobs[q], v->jobs_cap[q], n_jobs + 1, v));
    RA_TRY(voice_grow(&v->d_order[q], v->order_cap[q], n_work + 1, v));
    RA_TRY(voice_grow(&v->d_slot_begin[q], v->slots_cap[q], n_slots, v));
  }
  for (int q : {cur, (cur + 1) % 3}) RA_TRY(voice_grow(&v->d_states[q], v->states_cap[q], n_states + 1, v));
  RA_TRY(voice_grow(&v->d_seg, v->seg_cap, n_states + 1, v));
  if (!v->d_states[prev]) RA_TRY(voice_grow(&v->d_states[prev], v->states_cap[prev], 1, v));
  // this set and this state array were last read by the render of the block before last
  if (v->set_used[bs]) HIP_TRY(hipStreamWaitEvent(sP, v->ev_render_done[bs], 0));
  if (n_jobs) HIP_TRY(hipMemcpyAsync(v->d_jobs[bs], hj, bytes_jobs, hipMemcpyHostToDevice, sP));
  HIP_TRY(hipMemcpyAsync(v->d_slot_begin[bs], hs, bytes_slots, hipMemcpyHostToDevice, sP));
  if (n_work) HIP_TRY(hipMemcpyAsync(v->d_order[bs], ho, bytes_order, hipMemcpyHostToDevice, sP));
  {
    ra_voice::Staging &sg = v->staging[v->stage_parity];
    HIP_TRY(hipEventRecord(sg.done, sP));
    sg.used = true;
    v->stage_parity ^= 1;
  }
  lap(tp, 4);
  ra_voice::Timed tm{nullptr, nullptr, nullptr, nullptr};
  if (v->profiling) {
    hipEvent_t *evs[4] = {&tm.a, &tm.b, &tm.b2, &tm.c};
    for (auto *pe : evs) {
      if (!v->ev_pool.empty()) { *pe = v->ev_pool.back(); v->ev_pool.pop_back(); }
      else HIP_TRY(hipEventCreate(pe));
    }
    HIP_TRY(hipEventRecord(tm.a, sP));
  }
  ra_voice_launch_pre(v->p, v->d_jobs[bs], v->d_order[bs], (int)n_work, (int)n_states, v->d_states[prev], v->d_states[cur], v->d_seg, L, sP);
  HIP_TRY(hipGetLastError());
  if (v->profiling) HIP_TRY(hipEventRecord(tm.b, sP));
  HIP_TRY(hipEventRecord(v->ev_pre_done[bs], sP));
  HIP_TRY(hipStreamWaitEvent(sR, v->ev_pre_done[bs], 0));
  if (v->profiling) HIP_TRY(hipEventRecord(tm.b2, sR));
  if (ra_voice_launch_render(v->p, v->d_jobs[bs], v->d_slot_begin[bs], v->d_states[prev], v->d_states[cur], S, n, L, g_log2, d_out, sR) != 0)
    return fail(RA_EHIP, "ra_voice: the render launch was refused (too many (stream, tile) units, or hipFuncSetAttribute failed)");
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(v->ev_render_done[bs], sR));
  v->set_used[bs] = true;
  if (v->profiling) { HIP_TRY(hipEventRecord(tm.c, sR)); v->timed.push_back(tm); ++v->launches[0]; ++v->launches[1]; }
  lap(tp, 5);
  ++v->host_blocks;
  v->parity ^= 1;
  v->state_idx = (v->state_idx + 1) % 3;
  v->t += n;
  return RA_OK;
}
// A block either completes, or fails before it touched anything (a bad size: the call can be repeated), or fails midway: then the object is
// poisoned (ra_context's rule, include/resynth_amd.h) — its events are consumed, t / parity / state slots are not advanced — and refuses
// every later block; the caller recreates it.
int voice_block(ra_voice *v, int64_t n, float *d_out) {
  if (v->poisoned) return fail(RA_EINTERNAL, "an earlier block failed midway: the voice object is in an undefined state; destroy and recreate it");
  bool started = false;
  const int r = voice_block_impl(v, n, d_out, started);
  if (r && started) v->poisoned = true;
  return r;
}
} // namespace

int ra_voice_process_device_async(ra_voice *v, int64_t n, float *d_out) {
  if (!v) return fail(RA_EINVAL, "null context");
  if (n < 0 || (n > 0 && !d_out)) return fail(RA_EINVAL, "bad output");
  if (n == 0) return RA_OK;
  return voice_block(v, n, d_out);
}

int ra_voice_process_device(ra_voice *v, int64_t n, float *d_out) {
  RA_TRY(ra_voice_process_device_async(v, n, d_out));
  if (n > 0) HIP_TRY(hipStreamSynchronize(v->stream));
  return RA_OK;
}

int ra_voice_process_host(ra_voice *v, int64_t n, float *h_out) {
  if (!v) return fail(RA_EINVAL, "null context");
  if (n < 0 || (n > 0 && !h_out)) return fail(RA_EINVAL, "bad output");
  if (n == 0) return RA_OK;
  HIP_TRY(hipSetDevice(v->device));
  const int S = v->n_streams;
  if (n > v->out_cap) {
    HIP_TRY(hipStreamSynchronize(v->stream));
    (void)hipFree(v->out);
    v->out = nullptr;
    v->out_cap = n;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&v->out), sizeof(float) * 2 * static_cast<size_t>(S) * n));
  }
  if (int r = voice_block(v, n, v->out)) return r;
  HIP_TRY(hipMemcpyAsync(h_out, v->out, sizeof(float) * 2 * static_cast<size_t>(S) * n, hipMemcpyDeviceToHost, v->stream));
  HIP_TRY(hipStreamSynchronize(v->stream));
  return RA_OK;
}

int64_t ra_voice_dropped_note_ons(ra_voice *v) {
  if (!v) return fail(RA_EINVAL, "null context");
  int64_t d = 0;
  for (const auto &sy : v->synths) d += sy.dropped;
  return d;
}

int ra_voice_get_kernel_ms(ra_voice *v, double *ms2, int64_t *launches2, int32_t reset) {
  if (!v) return fail(RA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(v->device));
  HIP_TRY(hipStreamSynchronize(v->pre_stream));
  HIP_TRY(hipStreamSynchronize(v->stream));
  for (auto &t : v->timed) {
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, t.a, t.b));
    HIP_TRY(hipEventElapsedTime(&b, t.b2, t.c));
    v->ms[0] += a; v->ms[1] += b;
    v->ev_pool.push_back(t.a); v->ev_pool.push_back(t.b); v->ev_pool.push_back(t.b2); v->ev_pool.push_back(t.c);
  }
  v->timed.clear();
  if (ms2) { ms2[0] = v->ms[0]; ms2[1] = v->ms[1]; }
  if (launches2) { launches2[0] = v->launches[0]; launches2[1] = v->launches[1]; }
  if (reset) { v->ms[0] = v->ms[1] = 0; v->launches[0] = v->launches[1] = 0; }
  v->profiling = true; // (the first call switches the per-block events on)
  return RA_OK;
}

} // extern "C"
