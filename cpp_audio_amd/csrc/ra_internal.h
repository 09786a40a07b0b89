// Internal declarations shared by the host side (ra_capi.hip) and the gfx950 kernels.
// Data layout in HBM is described in DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/resynth_amd.h"

#define RA_MAX_CHANNELS 256
#define RA_MAX_FFT 16384
#define RA_ANALYSIS_THREADS 256
#define RA_N_FRAMES_PER_BUFFER 16 // audioelement.h:3
#define RA_LDS_LIMIT_BYTES (160 * 1024) // LDS a workgroup may ask for on gfx950 (MI355X_MICROARCH.md)

// ---- parameters every kernel reads (uniform; passed by value) -------------------------------------
struct RaParams {
  int sample_rate, N, H, L, M;       // M = L/2 complex points
  int log2M;
  int P, Ppad;                       // channels, padded to a multiple of 64
  int maxr;                          // pitch-list capacity per frame
  int n_streams;
  // step pipeline
  double min_volume, nearby, max_track;
  float tolerance, shift_pre, shift_post, harm_pre, harm_post;
  int use_autotune, autotune_type, n_allowed;
  double autotune_max_pitch, scale_root;
  double scale_pitches[8], scale_equi[7];
  float analysis_volume, input_gain, stereo_spread;
  // envelope (setAHDSR result, audioelement.h:923-952)
  int envA, envH, envD, envR;
  float SMinusOne;
  int phase_policy;
  uint64_t seed;
  double half_tone_ratio;
  // vocoder
  int voc_enabled, voc_S, voc_Wm, voc_Lm, voc_Mm, voc_Lc, voc_Mc, voc_bands;
  float vol_mod, vol_car, vol_voc;
  // the band-pass bank variant of the vocoder and the vocoder carrier synth
  int voc_kind;            // RA_VOCODER_FFT / RA_VOCODER_BANDPASS
  int car_P, car_Ppad;     // channels of the carrier synth (0 = none), padded to a multiple of 64
  int car_CH;              // channel slots per segment of the time-parallel synth: the power of two >= car_P (>= 4)
  int car_serial;          // 1 (reference_order 1): the sample-serial carrier synth kernel
  float car_vol[6];        // noise, saw, triangle, square, sine, pulse volume targets (lib.cpp:978-983)
  float car_pulse_width;   // lib.cpp:984
  int post_limit;          // post: "limit" (lib.cpp:1247-1258): the double mix goes through the limiter kernel before the float cast
  float fb_gain;   // output -> analysis feedback (lib.cpp:1626-1652): gain, and the delay in frames
  int fb_sz;
  int render_mode; // diagnostics: 0 regime renderer, 2 hop-parallel with the reference-order step, 3 with the cheap generic step
  int render_pack; // 1: two hops with at most 32 sounding channels each share a wave (ra_synth.hip: render_pair); 0 (reference_order 4): never
};
// the hop-parallel renderer applies the two centre gains after the channel sum (ra_synth_render_mono_kernel) and needs no per-channel gains
__host__ __device__ inline bool ra_hop_mono(const RaParams &p) { return p.render_mode == 0 && p.stereo_spread == 0.f; }

// ---- analysis -> tracking hand-off: one list per (stream, frame) ----------------------------------
// pitch/vol/key arrays are [n_streams][frames_cap][maxr]; count is [n_streams][frames_cap]

// ---- tracking -> synthesis hand-off: at most ONE command per (frame, channel) ---------------------
enum { RA_CMD_NONE = 0, RA_CMD_ON = 1, RA_CMD_OFF = 2, RA_CMD_CHANGE = 3 };
struct RaCmd {
  int32_t type;     // RA_CMD_*
  int32_t ord;      // position in the frame's one-shot order (simple_audio_context.h:74-76 runs them FIFO)
  float freq;       // Hz, already rounded to float like events.h:19-29
  float vel;
  int64_t noteid;
};

// ---- persistent tracking state, one per stream ----------------------------------------------------
struct RaPlayed {
  int64_t noteid;
  double pitch;
  int32_t channel;
  int32_t pad;
};
struct RaTrackState {
  int64_t next_noteid;
  int64_t dropped_note_on;
  int32_t n_played;
  int32_t pad;
};

// ---- persistent per-channel synth state: struct of arrays [n_streams][field][Ppad] -----------------
enum {
  SD_ENV_VALUE, SD_ENV_TOP, SD_CTRL_FROM, SD_CTRL_TO, SD_CTRL_CUR, SD_CTRL_C, SD_CTRL_FRES, SD_OSC_INC,
  SD_VOL_TARGET, SD_VOL_FINIT, SD_VOL_K, SD_VOL_Y,
  SD_THETA, // oscillator angle in rad/pi: authoritative in hop-parallel mode (cre/cim are derived from it), unused in reference order
  SD_HOP_MONO_COUNT, // the planes a hop's start state hands to the render kernel (seg_sd) when every channel is panned to the centre ...
  SD_GAIN_L = SD_HOP_MONO_COUNT, SD_GAIN_R,
  SD_HOP_COUNT,      // ... and with per-channel gains.  The rest is derived at the hop's start (chan_from_phase: phasor, rotation and
  SD_OSC_CRE = SD_HOP_COUNT, SD_OSC_CIM, SD_OSC_MRE, SD_OSC_MIM, SD_OSC_ALIAS, // anti-aliasing gain from theta and inc) or a copy (vol = y)
  SD_VOL,
  SD_COUNT
};
enum { SI_ENV_STATE, SI_ENV_COUNTER, SI_ENV_RELDELAY, SI_ENV_AHDCOUNTER, SI_ENV_MINCH, SI_ENV_NEXTMINCH,
       SI_HOP_COUNT, // a hop's start state (seg_si) ends here: its state word also carries the AHD state and the flags (ra_hop_word)
       SI_ENV_AHDSTATE = SI_HOP_COUNT, SI_FLAGS, SI_COUNT };
// seg_si's state word: envelope state (ENV_*, bits 0-3), AHD state + 1 (bits 4-7), flags (bits 8 up)
__host__ __device__ inline int ra_hop_word(int state, int ahdstate, int flags) { return state | ((ahdstate + 1) << 4) | (flags << 8); }
__host__ __device__ inline int ra_hop_state(int w) { return w & 15; }
enum { FL_HAS_FRES = 1, FL_HAS_VOL = 2, FL_HAS_TARGET = 4,
       FL_STEPPED0 = 8 }; // hop-start states only (seg_si): the propagate kernel already took this hop's first step for the channel
enum { ENV_SOON = 0, ENV_PRESSED = 1, ENV_RELEASED = 2, ENV_DONE1 = 3, ENV_DONE2 = 4 }; // audioelement.h:157-163

enum { RA_ERR_OVERFLOW = 1, RA_ERR_BOOKKEEPING = 2 };

// device buffers of one context
constexpr int RA_VOC_BIG_WORKGROUPS = 1024; // workgroups of a vocoder launch whose transform buffers live in HBM (ra_vocoder.hip)

struct RaBuffers {
  // constant tables
  double *half_window;      // [N/2]
  double2 *tw_half;         // [M/2]  e^{-2 pi i k / M}
  double2 *tw_full;         // [M+1]  e^{-2 pi i k / L}
  int *bitrev;              // [M]
  double *loud_pitches;     // [29]
  float *loud_elv;          // [29] at 60 phon
  double *allowed;          // [n_allowed]
  // stream state
  float *hist;              // [n_streams][N] last N input samples before the current block
  RaTrackState *track;      // [n_streams]
  RaPlayed *played;         // [n_streams][P]
  int64_t *chan_free_at;    // [n_streams][P]
  int32_t *chan_held;       // [n_streams][P]
  int32_t *chan_minch;      // [n_streams][P]
  double *sd;               // [n_streams][SD_COUNT][Ppad]
  int32_t *si;              // [n_streams][SI_COUNT][Ppad]
  int32_t *error_flags;     // [1]
  // per-block scratch (grown on demand)
  int64_t frames_cap;
  double *pl_pitch, *pl_vol; // [n_streams][frames_cap][maxr]
  float *pl_key;
  int32_t *pl_count;        // [n_streams][frames_cap]
  // per item, derived from the list alone by ra_track_prep_kernel (frame-parallel) so that the tracking chain does not compute them:
  float *pl_nf, *pl_nv;     // the event payload: frequency and velocity as floats (events.h:19-29)
  int32_t *pl_nmc;          // min change duration of that frequency
  int32_t *pl_ord;          // the frame's loudness order: item index by descending key, ties in index order
  RaCmd *cmd;               // [n_streams][frames_cap][Ppad]
  int32_t *cmd_non;         // [n_streams][frames_cap] number of note-on commands of the frame (their RaCmd.type carries the rank << 8)
  double *voc_mix;          // [n_streams][n] when the vocoder is enabled
  int64_t mix_cap;
  double *seg_sd;           // [n_streams][seg_cap][SD_HOP_COUNT][Ppad] channel state at the start of every segment of the block
  int32_t *seg_si;          // [n_streams][seg_cap][SI_HOP_COUNT][Ppad]
  int64_t seg_cap;
  // ---- output -> analysis feedback (only allocated when its gain is non-zero) ----
  float *fb_hist;           // [n_streams][fb_hist_len] mono output float(L + R) of the fb_hist_len frames before the current block
  float *fb_block;          // [n_streams][n] mono output of the current block, written by the render as it goes
  int64_t fb_hist_len, fb_block_cap;
  double *dbg_sqmag;        // optional [frames][M] tap of |X|^2 (ra_sqmag), normally null
  unsigned long long *dbg_clk; // optional [128] in-kernel phase clocks (env RA_PHASE_CLOCKS=1, printed by ra_destroy), normally null
  unsigned long long *dbg_stamp; // optional [16] clock stamps of the two throughput kernels (ra_set_clock_stamps, RaClockStamp), normally null
  // ---- post: "limit" ----
  double *post_d;           // [n_streams][n][2] the double mix of the block, read by the limiter kernel
  int64_t post_cap;
  double *lim_state;        // [n_streams][2] compression, target
  // ---- vocoder carrier synth (only with carrier_n_channels > 0 and a vocoder volume non-zero) ----
  double *car_sd;           // [n_streams][SD_COUNT][car_Ppad] envelope / control / sine oscillator / outer volume, like sd
  int32_t *car_si;          // [n_streams][SI_COUNT][car_Ppad]
  double *car_ext;          // [n_streams][CX_COUNT][car_Ppad] the phase-controlled oscillators, the noise voice's index
  double *car_noise;        // [car_noise_len] the white-noise sound buffer (getWhiteNoise, source/sound.cpp:3-21), null while the noise volume is 0
  int32_t car_noise_len;
  int64_t *car_noteid;      // [n_streams][car_Ppad] note id each channel last played (-1 = never used), gen.crtp.h:419
  int32_t *car_dropped;     // [n_streams] note-ons dropped for want of a free channel
  struct RaCarEv *car_ev;   // the block's note events, stream after stream, in queue order
  int32_t *car_ev_begin;    // [n_streams + 1]
  int64_t car_ev_cap;
  double *car_gen_d;        // [n_streams][n] the generated carrier as the double the vocoder compute receives ...
  float *car_gen_f;         // ... and as the float PeriodicFFT::feed stores (periodicfft.cpp:55-61)
  int64_t car_gen_cap;
  // the time-parallel carrier synth (ra_carrier.hip): the block cut into segments at the events and every 256 samples
  int4 *car_seg_info;       // [segments] stream, first sample (block-relative), length (0: unused slot), does anything sound
  double *car_seg_d;        // [segments][CS_COUNT][car_CH] channel state at the start of the segment
  int32_t *car_seg_i;       // [segments][SI_COUNT][car_CH]
  int64_t car_seg_cap;
  // ---- band-pass bank (vocoder_kind = RA_VOCODER_BANDPASS) ----
  double *bank_k;           // [3][64] one-pole coefficients per band: high-pass (band's lower edge), low-pass (upper edge), envelope follower
  double *bank_state;       // [n_streams][5][64] filter states: modulator hp / lp, envelope follower, carrier hp / lp
  double *bank_pow;         // [3][8][64] per coefficient of bank_k and band: (1 - k)^(16 * 2^e), e = 0..5 — the ratios of the time-parallel bank's scans (ra_voc_bank_scan_kernel)
  // ---- vocoder (only allocated when a vocoder volume is non-zero) ----
  double2 *voc_big;         // transform buffers in HBM for frames that do not fit LDS: [RA_VOC_BIG_WORKGROUPS + n_streams][voc_big_slot], else null
  int64_t voc_big_slot;
  double *voc_roll_tmp;     // [n_streams][voc_roll_slot] staging of the history roll when it does not fit LDS, else null
  int64_t voc_roll_slot;
  double *voc_half_window;  // [Wm/2] Gaussian 4 sigma
  double2 *vm_tw_half, *vm_tw_full; int *vm_bitrev;   // modulator FFT, length Lm
  double2 *vc_tw_half, *vc_tw_full; int *vc_bitrev;   // carrier FFT, length Lc
  int *voc_mod_band_start;  // [2*bands] modulator bins of band b are [v[2b], v[2b+1])
  int *voc_mod_bin_band;    // [Mm + 1] band of every modulator bin: -1 below the first band, `bands` above the last one and for the Nyquist bin (non-decreasing)
  int *voc_car_band;        // [Lc/2+1] band of each carrier bin, -1 = outside every band
  double *voc_xf_new;       // [S] EqualGainXFade new-signal gain for new_signal_idx = 0..S-1
  float *voc_mod_hist;      // [n_streams][Wm] last Wm voice samples
  float *voc_car_hist;      // [n_streams][2S] last 2S carrier samples
  double *voc_amp_prev;     // [n_streams][bands] band amplitudes of the last modulator frame before this block
  double *voc_sig_prev;     // [n_streams][2][2S] the two carrier frames before this block (older first)
  double *voc_amp;          // [n_streams][voc_mod_cap][bands]
  double *voc_sig;          // [n_streams][voc_car_cap][2S]
  int64_t voc_mod_cap, voc_car_cap;
};

// in-kernel phase timing for tuning: lane 0 of the workgroup adds the s_memtime ticks since the previous mark to slot i.
// lap<I>() accumulates in registers instead (inner loops); flush() adds those totals and lap counts to slots base + I / base + 4 + I.
struct RaPhaseClock {
  unsigned long long *dst, t, acc[4];
  unsigned laps[4], cnt[4];
  __device__ inline RaPhaseClock(unsigned long long *d, bool on) : dst(on ? d : nullptr), t(0) {
    for (int i = 0; i < 4; ++i) { acc[i] = 0; laps[i] = 0; cnt[i] = 0; }
    if (dst) t = __builtin_readcyclecounter();
  }
  __device__ inline void mark(int i) {
    if (!dst) return;
    unsigned long long now = __builtin_readcyclecounter();
    atomicAdd(dst + i, now - t);
    t = now;
  }
  template <int I> __device__ inline void lap() {
    if (!dst) return;
    unsigned long long now = __builtin_readcyclecounter();
    acc[I] += now - t; ++laps[I];
    t = now;
  }
  __device__ inline void count(int i) { if (dst) atomicAdd(dst + i, 1ull); }
  template <int I> __device__ inline void tally() { if (dst) ++cnt[I]; }
  __device__ inline void flush(int base) {
    if (!dst) return;
    for (int i = 0; i < 4; ++i) if (laps[i]) { atomicAdd(dst + base + i, acc[i]); atomicAdd(dst + base + 4 + i, (unsigned long long)laps[i]); }
    for (int i = 0; i < 4; ++i) if (cnt[i]) atomicAdd(dst + base + 8 + i, (unsigned long long)cnt[i]);
  }
};

// The clock a kernel's waves actually ran at (MI355X_MICROARCH.md, DVFS give-back item 6): s_memtime ticks are shader cycles, s_memrealtime
// ticks a constant 100 MHz; lane 0 of a workgroup stamps both once at its start and once at its end and adds the two differences and a count
// to dst[slot .. slot + 2].  Sum of shader ticks / sum of 100 MHz ticks x 0.1 = the duration-weighted mean clock in GHz over the stamped
// workgroups.  Only in the DIAGNOSTIC build (-DRA_CLOCK_STAMPS=1, `make clk` -> ../libresynth_amd_clk.so): the four live registers of a
// stamp cost the analysis kernel 77 more spilled VGPRs at its 128-register budget and the renderer 29; in the product build the type is
// empty and no stamp exists.
#ifndef RA_CLOCK_STAMPS
#define RA_CLOCK_STAMPS 0
#endif
#if RA_CLOCK_STAMPS
struct RaClockStamp {
  unsigned long long *dst, t0, r0;
  __device__ inline RaClockStamp(unsigned long long *d, bool on) : dst(on ? d : nullptr), t0(0), r0(0) {
    if (dst) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  }
  __device__ inline void finish(int slot) {
    if (!dst) return;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    atomicAdd(dst + slot, t1 - t0); atomicAdd(dst + slot + 1, r1 - r0); atomicAdd(dst + slot + 2, 1ull);
  }
};
#else
struct RaClockStamp {
  __device__ inline RaClockStamp(unsigned long long *, bool) {}
  __device__ inline void finish(int) {}
};
#endif

// Runs fn the first time the calling site is reached on the current device (function attributes are per device: a process may hold
// contexts on several GPUs).  `done` is the site's own static bit set; the lock keeps a second host thread from launching before the
// attributes are in place.
#include <mutex>
template <typename Fn>
inline void ra_once_per_device(unsigned long long &done, Fn fn) {
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  int d = 0;
  (void)hipGetDevice(&d);
  const unsigned long long bit = 1ull << (d & 63);
  if (!(done & bit)) { fn(); done |= bit; }
}

// frames of the block [t0, t0+n) for a PeriodicFFT fed from sample 0 with window W and stride S: fire = W - 1 + j S
struct RaFrameRange { int64_t lo, count; };
__host__ __device__ inline RaFrameRange ra_frames_firing(int64_t W, int64_t S, int64_t t0, int64_t n) {
  RaFrameRange r;
  int64_t a = t0 - (W - 1);
  r.lo = a <= 0 ? 0 : (a + S - 1) / S;
  int64_t bnd = t0 + n - 1 - (W - 1);
  if (bnd < 0) { r.count = 0; return r; }
  int64_t hi = bnd / S;
  r.count = hi >= r.lo ? hi - r.lo + 1 : 0;
  return r;
}
// the AHDSR stage lengths after setAHDSR (audioelement.h:923-952); uniform for a launch
struct RaEnvParams { int A, H, D, R; float SMinusOne; };
// one note of the main.resynth driver, already assigned to a pool slot (resynth.hpp:83-96)
struct RaNoteEv {
  int64_t start;   // sample at which the note starts (startFrame * stride)
  double inc;      // freq_to_angle_increment<double>(frequency, sampling_rate)
  double vol;      // DbToMag(amplitude)
  int32_t delay;   // onKeyReleased(stride + endFrame - startFrame)
  int32_t pad;
};
void ra_launch_notes(const RaNoteEv *d_ev, const int *d_slot_begin, int n_slots, RaEnvParams ep, int64_t T, double *d_partial, double *d_out,
                     double2 *d_phase_io, hipStream_t st);
// carrier_d: the carrier as doubles when it comes from the carrier synth (then d_carrier is its float image), else null
void ra_launch_vocoder(const RaParams &p, const RaBuffers &b, const float *d_voice, const float *d_carrier, const double *carrier_d, int64_t t0, int64_t n,
                       hipStream_t st);
// one note event of the carrier synth as the device sees it (the host has resolved key -> note id, events.h:228-281)
struct RaCarEv {
  int64_t sample;  // output frame whose step runs the event's one-shot
  int64_t noteid;
  int32_t type;    // 0 NoteOn, 1 NoteOff
  int32_t delay;   // frames handed to onKeyPressed / onKeyReleased
  float freq, vel;
};
enum { CX_PH_ANGLE, CX_PH_INC, CX_PH_ALIAS, CX_NZ_INDEX /* the noise voice's buffer index, an integer held as a double */, CX_COUNT };
void ra_launch_limiter(const RaParams &p, const RaBuffers &b, int64_t r0, int64_t r1, int64_t n, float *d_out, hipStream_t st);
void ra_launch_carrier_synth(const RaParams &p, const RaBuffers &b, int64_t t0, int64_t n, int64_t total_segs, hipStream_t st);
int ra_carrier_seg_doubles(); // doubles of channel state per (segment, channel slot) in car_seg_d
void ra_launch_voc_bank(const RaParams &p, const RaBuffers &b, const float *d_voice, const float *d_carrier, const double *carrier_d, int64_t t0, int64_t n,
                        hipStream_t st);

// one time-chunk of a ra_process_* block: the block gives I/O indexing and the history split, the chunk the work range
struct RaSpan {
  int64_t t0, n;      // the block: absolute index of its first sample, length
  int64_t ct0, ct1;   // this chunk's sample range [ct0, ct1)
  int64_t f_lo;       // first analysis frame firing in the block
  int64_t fb0, nfb;   // this chunk's frames: block-relative index of the first one, count (scratch is indexed block-relative)
  int64_t seg0;       // first slot of this chunk's nfb + 1 segments in seg_sd / seg_si
};

// ---- gen.voice (ra_voice.hip; the host half is ra_voice_host.hpp) ----
struct RaVoiceParams {
  int sample_rate;
  float sine_gain, log_ratio, max_db;
  float db_to_log2;                   // log_ratio / 20 * log2(10): 10^(log_ratio dB / 20) = 2^(db_to_log2 dB)
  float loud_freqs[29], loud_elv[29]; // include/loudness.h:9-39 and the equal-loudness row of the configured level (:199-216)
};
// One ramp of one note inside one block (a job): the LogRamp spec, the output factors, the envelope's three integers, where its samples
// lie in the block, and the state slots the two kernels exchange.
struct RaVoiceJob {
  double from, to, dur, cur0;   // LogRamp::setup(from, to, duration, start sample)
  double vol, fL, fR;           // the ramp's volume (spec volume); Element::baseVolume * velocity * the note's stereo gains
  double C, l2r;                // LogRamp's proportionality constant (audioelement.h:2560-2571); log2(to / from) for PROPORTIONAL_VALUE_DERIVATIVE, else 0
  int32_t begin, n;             // block-relative frame of the first step taken in this block; steps taken in this block
  int32_t j0;                   // index (1-based) of that first step within the ramp
  int32_t T, A, R;              // steps until the key release; attack and release lengths
  int32_t interp;               // ra_itp
  int32_t state_in;             // slot (previous block's buffer) of the state before step j0, -1: j0 == 1
  int32_t state_tiles;          // first slot (this block's buffer) of the states at the tile boundaries the job crosses
  int32_t state_out;            // slot (this block's buffer) for the state after the block's last step, -1: the job ends in this block
  int32_t stream, slot;
};
struct RaVoiceState { double cur, theta, y; int32_t swapped, loud_k; };
struct RaVoiceSeg { double dtheta, a, b; int32_t job, loud_k; }; // pre-pass scratch, one per state slot: what the slot's segment adds
void ra_voice_launch_pre(const RaVoiceParams &p, const RaVoiceJob *jobs, const int32_t *order, int n_work, int n_slots, const RaVoiceState *st_prev, RaVoiceState *st_cur,
                         RaVoiceSeg *seg, int L, hipStream_t st);
int ra_voice_launch_render(const RaVoiceParams &p, const RaVoiceJob *jobs, const int32_t *slot_begin, const RaVoiceState *st_prev, const RaVoiceState *st_cur, int n_streams,
                           int64_t n, int L, int g_log2, float *d_out, hipStream_t st);

// ---- gen.voice WIND (K10, ra_wind.cuh inside ra_voice.hip) ----
// A WIND ramp: FreqCtrl_<Mix<low-pass noise, band-pass noise, band-reject noise, sine>, ShortTermNoiseAdderCtrl> (soundengine.h:194-230).
// Everything a kernel needs that is the same for every ramp of the voice object:
struct RaWindParams {
  int order;                     // ORDER_FILTERS: sections per filter (Filter<.., VariableOrder>::setOrder)
  int pink_len, pink_start;      // the pink-noise buffer's length; the noise wrappers' index before the first step (setStartAngle(0): length / 2 - 1 after rounding)
  float g_bp, g_br, g_sine;      // Mix gains (the low-pass voice's gain is 0: refused otherwise)
  float width_min, width_max;    // BandAlgo_::width_range (octaves)
  float log_inc_min, log_inc_max, inv_approx_range; // SoundEngineFreqCtrl: the same range and abs mean for the outer control and the two bands
  int n_long, n_band;            // slow steps of the outer control's long-term iterator (int(n)) and of the bands' four iterators (int32(unsigned(n)))
  float short_rate, short_amplitude; // ShortTermNoiseAdderCtrl::ratio (the short-term step count as a float) and noise_amplitude
  int interp_long;               // the long-term iterator's interpolation (ra_itp): the engine's; every other iterator is linear
};
struct RaWindIter { int32_t pos, slow, n; float prev; }; // a SlowIter over the pink buffer: BufferIter position, slow_it, n_steps, prev
struct RaWindState {             // what a playing ramp carries from chunk to chunk besides its filters
  RaWindIter it[6];              // long-term, short-term, band-pass centre / width, band-reject centre / width
  double theta, y;               // the sine's phase sum (units of pi) and its loudness volume
  int32_t loud_k, pad;
};
struct RaWindJob {               // one playing ramp inside one block
  double vol, fL, fR;            // the spec's volume; Element::baseVolume * velocity * the note's stereo gains
  int32_t begin, n, j0;          // block-relative first frame, frames, 1-based index of that first step within the ramp
  int32_t T, A, R;               // the envelope's three integers
  int32_t stream, slot;          // slot: where the ramp's state lives ((stream * n_channels + channel) * 3 + ramp)
  int32_t pos[6][2];             // the declared start positions of the six iterators: initializeForRun's, set_sample_rate's
};
struct RaWindScratch {           // per chunk: [job][chunk frames] each; what is not needed is not allocated (nullptr); what the controls write exists twice (chunk k + 1 beside chunk k)
  double *k_bp_lp, *k_bp_hp, *k_br_lp, *k_br_hp; // the one-pole coefficients per sample
  double *p_bp, *p_br_lp, *p_br_hp, *p_sine;     // the voices' outputs per sample
  int64_t stride;                // frames per job
};
void ra_wind_launch_controls(const RaVoiceParams &vp, const RaWindParams &wp, const RaWindJob *jobs, int n_jobs, const double *pink, RaWindState *states, const RaWindScratch &sc, int c0,
                             int clen, hipStream_t st);
void ra_wind_launch_chains(const RaWindParams &wp, const RaWindJob *jobs, int n_jobs, const int32_t *stream_begin, int n_streams, const double *pink, double *fstates,
                           const RaWindScratch &sc, int c0, int clen, int64_t n, float *d_out, hipStream_t st);

// Profiling (ra_set_profiling): the two throughput kernels are timed by the launch itself — hipExtLaunchKernelGGL stamps a start and a stop event with
// the dispatch's own begin and end, the interval rocprofv3 reports — instead of by an event pair around the launch, which also spans the wait of
// the launch's queue for its turn beside the other queues (measured: the renderer's 1.33 ms read as 1.42).  The host sets the pair before it calls
// a launcher; a launcher that takes it says so.
struct RaExtTiming { hipEvent_t a, b; bool used; };
extern thread_local RaExtTiming *ra_ext_timing;

// launchers (defined next to their kernels)
void ra_launch_analysis(const RaParams &p, const RaBuffers &b, const float *d_voice, int64_t t0, int64_t n, int64_t f_lo, int64_t fb0, int64_t nfb_chunk,
                        int64_t nfb_total, hipStream_t st);
void ra_launch_track(const RaParams &p, const RaBuffers &b, int64_t f_lo, int64_t fb0, int64_t n_frames_blk, hipStream_t st);
void ra_launch_track_prep(const RaParams &p, const RaBuffers &b, int64_t fb0, int64_t n_frames_blk, hipStream_t st); // after the analysis of those frames
void ra_launch_synth_serial(const RaParams &p, const RaBuffers &b, const RaSpan &sp, float *d_out, double *d_out_d, hipStream_t st);
void ra_launch_synth_propagate(const RaParams &p, const RaBuffers &b, const RaSpan &sp, hipStream_t st);
void ra_launch_synth_render(const RaParams &p, const RaBuffers &b, const RaSpan &sp, float *d_out, double *d_out_d, hipStream_t st);
void ra_launch_roll_history(const RaParams &p, const RaBuffers &b, const float *d_voice, int64_t n, hipStream_t st);
void ra_launch_roll_feedback(const RaParams &p, const RaBuffers &b, int64_t n, hipStream_t st);
size_t ra_analysis_lds_bytes(const RaParams &p);
size_t ra_track_lds_bytes(const RaParams &p);

// frame f covers input samples [2 + f H, 2 + f H + N) and fires in iteration N + 1 + f H
__host__ __device__ inline int64_t ra_fire_sample(int N, int H, int64_t f) { return (int64_t)N + 1 + f * (int64_t)H; }
