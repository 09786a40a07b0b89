/*
 * resynth_amd — MI355X (gfx950) implementation of cpp.audio's rt.resynth analysis -> resynthesis hot path.
 *
 * C ABI of libresynth_amd.so.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 * The reference (OlivierSohn/cpp.audio) has no FFI for this path: its components meet through C++
 * templates and two type-erased callables.  Each entry point below names the
 * reference interface it stands in for (paths relative to the reference checkout); INTEGRATION.md
 * shows the few lines of C++ a maintainer adds on the reference side to bind them.
 *
 * Threading: one ra_context is used from one host thread at a time (the reference's audio thread /
 * offline-job constructor thread, source/rt.resynth.lib.cpp:1185-1235).
 * Errors: every call returns RA_OK or a negative RA_E* code; ra_last_error() gives the text.  The C++
 * shims in cpp_audio_amd/host/ turn codes back into the reference's exceptions / bool conventions.
 */
#ifndef RESYNTH_AMD_H
#define RESYNTH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RA_OK 0
#define RA_EINVAL -1      /* bad argument / config (reference: std::invalid_argument, lib.cpp:449-481) */
#define RA_ENODEVICE -2   /* no usable HIP device: there is NO CPU fallback */
#define RA_EHIP -3        /* a HIP runtime call failed */
#define RA_EOVERFLOW -4   /* a per-frame list exceeded its capacity (see ra_config.max_pitches_per_frame) */
#define RA_EINTERNAL -5   /* device-side consistency check failed (decision/synth channel bookkeeping disagree), or the context is
                             poisoned: a block failed after it had started to advance the streams (a HIP error midway); every entry
                             point then returns RA_EINTERNAL until ra_reset.  A block that fails before that point (bad argument,
                             allocation failure) leaves the streams untouched and may be retried. */

enum { RA_AUTOTUNE_SCALE = 0, RA_AUTOTUNE_CHORD = 1, RA_AUTOTUNE_INTERVALS = 2 };     /* rt.resynth.lib.autotune.cpp:3-8   */
enum { RA_SCALE_MAJOR = 0, RA_SCALE_MINOR_NATURAL = 1, RA_SCALE_MINOR_HARMONIC = 2 }; /* rt.resynth.lib.autotune.cpp:9-14  */
enum { RA_CHORD_SINGLE = 0, RA_CHORD_HARMONICS = 1, RA_CHORD_OCTAVES = 2 };           /* rt.resynth.lib.autotune.cpp:16-20 */
enum { RA_WINDOW_GAUSSIAN4 = 0, RA_WINDOW_HANN = 1, RA_WINDOW_RECT = 2 };             /* periodicfft.cpp:246-249 (+Hann, main.test_fft.cpp:79) */
enum { RA_PHASE_HASH = 0, RA_PHASE_ZERO = 1 };                                        /* gen.crtp.h:147-155 DefaultStartPhase */
/* the live FFT vocoder (vocoder.cpp:100-252,396-551) / the band-pass bank it replaced (vocoder.cpp:46-79,310-394,667-732: the members
 * commented out at :846-851; "32-band biquad bank" of BASELINE configs[2]) */
enum { RA_VOCODER_FFT = 0, RA_VOCODER_BANDPASS = 1 };
enum { RA_POST_NONE = 0, RA_POST_LIMIT = 1 };                                         /* Postprocessing, rt.resynth.lib.params.cpp:152-176 */

/*
 * The parameter block of one RtResynth (the std::atomic<> members of source/rt.resynth.lib.cpp:943-999,
 * same names, same defaults via ra_default_config).  Every stream of a context shares it.
 * window_size / stride are in frames; ra_even_window_size_frames / ra_window_center_stride_frames
 * reproduce the reference's seconds -> frames rules (lib.cpp:484-487, 1577-1578).
 */
typedef struct ra_config {
  int32_t sample_rate;
  int32_t window_size;          /* N, even; FFT length = next power of two >= N (<= 16384) */
  int32_t stride;               /* H */
  int32_t n_channels;           /* P = 2 * nVoices, the partial cap (gen.crtp.h:225); <= 256 */
  int32_t window_kind;          /* RA_WINDOW_*; the reference analyses with RA_WINDOW_GAUSSIAN4 */

  float min_volume;
  float nearby_distance_tones;
  float max_track_pitches;
  float autotune_tolerance_pitches;
  float pitch_shift_pre_autotune, pitch_shift_post_autotune;
  float pitch_harmonize_pre_autotune, pitch_harmonize_post_autotune;
  float stereo_spread;

  float env_attack_seconds, env_hold_seconds, env_decay_seconds, env_release_seconds, env_sustain_level;

  int32_t use_autotune;
  int32_t autotune_type;
  int32_t autotune_max_pitch;
  int32_t autotune_factor;
  int32_t autotune_scale_mode;
  int32_t autotune_root_note;   /* 0 = Do .. 11 = Si (include/midi.h:11-24) */
  int32_t autotune_root_transpose;
  int32_t autotune_chord_frequencies;
  uint64_t autotune_bit_chord;

  float vocoder_modulator_window_size_seconds;
  float vocoder_stride_seconds;
  int32_t vocoder_count_bands;  /* <= 50 (vocoder.cpp:98) */
  float vocoder_min_freq, vocoder_max_freq;
  float voice_volume, carrier_volume, vocoder_volume, analysis_volume;

  float analysis_input_gain;

  /* Declared policy for the reference's two unpinned random draws (start phase gen.crtp.h:152, pan lib.cpp:116):
   * a counter hash of (seed + stream index, note id). */
  int32_t phase_policy;
  uint64_t seed;

  /* capacity of the per-frame pitch list handed from analysis to tracking (the reference's vectors grow
   * on demand, lib.cpp:1102-1112); exceeding it sets RA_EOVERFLOW instead of truncating silently. */
  int32_t max_pitches_per_frame; /* default 512 (the fastest: the lists live in LDS beside the FFT buffer); 0 = sized for the worst case,
                                    min(FFT length / 4, what fits) — no spectrum can overflow it up to FFT length 8192; 1..4096, further limited by the 160 KB of LDS a workgroup may hold: the analysis keeps
                                    two lists of this capacity (twice that with a harmonize step) beside its FFT buffer, the tracking
                                    40 bytes per entry — ra_create returns RA_EINVAL with the byte counts when they do not fit */

  /* 0 (default): synthesis runs hop-parallel — per-hop channel states come from closed-form jumps, differing from
   * sample-by-sample stepping at rounding level (~1e-13).  1: one sample-serial launch per stream that follows the
   * reference's operation order exactly (slow; for bit-level comparisons).  2 / 3 / 4 are tuning aids: hop-parallel with the general step
   * on every sample (2: reference-order step, 3: the cheap generic step), and 4: the product renderer without the packing of two hops
   * into one wavefront (which changes the order of the channel sum, i.e. the last bits). */
  int32_t reference_order;

  /* Feedback of the output into the analysis input (analyze_until_input_starvation, lib.cpp:1626-1652): analysis input k =
   * analysis_input_gain * voice[k] + analysis_output_feedback_gain * mono_out[k - 1 - delay], delay = int(0.5 + output_delay_seconds *
   * sample_rate) frames, mono_out = float(L + R) of the final mix.  With a non-zero gain a block is processed in sequential pieces of
   * delay / stride + 1 analysis frames (the analysis of a piece needs the output of the pieces before it): correct for any delay, fast
   * only for delays of many strides (the reference default is 1 s).  Available with reference_order 0 and 1. */
  float analysis_output_feedback_gain; /* 0 */
  float output_delay_seconds;          /* 1 */

  int32_t vocoder_kind;                    /* RA_VOCODER_* */
  float vocoder_env_follower_cutoff_ratio; /* 1/20 (lib.cpp:985): envelope follower cut-off = ratio * the band's lower edge; bank only */
  /*
   * The vocoder carrier synth (SynthVocoderCarier, lib.cpp:212-221: VocoderCarrierElement, :21-52, initialised by
   * VocoderCarrierElementInitializer, :137-196).  carrier_n_channels = 2 * nVoices (reference: 254), 0 = no synth.  The vocoder takes its
   * carrier from the synth when ra_process_* is given no carrier buffer (the reference: realtime mode the synth, offline job the file,
   * lib.cpp:1405-1412); notes come from ra_carrier_events.  The noise voice (soundBufferWrapperAlgo<Sound::NOISE>, audioelement.h:1507-1583)
   * plays the white-noise sound buffer of source/sound.cpp:3-21 — int(sample_rate / .05f) frames of white_gaussian_noise (include/noise.h:3-5),
   * one buffer for all streams like the reference's process-wide static; the draws behind it (std::normal_distribution<float> over an
   * OS-seeded engine) are this build's declared policy, DESIGN.md §5.  The buffer exists only while vocoder_carrier_noise_volume != 0.
   */
  int32_t carrier_n_channels; /* 0..256 */
  float vocoder_carrier_noise_volume, vocoder_carrier_saw_volume, vocoder_carrier_triangle_volume, vocoder_carrier_square_volume,
      vocoder_carrier_sine_volume, vocoder_carrier_pulse_volume, vocoder_carrier_pulse_width; /* lib.cpp:978-984: 0 0 0 1 0 0 0.01 */

  /* RA_POST_*: the job config's "post" (RtResynth::init_post, lib.cpp:1245-1275).  RA_POST_LIMIT runs every output frame through the
   * limiter before the mono publication and the float cast; the Limiter itself is cpp.algorithms code that is absent here: its
   * behaviour is this build's own declared decision (parity unpinned).  ra_compute (the ComputeFunc boundary) is never
   * post-processed: post-processing belongs to SimpleAudioOutContext::step, after the computes. */
  int32_t post;
} ra_config;

/*
 * One note event of the vocoder carrier synth: what RtResynth::onInputMidiEvent (lib.cpp:1518-1545) hands to
 * SynthVocoderCarier::onEvent for a MIDI note-on / note-off, minus the MIDI device.  `sample` is the output frame whose
 * SimpleAudioOutContext::step runs the event's one-shot (simple_audio_context.h:74-76) — the event then acts on that very frame — and
 * `delay` the frames the one-shot passes on to onKeyPressed / onKeyReleased (gen.crtp.h:506-512,556-566; 0 = the "late" branch).
 * key: the NoteIdsGenerator key (events.h:228-281): a note-off releases the EARLIEST pending note-on of its key.
 */
typedef struct ra_carrier_event {
  int64_t sample;
  int32_t stream;
  int32_t type;      /* 0 NoteOn, 1 NoteOff */
  int32_t key;       /* MIDI key */
  int32_t delay;
  float frequency;   /* Hz; <= 0: Midi::midi_pitch_to_freq(key) (midi.h:258-264), as onInputMidiEvent does */
  float velocity;    /* on.velocity / 127.f */
} ra_carrier_event;

typedef struct ra_context ra_context;

/* One event as it reached the synth (ImplCRTP::onEvent, include/gen.crtp.h:382-622), for inspection/tests. */
typedef struct ra_event {
  int64_t sample;   /* output frame at which it took effect */
  int32_t type;     /* 0 NoteOn, 1 NoteOff, 2 NoteChange */
  int32_t channel;
  int64_t noteid;
  float frequency, velocity;
} ra_event;

typedef struct ra_stats {
  int64_t frames_analysed;      /* analysis frames fired, all streams */
  int64_t samples_rendered;     /* per stream */
  int64_t dropped_note_on;      /* RtResynth::countDroppedNoteOns, lib.cpp:549-551, all streams */
  int32_t error_flags;          /* bit0 pitch-list overflow, bit1 channel bookkeeping mismatch */
  int32_t reserved;
  int64_t carrier_dropped_note_on; /* carrier synth note-ons that found no free channel (onDroppedNote, gen.crtp.h:624-627), all streams */
} ra_stats;

/* A note for the main.resynth driver: DeducedNote<double> as used by include/resynth.hpp:58-61,84-96. */
typedef struct ra_note {
  int32_t start_frame, end_frame; /* in hop units, like the reference (multiplied by stride inside) */
  double frequency;               /* Hz */
  double amplitude_db;
} ra_note;

const char *ra_last_error(void);
int ra_device_count(void);

/* defaults of source/rt.resynth.lib.cpp:943-999 */
int ra_default_config(ra_config *cfg);
/* RtResynth::getEvenWindowSizeFrames (lib.cpp:484-487) and the stride lambda (lib.cpp:1577-1578) */
int ra_even_window_size_frames(float window_size_seconds, int sample_rate);
int ra_window_center_stride_frames(float stride_seconds, int sample_rate);

/*
 * Replaces constructing n_streams RtResynth objects in offline mode (RtResynth(RtResynthOfflineJob&),
 * lib.cpp:1185-1196: restorePreset + init()).  device = HIP device ordinal.
 */
int ra_create(const ra_config *cfg, int32_t n_streams, int32_t device, ra_context **out);
void ra_destroy(ra_context *ctx);
/* back to the state right after ra_create (RtResynth::init after teardown, lib.cpp:1305-1307) */
int ra_reset(ra_context *ctx);
/* launch on this hipStream_t (NULL = the default stream).  Lets a host framework order us on its own stream. */
int ra_set_hip_stream(ra_context *ctx, void *hip_stream);

/*
 * Replaces the body of the offline loop lib.cpp:1204-1234 for all streams at once: for every frame,
 * input_f (voice sample -> analysis queue, :1208), analyze_until_input_starvation (:1211),
 * output_f = SimpleAudioOutContext::step (:1215-1217, include/simple_audio_context.h:69-112), write_output (:1220).
 *
 * d_voice / d_carrier: DEVICE pointers, [n_streams][n_frames] float32 mono (either may be NULL = silence;
 * RtResynthOfflineJob::read_voice / read_carrier, rt.resynth.lib.params.cpp:291-313).
 * d_out: DEVICE pointer, [n_streams][n_frames][2] float32 interleaved (write_output, params.cpp:320-324).
 * Consecutive calls continue the same streams (block-size independent results).  Asynchronous on the
 * context's HIP stream; ra_synchronize waits and reports device-side errors.
 */
int ra_process_device(ra_context *ctx, const float *d_voice, const float *d_carrier, int64_t n_frames, float *d_out);
/* same with HOST buffers (does H2D, process, D2H, synchronises) */
int ra_process_host(ra_context *ctx, const float *h_voice, const float *h_carrier, int64_t n_frames, float *h_out);
int ra_synchronize(ra_context *ctx);
/*
 * ra_process_host for long blocks fed from the host: the block is cut into chunks of chunk_frames (e.g. one second); chunk k + 1's input is
 * copied in and chunk k - 1's output copied out while chunk k runs — three queues, two sets of chunk-sized device buffers — so that a block
 * costs max(link, kernels) instead of their sum.  The chunks are ordinary consecutive blocks: results equal ra_process_host's up to the
 * rounding of a block cut (bit for bit with reference_order = 1).  Worth it from page-locked buffers (ra_host_alloc): pageable copies are
 * staged by the runtime and do not overlap.  With an output feedback the call is ra_process_host.
 */
int ra_process_host_pipelined(ra_context *ctx, const float *h_voice, const float *h_carrier, int64_t n_frames, float *h_out, int64_t chunk_frames);
/*
 * Page-locked host memory for the buffers handed to ra_process_host (hipHostMalloc / hipHostFree): from pageable memory the runtime stages
 * every copy through its own pinned bounce buffers (~12 GB/s measured on the test box), from these the copies run at the link's rate.  Where the
 * streams come from the host — BASELINE configs[4]: 1000 streams per GPU — the link, not the kernels, paces a block.
 * The reference has no counterpart (its job reads and writes WAV files frame by frame, rt.resynth.lib.params.cpp:291-324).
 */
void *ra_host_alloc(int64_t bytes);
void ra_host_free(void *p);

/*
 * Streaming variant for back-to-back blocks: like ra_process_device, but the caller's stream is NOT made to wait for the block.
 * The next block's analysis / tracking / state propagation then overlap this block's rendering (per-block scratch exists twice; every
 * stage still runs its blocks in order, so results are identical to the synchronous calls).  Contract: inputs stay untouched and
 * `d_out` is not read until ra_join (which makes the caller's stream wait for all blocks issued so far, without blocking the host)
 * or ra_synchronize.  Every other entry point joins first.  This is how an offline job (the reference's block loop,
 * source/rt.resynth.lib.cpp:1204-1234) keeps the GPU full across block boundaries.
 */
int ra_process_device_async(ra_context *ctx, const float *d_voice, const float *d_carrier, int64_t n_frames, float *d_out);
int ra_join(ra_context *ctx);

/*
 * Queues note events for the carrier synth (HOST pointer).  Per stream, samples must not decrease and must not lie before the frames
 * already processed; events are applied by the ra_process_* call whose block contains them.  Returns RA_OK, or RA_EINVAL at the first
 * bad event (earlier ones stay queued): unknown stream / type, a sample in the past, a note-off for a key with no pending note-on.
 * ra_reset drops what is still queued.
 */
int ra_carrier_events(ra_context *ctx, const ra_carrier_event *h_events, int64_t n_events);
/* debug / test tap: the carrier the vocoder was given during the LAST ra_process_* call, [n_frames] doubles of one stream (HOST pointer) */
int ra_get_carrier(ra_context *ctx, int32_t stream, double *h_carrier, int64_t cap);

/*
 * The reference's block-compute boundary: ComputeFunc = bool(double* buffer, int nFrames)
 * (include/simple_audio_context.h:24-26), as registered by ImplCRTP::initialize (gen.crtp.h:258-264) and
 * Vocoder::initialize (vocoder.cpp:760-815).  ra_feed pushes the n input frames of stream `stream` those computes will consume
 * (input_f, lib.cpp:1285-1291; h_carrier may be NULL: silence — or, on a context with a carrier synth, that synth's notes);
 * ra_compute then ACCUMULATES (+=) nFrames (1..16) of stream `stream` into the caller's interleaved double buffer [nFrames][2],
 * exactly like the two registered computes together (not post-processed: the post-processors run after every compute).
 * One context serves the callables of all its streams: per tile every stream is fed before the first compute runs, as in
 * SimpleAudioOutContext::step (:80-102); a stream that lags makes ra_compute fail with RA_EINVAL.
 * A context without vocoder and without output feedback renders AHEAD: events apply where their analysis frame fires (:74-76), so the
 * compute that reaches a fire sample renders the whole hop up to the next one for every stream in one launch sequence, and the
 * following computes are served from a host ring (the hop's input is rolled into the window history once it has been fed).  With a
 * vocoder or a feedback every tile is a block of its own.  post = limit together with an output feedback is refused here (the loop feeds
 * back the limited signal, which only the caller of the computes has): use ra_process_*.
 * HOST pointers.  Returns 1 where the reference functor returns true, or a negative RA_E* code.
 */
int ra_feed(ra_context *ctx, int32_t stream, const float *h_voice, const float *h_carrier, int32_t n_frames);
int ra_compute(ra_context *ctx, int32_t stream, double *h_buffer, int32_t n_frames);

/*
 * Per-kernel device time, measured with HIP events on the context's launch stream — the GPU-side counterpart of the
 * reference's per-stage ThreadCPUTimer gauges (periodicfft.cpp:140-151, lib.cpp:1586-1617, getDurationFft/Extract/Step).
 * ms6 = accumulated milliseconds of {analysis (fft+extract+stateless step), tracking (step), vocoder, synthesis state
 * propagation, synthesis render (compute), history copy} over `calls` ra_process_* calls since profiling was enabled.
 * (Since r04 the tracking pre-pass ra_track_prep_kernel — the per-item half of synthesize_sounds' payloads and of
 * order_pitches_by_perceived_loudness, lib.cpp:316-376,1716-1727 — is timed as a stage of its own, see ra_get_stage_timing.)
 */
int ra_set_profiling(ra_context *ctx, int32_t enable);
/* Measurement aid: with on != 0 the stages of a block (same chunks, same launches) run one after the other on the caller's stream
 * instead of overlapping on the context's internal streams, so that ra_get_kernel_ms reports what each kernel costs with the chip to
 * itself.  Results are identical either way. */
int ra_set_pipeline_serial(ra_context *ctx, int32_t on);
int ra_get_kernel_ms(ra_context *ctx, double *ms6, int64_t *calls);
/* number of kernel launches behind each entry of ms6 (a block is cut into pipeline chunks: several launches per stage and call) */
int ra_get_kernel_launches(ra_context *ctx, int64_t *launches6);
/* every timed stage: ms[i] / launches[i] for i < min(cap, returned count), in the order {analysis, tracking, vocoder, propagation, render,
 * history copy, tracking pre-pass}; returns the number of stages (7) or a negative RA_E* code */
int ra_get_stage_timing(ra_context *ctx, int32_t cap, double *ms, int64_t *launches, int64_t *calls);

/*
 * Diagnostic: the clock the chip HELD inside the two throughput kernels.  With on != 0 lane 0 of every analysis / render workgroup stamps
 * s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end; ra_get_clock_stamps returns, for {analysis, render}, the
 * duration-weighted mean clock in GHz over the workgroups stamped since the switch-on (or the last call of ra_set_clock_stamps, which
 * also clears the sums) and their number.  Only the diagnostic build has the stamps (make -C cpp_audio_amd/csrc clk -> libresynth_amd_clk.so:
 * they cost both kernels registers); in the product build ra_set_clock_stamps(on) returns RA_EINVAL and no stamp exists.  Nothing in the
 * reference corresponds: it exists so that bench.py can say against which clock its VALU issue fractions are taken.
 */
int ra_set_clock_stamps(ra_context *ctx, int32_t on);
int ra_get_clock_stamps(ra_context *ctx, double *ghz2, int64_t *workgroups2);

/* counters and device-side error flags (synchronises) */
int ra_get_stats(ra_context *ctx, ra_stats *stats);
/* events of the LAST ra_process_* call for one stream, in the order the reference would have run them; returns count or <0 */
int64_t ra_get_events(ra_context *ctx, int32_t stream, ra_event *h_events, int64_t cap);
/* autotuned pitch lists (the output of the stateless half of RtResynth::step, lib.cpp:1676-1708) of the LAST
 * ra_process_* call for one stream: fire_sample[f], offsets[f..f+1] into pitch/volume; returns frame count or <0 */
int64_t ra_get_frame_pitches(ra_context *ctx, int32_t stream, int64_t *h_fire_sample, int64_t *h_offsets, int64_t frame_cap,
                             double *h_pitch, double *h_volume, int64_t item_cap);

/*
 * Replaces resynth<T>(notes, stride, sampling_rate, path) of include/resynth.hpp:4-100 minus the WAV writer:
 * renders the mono double signal the reference would have written.  h_out may be NULL to query the length.
 * Returns the number of frames (>= 0) or a negative RA_E* code.
 */
int64_t ra_resynth_notes(const ra_note *notes, int32_t n_notes, int32_t stride, int32_t sample_rate, int32_t device, double *h_out,
                         int64_t cap);

/*
 * Replaces deduceNotesSlow(begin, end, sample_rate, windowed_signal_stride = 1, half_window, window_center_stride, zero_padding_factor = 1,
 * c_diffHalfToneApartLog2Freqs) as main.resynth calls it (source/main.resynth.cpp:39-46): the notes of a mono signal, for ra_resynth_notes.
 * deduceNotesSlow is cpp.algorithms code that is absent here (PARITY UNPINNED); what is built:
 *   - frame h = the 2 * half_window_size samples from h * window_center_stride on, windowed (window_kind, the half window mirrored),
 *     zero padded to the next power of two, |X|^2, local maxima refined by the parabola through three dB values — the very analysis of
 *     the rt.resynth path (findFrequenciesSqMag + extractLocalMaxFreqsMags), run by the same kernel;
 *   - peaks are linked into notes frame by frame with the rule of track_pitches (rt.resynth.lib.algo.cpp:256-305): peaks and open notes in
 *     ascending pitch, a peak continues the first not-yet-continued note within half a tone... (threshold: one semitone of pitch =
 *     c_diffHalfToneApartLog2Freqs in log2 frequency), else it opens a note; a note no peak continued is closed;
 *   - a note's frequency is the volume-weighted mean pitch of its peaks (PonderateByVolume, lib.cpp:400-403) converted back to Hz, its
 *     amplitude the largest of their dB values; start / end frame = the first / last frame it was seen in (hop units, resynth.hpp:58-61).
 * Notes come out ordered by closing frame, then pitch.  h_signal: HOST pointer, mono float32.  Returns the number of notes (all of them
 * counted, the first `cap` stored; h_notes may be NULL to query) or a negative RA_E* code.
 */
int64_t ra_deduce_notes(const float *h_signal, int64_t n_frames, int32_t sample_rate, int32_t half_window_size, int32_t window_kind,
                        int32_t window_center_stride, int32_t device, ra_note *h_notes, int64_t cap);
/* The same for n_signals signals of n_frames each (h_signals [n_signals][n_frames], HOST pointer) through one context: one analysis launch
 * sequence per block over all of them, every signal's linking on a host thread.  The notes of signal s are
 * h_notes[h_offsets[s] .. h_offsets[s + 1]) (h_offsets [n_signals + 1], may be NULL); returns the total number of notes (the first `cap`
 * stored) or a negative RA_E* code.  A batch of 256 signals of 10 s is one call, not 256 contexts (main.resynth.cpp:39-46 over a folder). */
int64_t ra_deduce_notes_batch(const float *h_signals, int32_t n_signals, int64_t n_frames, int32_t sample_rate, int32_t half_window_size, int32_t window_kind,
                              int32_t window_center_stride, int32_t device, ra_note *h_notes, int64_t cap, int64_t *h_offsets);

/* audioelement::AHDSR (include/audioelement.h:314-335) in frames; every interpolation is itp::LINEAR on this path */
typedef struct ra_ahdsr {
  int32_t attack, hold, decay, release;
  float sustain;
} ra_ahdsr;

/*
 * One note of one driver-(A) element — Enveloped<VolumeAdjusted<SineOscillatorAlgo<double>>, AHDSREnvelope> (resynth.hpp:13-15) —
 * i.e. the samples its step()/imag() pairs yield after forgetPastSignals(); setAngleIncrements(angle_increment);
 * setVolumeTarget(volume); setAHDSR(*env, sample_rate); onKeyPressed(0); onKeyReleased(release_delay)  (resynth.hpp:83-96),
 * until getEnvelope().isEnvelopeRTActive() turns false.  phase_re_im[2]: the oscillator phasor carried by the (reused)
 * element, in: before the note, out: after it (a new element starts at {1, 0}, audioelement.h:2438).
 * h_out may be NULL to query the length.  Returns the number of frames or a negative RA_E* code.
 */
int64_t ra_render_note(double angle_increment, double volume, const ra_ahdsr *env, int32_t release_delay, int32_t next_min_change_before,
                       int32_t sample_rate, int32_t device, double *phase_re_im, double *h_out, int64_t cap);

/* building blocks exposed for tests and for finer-grained integration (HOST pointers, one call = one launch) */
/* findFrequenciesSqMag (periodicfft.cpp:304-311) on n_windows windows of `window_size` float samples -> [n_windows][L/2] */
int ra_sqmag(const float *h_windows, int32_t n_windows, int32_t window_size, int32_t window_kind, int32_t device, double *h_sqmag,
             int32_t *fft_length);

/*
 * ---- gen.voice: the SWEEP, BIRDS, ROBOTS and WIND voices ----------------------------------------------------------
 * voice::Impl_<..., MODE, ...> (include/gen.voice.h:1087-1190): the same ImplCRTP compute / onEvent boundary as the sine bank, the
 * element being VolumeAdjusted<SoundEngine<MODE, 2, ...>> (include/soundengine.h:240-1127).  A context of its own: the voice is driven
 * by note events, not by the analysis.  Parameters are the program's, de-normalised (gen.voice.h:55-208,474-529,533-621).
 * At a note-on the engine's Markov chain (create_birds :565-595, create_robot :597-695, create_sweep :697-719, create_wind :721-735;
 * do_initialize :963-1003) writes up to 30 ramp specs; orchestrate_algos / playNextSpec (:744-803) then play them one after the other on
 * the engine's three cross-fading ramps.  Here the chain and the orchestration — integer and PRNG work — run on the host at the note-on
 * (ra_voice_expand_note shows the result), and every ramp becomes a job of the device kernels (csrc/ra_voice.hip, csrc/ra_wind.cuh).
 * What the path takes from the absent cpp.algorithms is decided in DESIGN.md §5 (PARITY UNPINNED): MarkovChain restated from its call
 * sites; mersenne<> draws = the declared counter hash (salt 5: the k-th draw after the per-note reseed; salt 3: SWEEP's random pan;
 * salt 9: the pink-noise generator's white draws; salt 11: the start positions of a WIND ramp's iterators);
 * itp::interpolate = Penner's easing equations (enum ra_itp: this build's numbering).
 * WIND (soundengine.h:194-230,721-735): a ramp mixes three filtered pink-noise voices and the sine; its generators (include/noise.h,
 * source/sound.cpp) and controls are the reference's, its order-n Filter is n one-poles in series and its band-pass' compensation is
 * expt<VariableOrder>(.) = 1 (decided).  The Mix's first voice — a low-pass whose frequency nothing ever sets — is
 * undefined in the reference: pink_noise_lp_gain must be 0 (as in every program of the reference), else RA_EINVAL.
 */
enum { RA_VOICE_BIRDS = 0, RA_VOICE_ROBOTS = 1, RA_VOICE_SWEEP = 2, RA_VOICE_WIND = 3 }; /* SoundEngineMode, soundengine.h:31-40 */
enum { RA_XFADE_FREQ_NO = 0, RA_XFADE_FREQ_NONTRIVIAL = 1, RA_XFADE_FREQ_ALL = 2 };     /* FreqXfade, soundengine.h:3-11 */
enum ra_itp {
  RA_ITP_LINEAR = 0, RA_ITP_PROPORTIONAL_VALUE_DERIVATIVE = 1,
  RA_ITP_EASE_IN_QUAD, RA_ITP_EASE_OUT_QUAD, RA_ITP_EASE_INOUT_QUAD, RA_ITP_EASE_IN_CUBIC, RA_ITP_EASE_OUT_CUBIC, RA_ITP_EASE_INOUT_CUBIC,
  RA_ITP_EASE_IN_QUART, RA_ITP_EASE_OUT_QUART, RA_ITP_EASE_INOUT_QUART, RA_ITP_EASE_IN_QUINT, RA_ITP_EASE_OUT_QUINT, RA_ITP_EASE_INOUT_QUINT,
  RA_ITP_EASE_IN_SINE, RA_ITP_EASE_OUT_SINE, RA_ITP_EASE_INOUT_SINE, RA_ITP_EASE_IN_EXPO, RA_ITP_EASE_OUT_EXPO, RA_ITP_EASE_INOUT_EXPO,
  RA_ITP_EASE_IN_CIRC, RA_ITP_EASE_OUT_CIRC, RA_ITP_EASE_INOUT_CIRC, RA_ITP_END
};
typedef struct ra_voice_config {
  int32_t sample_rate;
  int32_t n_channels;            /* 2 * nVoices (gen.crtp.h:225), <= 256 */
  float length_ms;               /* LENGTH (10 .. 500) */
  float length_exponent;         /* LENGTH_EXPONENT (0 .. 7): a ramp lasts length_ms * 2^exponent */
  float xfade_length_seconds;    /* XFADE_LENGTH (0.0023 .. 0.0454): attack and release of a ramp's envelope */
  float low_freq, high_freq;     /* LOW_FREQ, HIGH_FREQ: the sweep's ends, Hz (SWEEP) */
  float sine_gain;               /* SINE_GAIN */
  float loudness_level;          /* LOUDNESS_LEVEL, phons (20 .. 100) */
  float loudness_compensation_amount; /* LOUDNESS_COMPENSATION_AMOUNT (log_ratio, 0 .. 1) */
  int32_t loudness_ref_freq_index;    /* LOUDNESS_REF_FREQ_INDEX (0 .. 15) */
  float pan;                     /* PAN (-1 .. 1) */
  int32_t random_pan;            /* RANDOM_PAN as a boolean */
  uint64_t seed;                 /* key of the declared draws; stream s uses seed + s */
  /* ---- BIRDS / ROBOTS ---- */
  int32_t mode;                  /* RA_VOICE_* */
  int32_t interpolation;         /* INTERPOLATION: the ramps' frequency law (ra_itp) */
  int32_t markov_start_node;     /* MARKOV_START_NODE (0 .. 2) */
  int32_t markov_pre_tries;      /* MARKOV_PRE_TRIES (0 .. 20): normalised steps without lambdas */
  int32_t markov_min_path_length;/* MARKOV_MIN_PATH_LENGTH (0 .. 20): normalised steps with lambdas */
  int32_t markov_additional_tries;/* MARKOV_ADDITIONAL_TRIES (0 .. 20): raw steps with lambdas */
  float articulative_pause_seconds; /* MARKOV_ARTICULATIVE_PAUSE_LENGTH (0 .. 0.5) */
  float freq_scatter;            /* FREQ_SCATTER (0 .. 1) */
  float length_exponent_scatter; /* LENGTH_EXPONENT_SCATTER (0 .. 1): the exponent is drawn in exponent * (1 -+ scatter) */
  float phase_ratio1, phase_ratio2; /* PHASE_RATIO1 / 2 (clamped to 0 .. 1; ratio2 is carried but unused, as in the reference) */
  float d1, d2;                  /* D1, D2 (ROBOTS): halftones one of the two tones is raised by (0 .. 47) */
  float harmonic_attenuation;    /* HARMONIC_ATTENUATION (ROBOTS; clamped to 0 .. 0.99) */
  int32_t xfade_freq;            /* MARKOV_XFADE_FREQ (BIRDS; RA_XFADE_FREQ_*) */
  float freq_transition_seconds; /* FREQ_TRANSITION_LENGTH (BIRDS; 0 .. 0.5) */
  int32_t freq_interpolation;    /* FREQ_TRANSITION_INTERPOLATION (BIRDS; ra_itp) */
  int32_t seed_param;            /* SEED (0 .. 257): 0 = the next of 1 .. 257 per note-on of a stream, else this seed for every note */
  /* ---- WIND (gen.voice.h:142-175) ---- */
  float pink_noise_lp_gain;      /* PINK_NOISE_LP_GAIN: must be 0 (the voice is undefined in the reference, see above) */
  float pink_noise_bp_gain;      /* PINK_NOISE_BP_GAIN (0 .. 1): Mix gain of the band-passed pink noise */
  float pink_noise_br_gain;      /* PINK_NOISE_BR_GAIN (0 .. 1): of the band-rejected pink noise */
  float bp_octave_width_min, bp_octave_width_max;                 /* PINK_NOISE_BP_OCTAVE_WIDTH_MIN / MAX (0 .. 5): the bands' half width, octaves */
  float center_octave_min_long_term, center_octave_max_long_term; /* CENTER_OCTAVE_MIN / MAX_LONG_TERM (0 .. 10): centres range over 10 Hz * 2^value */
  float center_short_term_ratio; /* CENTER_SHORT_TERM_RATIO (0 .. 1): octaves the short-term noise moves the sine by */
  float seconds_slow_iter_short_term, seconds_slow_iter_long_term; /* SECONDS_SLOW_ITER_*: the parameter x — a slow iterator takes sample_rate * 2.268^x steps per noise sample */
  int32_t order_filters;         /* ORDER_FILTERS (1 .. 258) */
} ra_voice_config;
typedef struct ra_voice_event {
  int64_t sample;    /* frame of Impl_::doProcessing at which the event reaches onEvent (gen.voice.h:1159-1165) */
  int32_t stream;
  int32_t type;      /* 0 NoteOn, 1 NoteOff */
  int64_t noteid;    /* unique per note (events.h:55-107); a note-off names its note-on's id */
  float frequency;   /* > 0 (SoundEngine::do_initialize refuses base_freq <= 0, soundengine.h:972-975); SWEEP ignores its value */
  float velocity;
} ra_voice_event;
typedef struct ra_voice ra_voice;
int ra_voice_default_config(ra_voice_config *cfg); /* the "Sweep 1" program (gen.voice.h:752-754) with its sweep ends at 100 Hz / 20 kHz */
/* the reference's own programs (ImplBase::getPrograms, gen.voice.h:700-810) with make_common's defaults (:474-531): mode RA_VOICE_BIRDS
 * index 0 .. 8, RA_VOICE_ROBOTS 0 .. 1, RA_VOICE_SWEEP 0 .. 1, RA_VOICE_WIND 0 .. 14; returns RA_EINVAL beyond.  `name` (may be NULL) receives the program's name. */
int ra_voice_program(int32_t mode, int32_t index, ra_voice_config *cfg, const char **name);
/* host-only (no GPU): the pink-noise sound buffer WIND's voices and controls read (getPinkNoise, source/sound.cpp:7-10; include/noise.h:82-165)
 * over the declared draws — returns its length, copies min(cap, length) samples to h_out (may be NULL), its abs mean to *abs_mean (may be
 * NULL) — and the declared start position of one of a ramp's six iterators (0 long-term, 1 short-term, 2 / 3 band-pass centre / width,
 * 4 / 5 band-reject centre / width; which: 0 initializeForRun, 1 set_sample_rate; include/sound.h:196-200) */
int64_t ra_voice_pink_noise(int32_t sample_rate, double *h_out, int64_t cap, float *abs_mean);
int32_t ra_voice_wind_start_position(uint64_t stream_seed, int64_t noteid, int32_t play, int32_t iterator, int32_t which, int64_t buffer_length);
int ra_voice_create(const ra_voice_config *cfg, int32_t n_streams, int32_t device, ra_voice **out);
void ra_voice_destroy(ra_voice *v);
/* queue events (HOST pointer; per stream in non-decreasing sample order, not before the frames already processed) */
int ra_voice_events(ra_voice *v, const ra_voice_event *h_events, int64_t n_events);
/* Impl_::doProcessing over n_frames for every stream: h_out [n_streams][n_frames][2] float32 (HOST pointer); consecutive calls continue.
 * A block that fails after it has started to consume the queued events (a staging or device allocation, a HIP call) leaves the object
 * poisoned, like ra_context: every later ra_voice_process_* returns RA_EINTERNAL and the caller recreates the object. */
int ra_voice_process_host(ra_voice *v, int64_t n_frames, float *h_out);
/* the same with the output resident in HBM: d_out [n_streams][n_frames][2] float32 (DEVICE pointer); returns when the block is done */
int ra_voice_process_device(ra_voice *v, int64_t n_frames, float *d_out);
/* hipStream_t on which every later call enqueues the work that writes d_out (NULL = the null stream) */
int ra_voice_set_hip_stream(ra_voice *v, void *hip_stream);
/* ra_voice_process_device without the wait: the block's bookkeeping runs on the host, its render is enqueued on the object's stream
 * and the call returns; d_out is complete when that stream has drained (hipStreamSynchronize / ra_voice_synchronize).  The block's
 * copies and state pre-pass run on a queue the object owns, which the render waits for: block k + 1's pre-pass beside block k's render. */
int ra_voice_process_device_async(ra_voice *v, int64_t n_frames, float *d_out);
int ra_voice_synchronize(ra_voice *v);
/* note-ons that found no channel (onDroppedNote), all streams; negative RA_E* on error */
int64_t ra_voice_dropped_note_ons(ra_voice *v);
/* device time of the two kernels of a block, accumulated since the last call with reset != 0 (HIP events on the queue each runs on):
 * ms2 = {state pre-pass, render}, launches2 likewise; synchronises */
int ra_voice_get_kernel_ms(ra_voice *v, double *ms2, int64_t *launches2, int32_t reset);

/*
 * Host-only (no device needed): what the engine makes of ONE note-on — setupAudioElement + initialize_<mode> + the orchestration's
 * schedule.  `seed_value` is the value the per-note reseed takes (SEED, or the stream's running counter), `stream_seed` the key
 * (cfg->seed + stream).  Outputs (each may be NULL; capacities in entries):
 *   spec_d [.][4] = {from, to, duration in samples, start sample} (angle increments, LogRamp::setup), spec_i [.][3] = {ra_itp, silence
 *   follows, 0}, spec_vol [.]: the ramp specs in build order;
 *   played [.][4] = {element step (1-based) at which playNextSpec starts the spec, ramp 0 .. 2, spec, steps until the ramp's key release};
 *   info [4] = {specs, plays, element steps until every ramp is finished and no spec is left, float bits of the left stereo gain}.
 * Returns RA_OK, RA_EINVAL (bad config / frequency) or RA_EOVERFLOW (a capacity is too small; info is still filled).
 */
int ra_voice_expand_note(const ra_voice_config *cfg, float frequency, uint64_t stream_seed, int32_t seed_value, int64_t noteid, double *spec_d, int32_t *spec_i,
                         float *spec_vol, int64_t spec_cap, int64_t *played, int64_t played_cap, int64_t *info);

#ifdef __cplusplus
}
#endif
#endif
