/*
 * TEST INFRASTRUCTURE — CPU oracle for the rt.resynth hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
 * under oracle/.  The product (cpp_audio_amd/, include/) never includes or links this.
 *
 * Parameter block of one RtResynth instance, restated from the atomics of
 * /root/reference/source/rt.resynth.lib.cpp:943-999 (defaults identical) plus the two
 * sizes the reference derives from seconds (:484-487, :1576-1578).
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_AUTOTUNE_SCALE = 0, ORA_AUTOTUNE_CHORD = 1, ORA_AUTOTUNE_INTERVALS = 2 };     /* rt.resynth.lib.autotune.cpp:3-8  */
enum { ORA_SCALE_MAJOR = 0, ORA_SCALE_MINOR_NATURAL = 1, ORA_SCALE_MINOR_HARMONIC = 2 }; /* rt.resynth.lib.autotune.cpp:9-14 */
enum { ORA_CHORD_SINGLE = 0, ORA_CHORD_HARMONICS = 1, ORA_CHORD_OCTAVES = 2 };           /* rt.resynth.lib.autotune.cpp:16-20 */
enum { ORA_WINDOW_GAUSSIAN4 = 0, ORA_WINDOW_HANN = 1, ORA_WINDOW_RECT = 2 };
enum { ORA_PHASE_HASH = 0, ORA_PHASE_ZERO = 1 };

typedef struct ora_config {
  int32_t sample_rate;
  int32_t window_size;          /* N, even (getEvenWindowSizeFrames, lib.cpp:484-487) */
  int32_t stride;               /* H (lib.cpp:1577-1578) */
  int32_t n_channels;           /* P = 2 * nVoices (gen.crtp.h:225) */
  int32_t window_kind;          /* ORA_WINDOW_* ; reference = Gaussian 4 sigma */

  float min_volume;             /* 0.0001 */
  float nearby_distance_tones;  /* 0.4 */
  float max_track_pitches;      /* 1.0 */
  float autotune_tolerance_pitches; /* 100 */
  float pitch_shift_pre_autotune, pitch_shift_post_autotune;
  float pitch_harmonize_pre_autotune, pitch_harmonize_post_autotune;
  float stereo_spread;          /* reference default 1; parity runs use 0 */

  float env_attack_seconds, env_hold_seconds, env_decay_seconds, env_release_seconds, env_sustain_level;

  int32_t use_autotune;
  int32_t autotune_type;
  int32_t autotune_max_pitch;   /* 150 */
  int32_t autotune_factor;      /* 2 */
  int32_t autotune_scale_mode;
  int32_t autotune_root_note;   /* 0 = Do ... 11 = Si */
  int32_t autotune_root_transpose;
  int32_t autotune_chord_frequencies;
  uint64_t autotune_bit_chord;  /* 0b10010001 */

  /* vocoder */
  float vocoder_modulator_window_size_seconds; /* 0.10 */
  float vocoder_stride_seconds;                /* 0.005 */
  int32_t vocoder_count_bands;                 /* 5 */
  float vocoder_min_freq, vocoder_max_freq;    /* 100, 20000 */
  float voice_volume, carrier_volume, vocoder_volume, analysis_volume;

  float analysis_input_gain;    /* 1 */

  /* declared policy replacing the reference's unpinned mersenne<> engines */
  int32_t phase_policy;
  uint64_t seed;

  /* feedback of the (mono) output into the analysis input, lib.cpp:1626-1652 */
  float analysis_output_feedback_gain; /* 0 */
  float output_delay_seconds;          /* 1 */

  /* SURVEY.md §8 f2 */
  int32_t vocoder_kind;                    /* 0 = the live FFT vocoder, 1 = the band-pass bank it replaced (vocoder.cpp:46-79,310-394,667-732) */
  float vocoder_env_follower_cutoff_ratio; /* 1/20, lib.cpp:985 */
  int32_t carrier_n_channels;              /* channels of the vocoder carrier synth (reference 2 * 127, lib.cpp:212-221); 0 = none */
  float vocoder_carrier_noise_volume, vocoder_carrier_saw_volume, vocoder_carrier_triangle_volume, vocoder_carrier_square_volume,
      vocoder_carrier_sine_volume, vocoder_carrier_pulse_volume, vocoder_carrier_pulse_width; /* lib.cpp:978-984: 0 0 0 1 0 0 0.01 */
  int32_t post; /* 0 none, 1 limit (params.cpp:152-176, lib.cpp:1245-1275) */
} ora_config;

/* one note event of the vocoder carrier synth (RtResynth::onInputMidiEvent, lib.cpp:1518-1545) */
typedef struct ora_carrier_event {
  int64_t sample;  /* output frame whose step() runs the event's one-shot */
  int32_t type;    /* 0 NoteOn, 1 NoteOff */
  int32_t key;     /* MIDI key = NoteIdsGenerator key */
  int32_t delay;   /* frames handed to onKeyPressed / onKeyReleased */
  float frequency; /* Hz; <= 0: midi_pitch_to_freq(key) */
  float velocity;
} ora_carrier_event;

#ifdef __cplusplus
}
#endif
