// Test driver for cpp_audio_amd/host/preset.hpp:
//   preset_tool parse <preset.json> <sample_rate>   -> prints the ra_config that restorePreset yields, one "key value" per line
//   preset_tool roundtrip <in.json> <out.json>      -> Preset::from_file -> to_file
//   preset_tool save <sample_rate> <out.json>       -> save_as_preset(ra_default_config)
//   preset_tool jobcfg <job.json>                   -> prints the parsed job config
//   preset_tool job <job.json> <n_channels> <seed>  -> run_offline_job on device 0 (GPU), prints frames and dropped note-ons
#include <cstdio>
#include <cstring>
#include <iostream>

#include "../../cpp_audio_amd/host/preset.hpp"

using namespace resynth_amd;

static void dump(const ra_config &c) {
  std::printf("sample_rate %d\nwindow_size %d\nstride %d\nn_channels %d\n", c.sample_rate, c.window_size, c.stride, c.n_channels);
  std::printf("min_volume %.9g\nnearby_distance_tones %.9g\nmax_track_pitches %.9g\nautotune_tolerance_pitches %.9g\n", c.min_volume,
              c.nearby_distance_tones, c.max_track_pitches, c.autotune_tolerance_pitches);
  std::printf("pitch_shift_pre_autotune %.9g\npitch_shift_post_autotune %.9g\npitch_harmonize_pre_autotune %.9g\npitch_harmonize_post_autotune %.9g\n",
              c.pitch_shift_pre_autotune, c.pitch_shift_post_autotune, c.pitch_harmonize_pre_autotune, c.pitch_harmonize_post_autotune);
  std::printf("stereo_spread %.9g\nenv_attack_seconds %.9g\nenv_hold_seconds %.9g\nenv_decay_seconds %.9g\nenv_release_seconds %.9g\nenv_sustain_level %.9g\n",
              c.stereo_spread, c.env_attack_seconds, c.env_hold_seconds, c.env_decay_seconds, c.env_release_seconds, c.env_sustain_level);
  std::printf("use_autotune %d\nautotune_type %d\nautotune_max_pitch %d\nautotune_factor %d\nautotune_scale_mode %d\nautotune_root_note %d\n"
              "autotune_root_transpose %d\nautotune_chord_frequencies %d\nautotune_bit_chord %llu\n",
              c.use_autotune, c.autotune_type, c.autotune_max_pitch, c.autotune_factor, c.autotune_scale_mode, c.autotune_root_note,
              c.autotune_root_transpose, c.autotune_chord_frequencies, (unsigned long long)c.autotune_bit_chord);
  std::printf("vocoder_modulator_window_size_seconds %.9g\nvocoder_stride_seconds %.9g\nvocoder_count_bands %d\nvocoder_min_freq %.9g\nvocoder_max_freq %.9g\n",
              c.vocoder_modulator_window_size_seconds, c.vocoder_stride_seconds, c.vocoder_count_bands, c.vocoder_min_freq, c.vocoder_max_freq);
  std::printf("voice_volume %.9g\ncarrier_volume %.9g\nvocoder_volume %.9g\nanalysis_volume %.9g\nanalysis_input_gain %.9g\n", c.voice_volume,
              c.carrier_volume, c.vocoder_volume, c.analysis_volume, c.analysis_input_gain);
  std::printf("analysis_output_feedback_gain %.9g\noutput_delay_seconds %.9g\n", c.analysis_output_feedback_gain, c.output_delay_seconds);
}

int main(int argc, char **argv) {
  try {
    if (argc >= 4 && !std::strcmp(argv[1], "parse")) {
      ra_config c;
      check(ra_default_config(&c));
      c.sample_rate = std::atoi(argv[3]);
      apply_preset(Preset::from_file(argv[2]), c);
      dump(c);
      return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "roundtrip")) {
      Preset::from_file(argv[2]).to_file(argv[3]);
      return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "save")) {
      ra_config c;
      check(ra_default_config(&c));
      c.sample_rate = std::atoi(argv[2]);
      save_as_preset(c).to_file(argv[3]);
      return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "jobcfg")) {
      OfflineJobConfig j = OfflineJobConfig::from_file(argv[2]);
      std::printf("preset_file %s\ninput_voice_file %s\ninput_carrier_file %s\noutput_file %s\npost %s\n", j.preset_file.c_str(),
                  j.input_voice_file.c_str(), j.input_carrier_file.c_str(), j.output_file.c_str(), j.post == Postprocessing::None ? "none" : "limit");
      std::fputs(j.to_json_text().c_str(), stdout);
      return 0;
    }
    if (argc >= 5 && !std::strcmp(argv[1], "job")) {
      ra_config c;
      check(ra_default_config(&c));
      c.n_channels = std::atoi(argv[3]);
      c.seed = std::strtoull(argv[4], nullptr, 10);
      ra_stats st = run_offline_job_file(argv[2], c);
      std::printf("samples_rendered %lld\ndropped_note_on %lld\nerror_flags %d\n", (long long)st.samples_rendered, (long long)st.dropped_note_on, st.error_flags);
      return 0;
    }
    std::fprintf(stderr, "usage: see the header of tests/host/preset_tool.cpp\n");
    return 2;
  } catch (const std::exception &e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
}
