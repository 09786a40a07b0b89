/*
 * TEST INFRASTRUCTURE — CPU oracle, part 2: the in-repo decision pipeline of rt.resynth.
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 * Pinned by the reference's only known-answer tests for this path,
 * source/rt.resynth.lib.test.cpp:5-28 (see tests/test_oracle_kat.py).
 */
#pragma once
#include "oracle_dsp.hpp"
#include <array>
#include <cstdint>
#include <optional>

namespace ora {

// ---- include/midi.h -------------------------------------------------------------------------
constexpr int num_halftones_per_octave = 12;    // midi.h:112
constexpr float freq_A = 440.f;                 // midi.h:114
constexpr double A_pitch = 69.0;                // midi.h:115
constexpr int ref_A_octave = 4;                 // midi.h:116
constexpr double max_audible_midi_pitch = 151.; // midi.h:117

enum Note { Do, Dod, Re, Red, Mi, Fa, Fad, Sol, Sold, La, Lad, Si }; // midi.h:11-24

inline int half_tones_distance(int a, int b) { return b - a; } // midi.h:73-79

struct NoteOctave { // midi.h:120-162
  int note;
  int octave;
  NoteOctave add_halftones(long n) const {
    NoteOctave res(*this);
    while (n >= num_halftones_per_octave) { n -= num_halftones_per_octave; ++res.octave; }
    while (n < 0) { n += num_halftones_per_octave; --res.octave; }
    for (; n > 0; --n) {
      if (res.note == Si) { res.note = Do; ++res.octave; }
      else res.note = res.note + 1;
    }
    return res;
  }
  int dist_halftones(NoteOctave const &other) const {
    return half_tones_distance(note, other.note) + num_halftones_per_octave * (other.octave - octave);
  }
  bool operator==(NoteOctave const &o) const { return note == o.note && octave == o.octave; }
};

struct Midi { // midi.h:234-268 with tuning_stretch = 1
  double half_tone_ratio = std::pow(2.0, 1.0 / num_halftones_per_octave);
  double tuning_stretch = 1.0;
  double get_pitch(NoteOctave const &n) const { // midi.h:248-251
    return A_pitch + tuning_stretch * NoteOctave{La, ref_A_octave}.dist_halftones(n);
  }
  // midi.h:223-232: nullopt when freq <= 0
  std::optional<double> frequency_to_midi_pitch(double freq) const {
    if (freq <= 0) return {};
    return A_pitch + (num_halftones_per_octave / tuning_stretch) * std::log2(freq / freq_A);
  }
  // the same for analysed peaks: log2 as the explicit sequence of oracle_math.hpp (its last bits decide events, see there)
  std::optional<double> frequency_to_midi_pitch_analysis(double freq) const {
    if (freq <= 0) return {};
    if (libm_mode()) return A_pitch + (num_halftones_per_octave / tuning_stretch) * std::log2(freq / freq_A);
    return A_pitch + (num_halftones_per_octave / tuning_stretch) * det_log2(freq / freq_A);
  }
  // the same for the frequency of an event (lib.cpp:320): the power as the explicit sequence of oracle_math.hpp, so that the float it is
  // rounded to is reproducible (libm powers differ by an ulp now and then: a different float about once in 1e8 events)
  double midi_pitch_to_freq_event(double pitch) const {
    if (libm_mode()) return midi_pitch_to_freq(pitch);
    return freq_A * det_pow(half_tone_ratio, pitch - A_pitch);
  }
  double midi_pitch_to_freq(double pitch) const { // midi.h:258-264
    return freq_A * std::pow(half_tone_ratio, pitch - A_pitch);
  }
  double midi_pitch_offset_nth_harmonic(int n) const { // midi.h:270-272 (log2 of a float)
    return (num_halftones_per_octave / tuning_stretch) * std::log2(static_cast<float>(n));
  }
};

// ---- include/loudness.h ---------------------------------------------------------------------
namespace loudness {
constexpr int n_freq = 29;
constexpr float freqs[n_freq] = {20.f, 25.f, 31.5f, 40.f, 50.f, 63.f, 80.f, 100.f, 125.f, 160.f, 200.f, 250.f, 315.f,
                                 400.f, 500.f, 630.f, 800.f, 1000.f, 1250.f, 1600.f, 2000.f, 2500.f, 3150.f, 4000.f,
                                 5000.f, 6300.f, 8000.f, 10000.f, 12500.f}; // loudness.h:9-39
constexpr float alpha_f_[n_freq] = {.532f, .506f, .480f, .455f, .432f, .409f, .387f, .367f, .349f, .330f,
                                    .315f, .301f, .288f, .276f, .267f, .259f, .253f, .250f, .246f, .244f,
                                    .243f, .243f, .243f, .242f, .242f, .245f, .254f, .271f, .301f}; // :58-88
constexpr float Lu_[n_freq] = {-31.6f, -27.2f, -23.0f, -19.1f, -15.9f, -13.0f, -10.3f, -8.1f, -6.2f, -4.5f,
                               -3.1f,  -2.0f,  -1.1f,  -0.4f,  0.f,    .3f,    .5f,    0.f,   -2.7f, -4.1f,
                               -1.0f,  1.7f,   2.5f,   1.2f,   -2.1f,  -7.1f,  -11.2f, -10.7f, -3.1f}; // :90-120
constexpr float Tf_[n_freq] = {78.5f, 68.7f, 59.5f, 51.1f, 44.0f, 37.5f, 31.5f, 26.5f, 22.1f, 17.9f,
                               14.4f, 11.4f, 8.6f,  6.2f,  4.4f,  3.0f,  2.2f,  2.4f,  3.5f,  1.7f,
                               -1.3f, -4.2f, -6.0f, -5.4f, -1.5f, 6.0f,  12.6f, 13.9f, 12.3f}; // :122-152

// loudness.h:188-197 (reference evaluates with sprout:: constexpr float math; std:: here — the
// ulp-level difference is unpinned and only feeds a sort key).
inline float compute_equal_loudness_volume(int i, float LN) {
  float alpha_f = alpha_f_[i], Lu = Lu_[i], Tf = Tf_[i];
  float Af = 4.47e-3f * (std::pow(10.f, .025f * LN) - 1.14f) +
             std::pow(.4f * std::pow(10.f, ((Tf + Lu) * .1f) - 9.f), alpha_f);
  return 94.f - Lu + (10.f / alpha_f) * std::log(Af) / std::log(10.f);
}

struct Tables {
  std::array<double, n_freq> pitches;              // loudness.h:43-54
  std::array<std::array<float, n_freq>, 9> elvs;   // loudness.h:199-216
  Tables() {
    Midi midi;
    for (int i = 0; i < n_freq; ++i) pitches[i] = *midi.frequency_to_midi_pitch(freqs[i]);
    for (int l = 0; l < 9; ++l)
      for (int i = 0; i < n_freq; ++i) elvs[l][i] = compute_equal_loudness_volume(i, (l + 2) * 10.f);
  }
};
inline Tables const &tables() { static Tables t; return t; }

inline int phons_to_index(float level) { // loudness.h:220-229
  int i = static_cast<int>(level * .1f) - 2;
  i = std::max(0, i);
  i = std::min(8, i);
  return i;
}

// loudness.h:159-186
inline std::pair<int, float> closest(std::array<double, n_freq> const &arr, double value) {
  int min_ = 0, max_ = n_freq - 1;
  do {
    int center = (min_ + max_) / 2;
    if (value < arr[center]) max_ = center; else min_ = center;
  } while (max_ - min_ >= 2);
  if (value <= arr[min_]) return {min_, 1.f};
  if (value >= arr[max_]) return {max_, 1.f};
  return {max_, static_cast<float>((value - arr[min_]) / (arr[max_] - arr[min_]))};
}

// loudness.h:231-244
inline float equal_loudness_volume_db(double pitch, int level) {
  auto const [i, ratio] = closest(tables().pitches, pitch);
  auto const &elv = tables().elvs[level];
  if (ratio == 1.f) return elv[i];
  return ratio * elv[i] + (1.f - ratio) * elv[i - 1];
}
} // namespace loudness

// ---- source/rt.resynth.lib.algo.cpp ---------------------------------------------------------
constexpr float pitch_epsilon = 0.0001f; // algo.cpp:3

struct PitchVolume { double midipitch; double volume; }; // algo.cpp:5-8

// algo.cpp:10-29
inline void frequencies_to_pitches(Midi const &midi, std::vector<FreqMag> const &fs, std::vector<PitchVolume> &res) {
  res.clear();
  for (auto const &f : fs)
    if (auto pitch = midi.frequency_to_midi_pitch_analysis(f.freq)) res.push_back({*pitch, db_to_mag_analysis(f.mag_db)});
}

// algo.cpp:42-104 (only the two reduction methods RtResynth uses, lib.cpp:400-403)
struct PitchInterval {
  double min_pitch, max_pitch, maxVolumePitch;
  double maxVolume = 0, sumProductsPitchVolume = 0, sumVolumes = 0;
  explicit PitchInterval(PitchVolume const &pv) : min_pitch(pv.midipitch), max_pitch(pv.midipitch), maxVolumePitch(pv.midipitch) { aggregate(pv); }
  void extend(PitchVolume const &pv) {
    min_pitch = std::min(min_pitch, pv.midipitch);
    max_pitch = std::max(max_pitch, pv.midipitch);
    aggregate(pv);
  }
  double ponderatedPitch() const { return sumProductsPitchVolume / sumVolumes; }
  void aggregate(PitchVolume const &pv) {
    sumVolumes += pv.volume;
    sumProductsPitchVolume += pv.midipitch * pv.volume;
    if (maxVolume < pv.volume) { maxVolume = pv.volume; maxVolumePitch = pv.midipitch; }
  }
};

inline double diameter(double a, double b, double c) { return std::max({a, b, c}) - std::min({a, b, c}); } // algo.cpp:107-110

// algo.cpp:124-161
inline void aggregate_pitches(double nearby_distance_tones, std::vector<PitchVolume> const &pvs, std::vector<PitchInterval> &out) {
  out.clear();
  std::optional<PitchInterval> cur;
  for (auto const &pv : pvs) {
    if (cur && diameter(cur->min_pitch, cur->max_pitch, pv.midipitch) > nearby_distance_tones) {
      out.push_back(*cur);
      cur.reset();
    }
    if (!cur) cur = PitchInterval{pv}; else cur->extend(pv);
  }
  if (cur) out.push_back(*cur);
}

// algo.cpp:166-184 with PonderateByVolume / SumVolumes
inline void reduce_pitches(double min_volume, std::vector<PitchInterval> const &in, std::vector<PitchVolume> &out) {
  out.clear();
  for (auto const &i : in) {
    double vol = i.sumVolumes;
    if (vol < min_volume) continue;
    out.push_back({i.ponderatedPitch(), vol});
  }
}

// canonical std::lower_bound on the pitch key (used by find_closest_pitch, autotune.cpp:196-201)
template <typename T, typename F>
int lower_bound_pitch(std::vector<T> const &v, double pitch, F get) {
  int first = 0, len = static_cast<int>(v.size());
  while (len > 0) {
    int half = len >> 1, mid = first + half;
    if (get(v[mid]) < pitch) { first = mid + 1; len = len - half - 1; } else len = half;
  }
  return first;
}

// autotune.cpp:189-218 — returns index or -1; ties go to the LOWER neighbour, distances in float
template <typename T, typename F>
int find_closest_pitch(double pitch, std::vector<T> const &v, F get) {
  int n = static_cast<int>(v.size());
  if (n == 0) return -1;
  int lb = lower_bound_pitch(v, pitch, get);
  if (lb == n) return lb - 1;
  if (lb == 0) return 0;
  float dist_lb = std::abs(static_cast<float>(pitch - get(v[lb])));
  float dist_prev = std::abs(static_cast<float>(pitch - get(v[lb - 1])));
  return dist_lb < dist_prev ? lb : lb - 1;
}

// algo.cpp:307-316
inline void shift_pitches(float shift, std::vector<PitchVolume> &p) {
  if (!shift) return;
  for (auto &x : p) x.midipitch += shift;
}

// algo.cpp:318-371 (merge_sort_step: stable two-way merge, first range wins ties)
inline void harmonize_pitches(float amount, std::vector<PitchVolume> &work, std::vector<PitchVolume> &pitches) {
  if (!amount) return;
  int const sz = static_cast<int>(pitches.size());
  pitches.reserve(2 * sz + 1); // the reference relies on capacity not being exceeded (lib.cpp:1105)
  for (int i = 0; i < sz; ++i) {
    PitchVolume const p = pitches[i];
    double const harmonized = p.midipitch + amount;
    int c = find_closest_pitch(harmonized, pitches, [](PitchVolume const &pv) { return pv.midipitch; });
    float const dist = std::abs(static_cast<float>(pitches[c].midipitch - harmonized));
    if (dist <= pitch_epsilon) pitches[c].volume += p.volume;
    else pitches.push_back({harmonized, p.volume});
  }
  work.clear();
  int a = 0, b = sz, ae = sz, be = static_cast<int>(pitches.size());
  while (a < ae && b < be) {
    if (pitches[b].midipitch < pitches[a].midipitch) work.push_back(pitches[b++]); else work.push_back(pitches[a++]);
  }
  while (a < ae) work.push_back(pitches[a++]);
  while (b < be) work.push_back(pitches[b++]);
  work.swap(pitches);
}

// ---- source/rt.resynth.lib.autotune.cpp + lib.cpp:1761-1873 ---------------------------------
struct MusicalScalePitches { // autotune.cpp:89-142
  std::vector<double> pitches, equidistant;
  explicit MusicalScalePitches(std::vector<double> const &all) : pitches(all) {
    for (size_t i = 0; i + 1 < all.size(); ++i) equidistant.push_back(0.5 * (all[i] + all[i + 1]));
  }
  double closest_pitch(double root_pitch, double pitch) const {
    double const half_tones_dist = pitch - root_pitch;
    double const octave_dist = half_tones_dist / num_halftones_per_octave;
    int octaves_translation = (octave_dist >= 0.) ? static_cast<int>(octave_dist) : static_cast<int>(octave_dist) - 1;
    double const translated = pitch - (octaves_translation * num_halftones_per_octave);
    double const rel = translated - root_pitch;
    int i = 0;
    for (int sz = static_cast<int>(equidistant.size()); i < sz; ++i)
      if (rel < equidistant[i]) break;
    return pitch - (rel - pitches[i]);
  }
};
inline MusicalScalePitches const &getMusicalScale(int mode) { // autotune.cpp:144-186
  static const MusicalScalePitches major{{0., 2., 4., 5., 7., 9., 11., 12.}};
  static const MusicalScalePitches minor_natural{{0., 2., 3., 5., 7., 8., 10., 12.}};
  static const MusicalScalePitches minor_harmonic{{0., 2., 3., 5., 7., 8., 11., 12.}};
  return mode == 0 ? major : mode == 1 ? minor_natural : minor_harmonic;
}

struct Autotune { // lib.cpp:1761-1873 (mkAutotuneFunction), built once since params are constant in a job
  bool use = false;
  int type = 0;
  MusicalScalePitches const *scale = nullptr;
  double root_pitch = 0;
  std::vector<double> allowed;

  void build(bool use_autotune, int autotune_type, int scale_mode, int root_note, int transpose, int chord_freqs,
             uint64_t bit_chord, int factor) {
    use = use_autotune;
    type = autotune_type;
    allowed.clear();
    if (!use) return;
    Midi midi;
    if (type == 1) { // Chord, lib.cpp:1767-1827
      int offset = half_tones_distance(Do, root_note);
      if (offset < 0) offset += num_halftones_per_octave;
      offset += transpose;
      double const C_pitch = A_pitch + half_tones_distance(La, Do) + num_halftones_per_octave;
      double const root = C_pitch + offset;
      bool single = chord_freqs == 0;
      if (chord_freqs == 0 || chord_freqs == 2) {
        int const omin = single ? 0 : -5, omax = single ? 0 : 5;
        for (int octave = omin; octave <= omax; ++octave) {
          int const add = num_halftones_per_octave * octave;
          for (int i = 0; i < 64; ++i)
            if ((bit_chord >> i) & 1) allowed.push_back(root + i + add);
        }
      } else { // Harmonics
        constexpr int n_harmo = 36;
        for (int h = 0; h < n_harmo; ++h) {
          double add = static_cast<int>(0.5 + midi.midi_pitch_offset_nth_harmonic(h + 1)); // midi.h:279-287
          for (int i = 0; i < 64; ++i)
            if ((bit_chord >> i) & 1) { allowed.push_back(root + add + i); allowed.push_back(root - add + i); }
        }
      }
      std::sort(allowed.begin(), allowed.end());
    } else if (type == 2) { // FixedSizeIntervals, lib.cpp:1828-1862
      int o = half_tones_distance(Do, root_note);
      if (o < 0) o += num_halftones_per_octave;
      double const offset = o + transpose;
      allowed.push_back(offset);
      if (factor) {
        for (double v = offset - factor; v > 0.; v -= factor) allowed.push_back(v);
        for (double v = offset + factor; v < max_audible_midi_pitch; v += factor) allowed.push_back(v);
      }
      std::sort(allowed.begin(), allowed.end());
    } else { // MusicalScale, lib.cpp:1863-1871
      scale = &getMusicalScale(scale_mode);
      root_pitch = A_pitch + transpose + half_tones_distance(La, root_note);
    }
  }
  std::optional<double> operator()(double pitch) const {
    if (!use) return pitch;
    if (type == 0) return scale->closest_pitch(root_pitch, pitch);
    int i = find_closest_pitch(pitch, allowed, [](double p) { return p; });
    if (i < 0) return {};
    return allowed[i];
  }
};

// algo.cpp:191-229
inline void autotune_pitches(double max_pitch, float pitch_tolerance, Autotune const &f, std::vector<PitchVolume> const &in,
                             std::vector<PitchVolume> &out) {
  out.clear();
  for (auto const &pv : in) {
    std::optional<double> t;
    if (pv.midipitch <= max_pitch) {
      t = f(pv.midipitch);
      if (t && *t > max_pitch) t = pv.midipitch;
    } else t = pv.midipitch;
    if (t) {
      if (std::abs(*t - pv.midipitch) <= pitch_tolerance) {
        if (!out.empty() && (std::abs(out.back().midipitch - *t) < pitch_epsilon)) out.back().volume += pv.volume;
        else out.push_back({*t, pv.volume});
      }
    }
  }
}

struct PlayedNote { // algo.cpp:233-250
  int64_t note_on_frame_id;
  int64_t noteid;
  double midi_pitch;
  float cur_freq;
  float cur_velocity;
};

// algo.cpp:256-305
inline void track_pitches(double max_track, std::vector<PitchVolume> const &new_pitches, std::vector<PlayedNote> const &played,
                          std::vector<int> &pitch_changes /* -1 = none */, std::vector<char> &continue_playing) {
  pitch_changes.assign(new_pitches.size(), -1);
  continue_playing.assign(played.size(), 0);
  int idx = -1;
  size_t it = 0, end = played.size();
  for (auto const &pv : new_pitches) {
    ++idx;
    for (; it != end; ++it) {
      if (played[it].midi_pitch < pv.midipitch - max_track) continue;
      if (played[it].midi_pitch <= pv.midipitch + max_track) {
        pitch_changes[idx] = static_cast<int>(it);
        continue_playing[it] = 1;
        ++it;
      }
      break;
    }
  }
}

// algo.cpp:375-411 with the lambda of lib.cpp:1717-1727.  std::sort leaves the order of equal
// keys implementation-defined; fixed here as index-ascending (stable) and mirrored on the device.
inline void order_pitches_by_perceived_loudness(std::vector<PitchVolume> const &pitches, std::vector<float> &keys, std::vector<int> &order) {
  int const level = loudness::phons_to_index(60.f);
  keys.clear();
  for (auto const &pv : pitches) keys.push_back(static_cast<float>(pv.volume / loudness::equal_loudness_volume_db(pv.midipitch, level)));
  order.resize(pitches.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
  std::stable_sort(order.begin(), order.end(), [&keys](int a, int b) { return keys[a] > keys[b]; });
}

// algo.cpp:415-440 (sort: equal pitches keep their order — see note above)
inline void remove_dead_notes(std::vector<char> const &continue_playing, std::vector<PlayedNote> &played) {
  std::vector<PlayedNote> kept;
  for (size_t n = 0; n < played.size(); ++n)
    if (n >= continue_playing.size() || continue_playing[n]) kept.push_back(played[n]);
  played.swap(kept);
}
inline void sort_by_current_pitch(std::vector<PlayedNote> &played) {
  std::stable_sort(played.begin(), played.end(), [](PlayedNote const &n, PlayedNote const &m) { return n.midi_pitch < m.midi_pitch; });
}

} // namespace ora
