// Presets and offline job files of rt.resynth, over the C ABI.
//
// Mirrors the reference's JSON containers so that its preset / job files run unmodified:
//   resynth_amd::Preset            <->  Preset / Params<T>            (source/rt.resynth.lib.params.cpp:5-142): five typed maps
//                                       {"bool_params","enum_params","int32_params","uint64_params","float_params"}
//   resynth_amd::apply_preset      <->  RtResynth::restorePreset      (source/rt.resynth.lib.cpp:2043-2145): same keys; a missing key
//                                       keeps the current value and is reported on stderr like Params::read (params.cpp:18-27)
//   resynth_amd::save_as_preset    <->  RtResynth::saveAsPreset       (lib.cpp:1941-2041)
//   resynth_amd::OfflineJobConfig  <->  RtResynthOfflineJobConfig     (params.cpp:183-213): preset_file, input_voice_file,
//                                       input_carrier_file, output_file, post ("none" | "limit")
//   resynth_amd::run_offline_job   <->  the rt.resynth.job flow: read config, read preset, RtResynth(RtResynthOfflineJob&)
//
// The JSON reader below covers what nlohmann::json reads for these files: objects, strings, numbers, booleans, null, arrays.
// Keys of this path that the GPU build does not implement are carried through read/write unchanged and listed by
// unsupported_keys(): the vocoder *carrier synth* volumes (the carrier here is the job's carrier file), the envelope follower
// ratio (time-domain vocoder variant), pitch_wheel_multiplier (MIDI).
#pragma once
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "resynth_amd.hpp"
#include "wav.hpp"

namespace resynth_amd {

namespace json {

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0;
  bool is_integer = false; // no '.', 'e' or 'E' in the literal
  bool negative = false;
  uint64_t magnitude = 0;  // exact value of an integer literal (uint64 parameters do not fit a double)
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj; // insertion order kept

  const Value *find(const std::string &key) const {
    for (auto &kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
  }
  const Value &at(const std::string &key) const { // nlohmann's at(): throws when absent
    if (kind != Object) throw std::runtime_error("json: not an object while looking for '" + key + "'");
    const Value *v = find(key);
    if (!v) throw std::runtime_error("json: key '" + key + "' not found");
    return *v;
  }
};

class Parser {
public:
  explicit Parser(const std::string &text) : s_(text) {}
  Value parse() {
    Value v = value();
    ws();
    if (i_ != s_.size()) fail("trailing characters");
    return v;
  }

private:
  const std::string &s_;
  size_t i_ = 0;
  [[noreturn]] void fail(const std::string &what) const { throw std::runtime_error("json: " + what + " at offset " + std::to_string(i_)); }
  void ws() { while (i_ < s_.size() && std::isspace(static_cast<unsigned char>(s_[i_]))) ++i_; }
  bool eat(char c) { ws(); if (i_ < s_.size() && s_[i_] == c) { ++i_; return true; } return false; }
  void expect(char c) { if (!eat(c)) fail(std::string("expected '") + c + "'"); }
  Value value() {
    ws();
    if (i_ >= s_.size()) fail("unexpected end");
    char c = s_[i_];
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') { Value v; v.kind = Value::String; v.str = string(); return v; }
    if (!s_.compare(i_, 4, "true")) { i_ += 4; Value v; v.kind = Value::Bool; v.b = true; return v; }
    if (!s_.compare(i_, 5, "false")) { i_ += 5; Value v; v.kind = Value::Bool; v.b = false; return v; }
    if (!s_.compare(i_, 4, "null")) { i_ += 4; return Value(); }
    return number();
  }
  Value object() {
    Value v; v.kind = Value::Object;
    expect('{');
    if (eat('}')) return v;
    do {
      ws();
      if (i_ >= s_.size() || s_[i_] != '"') fail("expected a key");
      std::string k = string();
      expect(':');
      v.obj.emplace_back(std::move(k), value());
    } while (eat(','));
    expect('}');
    return v;
  }
  Value array() {
    Value v; v.kind = Value::Array;
    expect('[');
    if (eat(']')) return v;
    do v.arr.push_back(value()); while (eat(','));
    expect(']');
    return v;
  }
  std::string string() {
    std::string out;
    ++i_; // opening quote
    while (true) {
      if (i_ >= s_.size()) fail("unterminated string");
      char c = s_[i_++];
      if (c == '"') break;
      if (c != '\\') { out.push_back(c); continue; }
      if (i_ >= s_.size()) fail("bad escape");
      char e = s_[i_++];
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          if (i_ + 4 > s_.size()) fail("bad \\u escape");
          unsigned cp = static_cast<unsigned>(std::strtoul(s_.substr(i_, 4).c_str(), nullptr, 16));
          i_ += 4;
          if (cp < 0x80) out.push_back(static_cast<char>(cp));
          else if (cp < 0x800) { out.push_back(static_cast<char>(0xC0 | (cp >> 6))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
          else { out.push_back(static_cast<char>(0xE0 | (cp >> 12))); out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
          break;
        }
        default: fail("bad escape");
      }
    }
    return out;
  }
  Value number() {
    size_t start = i_;
    Value v; v.kind = Value::Number; v.is_integer = true;
    if (i_ < s_.size() && s_[i_] == '-') { v.negative = true; ++i_; }
    size_t digits0 = i_;
    while (i_ < s_.size() && std::isdigit(static_cast<unsigned char>(s_[i_]))) ++i_;
    if (i_ == digits0) fail("bad number");
    std::string mag = s_.substr(digits0, i_ - digits0);
    if (i_ < s_.size() && (s_[i_] == '.' || s_[i_] == 'e' || s_[i_] == 'E')) {
      v.is_integer = false;
      while (i_ < s_.size() && (std::isdigit(static_cast<unsigned char>(s_[i_])) || s_[i_] == '.' || s_[i_] == 'e' || s_[i_] == 'E' || s_[i_] == '+' || s_[i_] == '-')) ++i_;
    }
    v.num = std::strtod(s_.substr(start, i_ - start).c_str(), nullptr);
    if (v.is_integer) v.magnitude = std::strtoull(mag.c_str(), nullptr, 10);
    return v;
  }
};

inline Value parse_file(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open '" + path + "'");
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string text = ss.str();
  return Parser(text).parse();
}

inline std::string quote(const std::string &s) {
  std::string o = "\"";
  for (char c : s) {
    if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(c); }
    else if (c == '\n') o += "\\n";
    else o.push_back(c);
  }
  return o + "\"";
}

} // namespace json

// ---- Preset (params.cpp:54-142) ---------------------------------------------------------------------------------------------
struct Preset {
  std::map<std::string, bool> bool_params;
  std::map<std::string, int32_t> enum_params;
  std::map<std::string, int32_t> int32_params;
  std::map<std::string, uint64_t> uint64_params;
  std::map<std::string, float> float_params;

  void clear() { bool_params.clear(); enum_params.clear(); int32_params.clear(); uint64_params.clear(); float_params.clear(); }

  static Preset from_json(const json::Value &j) { // Preset::fromjson: all five sections are required (json::at)
    Preset p;
    for (auto &kv : j.at("bool_params").obj) {
      // nlohmann converts numbers to bool as well; the reference writes true / false
      p.bool_params[kv.first] = kv.second.kind == json::Value::Bool ? kv.second.b : kv.second.num != 0;
    }
    auto as_i32 = [](const json::Value &v, const std::string &k) -> int32_t {
      if (v.kind != json::Value::Number) throw std::runtime_error("preset: '" + k + "' is not a number");
      return static_cast<int32_t>(v.is_integer ? (v.negative ? -static_cast<int64_t>(v.magnitude) : static_cast<int64_t>(v.magnitude)) : static_cast<int64_t>(v.num));
    };
    for (auto &kv : j.at("enum_params").obj) p.enum_params[kv.first] = as_i32(kv.second, kv.first);
    for (auto &kv : j.at("int32_params").obj) p.int32_params[kv.first] = as_i32(kv.second, kv.first);
    for (auto &kv : j.at("uint64_params").obj) {
      if (kv.second.kind != json::Value::Number) throw std::runtime_error("preset: '" + kv.first + "' is not a number");
      p.uint64_params[kv.first] = kv.second.is_integer ? kv.second.magnitude : static_cast<uint64_t>(kv.second.num);
    }
    for (auto &kv : j.at("float_params").obj) {
      if (kv.second.kind != json::Value::Number) throw std::runtime_error("preset: '" + kv.first + "' is not a number");
      p.float_params[kv.first] = static_cast<float>(kv.second.num);
    }
    return p;
  }
  static Preset from_file(const std::string &path) { return from_json(json::parse_file(path)); }

  std::string to_json_text() const { // the layout of writeToJsonFile (params.cpp:238-247: std::setw(2))
    std::ostringstream os;
    os.precision(9); // a float survives 9 significant digits
    auto section = [&](const char *name, auto const &m, auto &&fmt, bool last) {
      os << "  " << json::quote(name) << ": {";
      bool first = true;
      for (auto &kv : m) {
        os << (first ? "\n" : ",\n") << "    " << json::quote(kv.first) << ": ";
        fmt(kv.second);
        first = false;
      }
      os << (first ? "}" : "\n  }") << (last ? "\n" : ",\n");
    };
    os << "{\n";
    section("bool_params", bool_params, [&](bool v) { os << (v ? "true" : "false"); }, false);
    section("enum_params", enum_params, [&](int32_t v) { os << v; }, false);
    section("float_params", float_params, [&](float v) { os << v; }, false);
    section("int32_params", int32_params, [&](int32_t v) { os << v; }, false);
    section("uint64_params", uint64_params, [&](uint64_t v) { os << v; }, true);
    os << "}\n";
    return os.str();
  }
  void to_file(const std::string &path) const {
    std::ofstream f(path);
    if (!f) throw std::runtime_error("cannot write '" + path + "'");
    f << to_json_text();
  }
};

// the keys of saveAsPreset / restorePreset this build has no use for (read and written, never applied)
inline const std::vector<std::string> &unsupported_keys() {
  static const std::vector<std::string> k = {"pitch_wheel_multiplier"}; // (MIDI pitch wheel: no MIDI device on this path)
  return k;
}

namespace detail {
template <typename M, typename T>
inline void read_param(const M &m, const std::string &name, T &val) { // Params::read, params.cpp:18-27
  auto it = m.find(name);
  if (it == m.end()) {
    std::cerr << "Preset param '" << name << "' not found" << std::endl;
    return;
  }
  val = static_cast<T>(it->second);
}
} // namespace detail

// RtResynth::restorePreset (lib.cpp:2043-2145) onto an ra_config.  The reference keeps the window and stride in seconds and
// converts when the analysis (re)starts (lib.cpp:484-487, 1577-1578): the same conversion happens here, with cfg.sample_rate.
inline void apply_preset(const Preset &p, ra_config &cfg) {
  using detail::read_param;
  if (cfg.sample_rate <= 0) throw std::invalid_argument("apply_preset: cfg.sample_rate must be set first (the job's sample rate)");
  float window_size_seconds = static_cast<float>(cfg.window_size) / cfg.sample_rate;
  float window_center_stride_seconds = static_cast<float>(cfg.stride) / cfg.sample_rate;
  const bool has_w = p.float_params.count("window_size_seconds") != 0, has_s = p.float_params.count("window_center_stride_seconds") != 0;
  read_param(p.float_params, "window_size_seconds", window_size_seconds);
  read_param(p.float_params, "window_center_stride_seconds", window_center_stride_seconds);
  if (has_w) cfg.window_size = ra_even_window_size_frames(window_size_seconds, cfg.sample_rate);
  if (has_s) cfg.stride = ra_window_center_stride_frames(window_center_stride_seconds, cfg.sample_rate);
  read_param(p.float_params, "min_volume", cfg.min_volume);
  read_param(p.float_params, "nearby_distance_tones", cfg.nearby_distance_tones);
  read_param(p.float_params, "max_track_pitches", cfg.max_track_pitches);
  read_param(p.float_params, "autotune_tolerance_pitches", cfg.autotune_tolerance_pitches);
  read_param(p.float_params, "pitch_shift_pre_autotune", cfg.pitch_shift_pre_autotune);
  read_param(p.float_params, "pitch_shift_post_autotune", cfg.pitch_shift_post_autotune);
  read_param(p.float_params, "pitch_harmonize_pre_autotune", cfg.pitch_harmonize_pre_autotune);
  read_param(p.float_params, "pitch_harmonize_post_autotune", cfg.pitch_harmonize_post_autotune);
  read_param(p.float_params, "stereo_spread", cfg.stereo_spread);
  read_param(p.float_params, "env_attack_seconds", cfg.env_attack_seconds);
  read_param(p.float_params, "env_hold_seconds", cfg.env_hold_seconds);
  read_param(p.float_params, "env_decay_seconds", cfg.env_decay_seconds);
  read_param(p.float_params, "env_release_seconds", cfg.env_release_seconds);
  read_param(p.float_params, "env_sustain_level", cfg.env_sustain_level);
  read_param(p.enum_params, "autotune_type", cfg.autotune_type);
  { bool b = cfg.use_autotune != 0; read_param(p.bool_params, "use_autotune", b); cfg.use_autotune = b ? 1 : 0; }
  read_param(p.int32_params, "autotune_max_pitch", cfg.autotune_max_pitch);
  read_param(p.int32_params, "autotune_factor", cfg.autotune_factor);
  read_param(p.enum_params, "autotune_musical_scale_mode", cfg.autotune_scale_mode);
  read_param(p.enum_params, "autotune_musical_scale_root_note", cfg.autotune_root_note);
  read_param(p.int32_params, "autotune_root_note_halftones_transpose", cfg.autotune_root_transpose);
  read_param(p.enum_params, "autotune_chord_frequencies", cfg.autotune_chord_frequencies);
  read_param(p.uint64_params, "autotune_bit_chord", cfg.autotune_bit_chord);
  read_param(p.float_params, "vocoder_carrier_noise_volume", cfg.vocoder_carrier_noise_volume);
  read_param(p.float_params, "vocoder_carrier_saw_volume", cfg.vocoder_carrier_saw_volume);
  read_param(p.float_params, "vocoder_carrier_triangle_volume", cfg.vocoder_carrier_triangle_volume);
  read_param(p.float_params, "vocoder_carrier_square_volume", cfg.vocoder_carrier_square_volume);
  read_param(p.float_params, "vocoder_carrier_sine_volume", cfg.vocoder_carrier_sine_volume);
  read_param(p.float_params, "vocoder_carrier_pulse_volume", cfg.vocoder_carrier_pulse_volume);
  read_param(p.float_params, "vocoder_carrier_pulse_width", cfg.vocoder_carrier_pulse_width);
  read_param(p.float_params, "vocoder_env_follower_cutoff_ratio", cfg.vocoder_env_follower_cutoff_ratio);
  read_param(p.float_params, "vocoder_modulator_window_size_seconds", cfg.vocoder_modulator_window_size_seconds);
  read_param(p.float_params, "vocoder_stride_seconds", cfg.vocoder_stride_seconds);
  read_param(p.int32_params, "vocoder_count_bands", cfg.vocoder_count_bands);
  read_param(p.float_params, "vocoder_min_freq", cfg.vocoder_min_freq);
  read_param(p.float_params, "vocoder_max_freq", cfg.vocoder_max_freq);
  read_param(p.float_params, "voice_volume", cfg.voice_volume);
  read_param(p.float_params, "carrier_volume", cfg.carrier_volume);
  read_param(p.float_params, "vocoder_volume", cfg.vocoder_volume);
  read_param(p.float_params, "analysis_volume", cfg.analysis_volume);
  read_param(p.float_params, "analysis_input_gain", cfg.analysis_input_gain);
  read_param(p.float_params, "analysis_output_feedback_gain", cfg.analysis_output_feedback_gain);
  read_param(p.float_params, "output_delay_seconds", cfg.output_delay_seconds);
}

// RtResynth::saveAsPreset (lib.cpp:1941-2041).  `carry` supplies the values of the keys this build does not interpret
// (e.g. the preset that was loaded); absent ones get the reference defaults of lib.cpp:943-999.
inline Preset save_as_preset(const ra_config &cfg, const Preset *carry = nullptr) {
  Preset p;
  auto carried = [&](const char *k, float dflt) {
    if (carry) { auto it = carry->float_params.find(k); if (it != carry->float_params.end()) return it->second; }
    return dflt;
  };
  p.float_params["pitch_wheel_multiplier"] = carried("pitch_wheel_multiplier", 2.f);
  p.float_params["window_size_seconds"] = static_cast<float>(cfg.window_size) / cfg.sample_rate;
  p.float_params["window_center_stride_seconds"] = static_cast<float>(cfg.stride) / cfg.sample_rate;
  p.float_params["min_volume"] = cfg.min_volume;
  p.float_params["nearby_distance_tones"] = cfg.nearby_distance_tones;
  p.float_params["max_track_pitches"] = cfg.max_track_pitches;
  p.float_params["autotune_tolerance_pitches"] = cfg.autotune_tolerance_pitches;
  p.float_params["pitch_shift_pre_autotune"] = cfg.pitch_shift_pre_autotune;
  p.float_params["pitch_shift_post_autotune"] = cfg.pitch_shift_post_autotune;
  p.float_params["pitch_harmonize_pre_autotune"] = cfg.pitch_harmonize_pre_autotune;
  p.float_params["pitch_harmonize_post_autotune"] = cfg.pitch_harmonize_post_autotune;
  p.float_params["stereo_spread"] = cfg.stereo_spread;
  p.float_params["env_attack_seconds"] = cfg.env_attack_seconds;
  p.float_params["env_hold_seconds"] = cfg.env_hold_seconds;
  p.float_params["env_decay_seconds"] = cfg.env_decay_seconds;
  p.float_params["env_release_seconds"] = cfg.env_release_seconds;
  p.float_params["env_sustain_level"] = cfg.env_sustain_level;
  p.enum_params["autotune_type"] = cfg.autotune_type;
  p.bool_params["use_autotune"] = cfg.use_autotune != 0;
  p.int32_params["autotune_max_pitch"] = cfg.autotune_max_pitch;
  p.int32_params["autotune_factor"] = cfg.autotune_factor;
  p.enum_params["autotune_musical_scale_mode"] = cfg.autotune_scale_mode;
  p.enum_params["autotune_musical_scale_root_note"] = cfg.autotune_root_note;
  p.int32_params["autotune_root_note_halftones_transpose"] = cfg.autotune_root_transpose;
  p.enum_params["autotune_chord_frequencies"] = cfg.autotune_chord_frequencies;
  p.uint64_params["autotune_bit_chord"] = cfg.autotune_bit_chord;
  p.float_params["vocoder_carrier_noise_volume"] = cfg.vocoder_carrier_noise_volume;
  p.float_params["vocoder_carrier_saw_volume"] = cfg.vocoder_carrier_saw_volume;
  p.float_params["vocoder_carrier_triangle_volume"] = cfg.vocoder_carrier_triangle_volume;
  p.float_params["vocoder_carrier_square_volume"] = cfg.vocoder_carrier_square_volume;
  p.float_params["vocoder_carrier_sine_volume"] = cfg.vocoder_carrier_sine_volume;
  p.float_params["vocoder_carrier_pulse_volume"] = cfg.vocoder_carrier_pulse_volume;
  p.float_params["vocoder_carrier_pulse_width"] = cfg.vocoder_carrier_pulse_width;
  p.float_params["vocoder_env_follower_cutoff_ratio"] = cfg.vocoder_env_follower_cutoff_ratio;
  p.float_params["vocoder_modulator_window_size_seconds"] = cfg.vocoder_modulator_window_size_seconds;
  p.float_params["vocoder_stride_seconds"] = cfg.vocoder_stride_seconds;
  p.int32_params["vocoder_count_bands"] = cfg.vocoder_count_bands;
  p.float_params["vocoder_min_freq"] = cfg.vocoder_min_freq;
  p.float_params["vocoder_max_freq"] = cfg.vocoder_max_freq;
  p.float_params["voice_volume"] = cfg.voice_volume;
  p.float_params["carrier_volume"] = cfg.carrier_volume;
  p.float_params["vocoder_volume"] = cfg.vocoder_volume;
  p.float_params["analysis_volume"] = cfg.analysis_volume;
  p.float_params["analysis_input_gain"] = cfg.analysis_input_gain;
  p.float_params["analysis_output_feedback_gain"] = cfg.analysis_output_feedback_gain;
  p.float_params["output_delay_seconds"] = cfg.output_delay_seconds;
  return p;
}

// ---- RtResynthOfflineJobConfig (params.cpp:183-213) ----------------------------------------------------------------------------
enum class Postprocessing { None, Limit }; // params.cpp:152-181

struct OfflineJobConfig {
  std::string preset_file, input_voice_file, input_carrier_file, output_file;
  Postprocessing post = Postprocessing::None;

  static OfflineJobConfig from_json(const json::Value &j) {
    auto str = [&](const char *k) -> std::string {
      const json::Value &v = j.at(k);
      if (v.kind != json::Value::String) throw std::runtime_error(std::string("job config: '") + k + "' is not a string");
      return v.str;
    };
    OfflineJobConfig c;
    c.preset_file = str("preset_file");
    c.input_voice_file = str("input_voice_file");
    c.input_carrier_file = str("input_carrier_file");
    c.output_file = str("output_file");
    const std::string post = str("post");
    if (post == "none") c.post = Postprocessing::None;
    else if (post == "limit") c.post = Postprocessing::Limit;
    else throw std::runtime_error("job config: unknown post '" + post + "'"); // from_string, params.cpp:170-181
    return c;
  }
  static OfflineJobConfig from_file(const std::string &path) { return from_json(json::parse_file(path)); }
  std::string to_json_text() const {
    std::ostringstream os;
    os << "{\n  \"input_carrier_file\": " << json::quote(input_carrier_file) << ",\n  \"input_voice_file\": " << json::quote(input_voice_file)
       << ",\n  \"output_file\": " << json::quote(output_file) << ",\n  \"post\": " << json::quote(post == Postprocessing::None ? "none" : "limit")
       << ",\n  \"preset_file\": " << json::quote(preset_file) << "\n}\n";
    return os.str();
  }
};

// The flow of the reference's offline job executable: job config -> RtResynthOfflineJob (WAV files) -> preset -> RtResynth(job).
// `base` carries what a preset does not (n_channels, seeds, phase policy...); paths in the config are used as written.
inline ra_stats run_offline_job(const OfflineJobConfig &jc, ra_config base, int device = 0) {
  base.post = jc.post == Postprocessing::Limit ? RA_POST_LIMIT : RA_POST_NONE; // init_post(job.getPostprocessing()), lib.cpp:1198,1245-1275
  wav::OfflineJob job(jc.input_voice_file, jc.input_carrier_file, jc.output_file);
  base.sample_rate = job.get_samplerate();
  if (!jc.preset_file.empty()) apply_preset(Preset::from_file(jc.preset_file), base);
  RtResynthAmd rs(job, base, device);
  job.flush();
  return rs.stats();
}
inline ra_stats run_offline_job_file(const std::string &job_config_path, const ra_config &base, int device = 0) {
  return run_offline_job(OfflineJobConfig::from_file(job_config_path), base, device);
}

} // namespace resynth_amd
