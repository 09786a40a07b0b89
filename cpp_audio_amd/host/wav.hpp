// Minimal RIFF/WAVE reader and writer for the offline harnesses of the hot path.
// Stands in for cpp.algorithms' WAVReader / WAVWriter as used by the reference at
// source/rt.resynth.lib.params.cpp:297-313,337-372 (mono inputs "as one float", stereo IEEE-float32 output) and
// include/resynth.hpp:29-35 (mono IEEE-float64 output).  Formats: PCM 16/24/32-bit signed, IEEE float 32/64,
// plain or WAVE_FORMAT_EXTENSIBLE headers, any channel count.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace resynth_amd {
namespace wav {

struct Data {
  int sample_rate = 0;
  int channels = 0;
  int bits = 0;
  bool is_float = false;
  std::vector<std::vector<double>> samples; // deinterlaced [channel][frame], PCM scaled to [-1, 1)
  size_t frames() const { return samples.empty() ? 0 : samples[0].size(); }
};

namespace detail {
inline uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
inline uint16_t rd16(const unsigned char *p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }
inline void wr32(std::vector<unsigned char> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((x >> (8 * i)) & 0xff); }
inline void wr16(std::vector<unsigned char> &v, uint16_t x) { for (int i = 0; i < 2; ++i) v.push_back((x >> (8 * i)) & 0xff); }
} // namespace detail

inline Data read(const std::string &path) {
  using namespace detail;
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("failed to read wav file '" + path + "'");
  std::vector<unsigned char> buf;
  unsigned char tmp[65536];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  std::fclose(f);
  if (buf.size() < 12 || std::memcmp(buf.data(), "RIFF", 4) || std::memcmp(buf.data() + 8, "WAVE", 4)) throw std::runtime_error("not a RIFF/WAVE file: " + path);
  Data d;
  int fmt_tag = 0, block_align = 0;
  size_t pos = 12, data_pos = 0, data_len = 0;
  bool have_fmt = false;
  while (pos + 8 <= buf.size()) {
    uint32_t len = rd32(&buf[pos + 4]);
    const unsigned char *body = &buf[pos + 8];
    if (!std::memcmp(&buf[pos], "fmt ", 4) && len >= 16 && pos + 8 + len <= buf.size()) {
      fmt_tag = rd16(body);
      d.channels = rd16(body + 2);
      d.sample_rate = static_cast<int>(rd32(body + 4));
      block_align = rd16(body + 12);
      d.bits = rd16(body + 14);
      if (fmt_tag == 0xFFFE && len >= 26) fmt_tag = rd16(body + 24); // extensible: first two bytes of the sub-format GUID
      have_fmt = true;
    } else if (!std::memcmp(&buf[pos], "data", 4)) {
      data_pos = pos + 8;
      data_len = std::min<size_t>(len, buf.size() - data_pos);
      break;
    }
    pos += 8 + len + (len & 1);
  }
  if (!have_fmt || !data_pos || d.channels < 1) throw std::runtime_error("wav: missing fmt/data chunk: " + path);
  d.is_float = fmt_tag == 3;
  if (!(fmt_tag == 1 || fmt_tag == 3)) throw std::runtime_error("wav: unsupported format tag");
  const int bytes = d.bits / 8;
  if (block_align != bytes * d.channels || !(bytes == 2 || bytes == 3 || bytes == 4 || bytes == 8)) throw std::runtime_error("wav: unsupported sample size");
  const size_t frames = data_len / block_align;
  d.samples.assign(d.channels, std::vector<double>(frames));
  for (size_t i = 0; i < frames; ++i)
    for (int c = 0; c < d.channels; ++c) {
      const unsigned char *p = &buf[data_pos + i * block_align + static_cast<size_t>(c) * bytes];
      double v;
      if (d.is_float) {
        if (bytes == 4) { float x; std::memcpy(&x, p, 4); v = x; }
        else if (bytes == 8) { std::memcpy(&v, p, 8); }
        else throw std::runtime_error("wav: unsupported float width");
      } else if (bytes == 2) v = static_cast<int16_t>(rd16(p)) / 32768.0;
      else if (bytes == 3) { int32_t x = (p[0] << 8) | (p[1] << 16) | (static_cast<int32_t>(static_cast<int8_t>(p[2])) << 24); v = (x >> 8) / 8388608.0; }
      else if (bytes == 4) v = static_cast<int32_t>(rd32(p)) / 2147483648.0;
      else throw std::runtime_error("wav: unsupported pcm width");
      d.samples[c][i] = v;
    }
  return d;
}

template <typename T>
inline void write_float(const std::string &path, const T *interleaved, size_t frames, int channels, int sample_rate) {
  using namespace detail;
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float32 or float64");
  std::vector<unsigned char> h;
  const uint32_t data_len = static_cast<uint32_t>(frames * channels * sizeof(T));
  h.insert(h.end(), {'R', 'I', 'F', 'F'});
  wr32(h, 36 + data_len);
  h.insert(h.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '});
  wr32(h, 16);
  wr16(h, 3); // IEEE float
  wr16(h, static_cast<uint16_t>(channels));
  wr32(h, static_cast<uint32_t>(sample_rate));
  wr32(h, static_cast<uint32_t>(sample_rate * channels * sizeof(T)));
  wr16(h, static_cast<uint16_t>(channels * sizeof(T)));
  wr16(h, static_cast<uint16_t>(8 * sizeof(T)));
  h.insert(h.end(), {'d', 'a', 't', 'a'});
  wr32(h, data_len);
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("failed to write wav file");
  std::fwrite(h.data(), 1, h.size(), f);
  std::fwrite(interleaved, sizeof(T), frames * channels, f);
  std::fclose(f);
}
inline void write_f32(const std::string &path, const float *interleaved, size_t frames, int channels, int sr) { write_float(path, interleaved, frames, channels, sr); }
inline void write_f64(const std::string &path, const double *interleaved, size_t frames, int channels, int sr) { write_float(path, interleaved, frames, channels, sr); }

// RtResynthOfflineJob-shaped harness over WAV files (rt.resynth.lib.params.cpp:249-386): mono inputs only, stereo float32 output
class OfflineJob {
public:
  OfflineJob(const std::string &voice_file, const std::string &carrier_file, const std::string &output_file) : out_path_(output_file) {
    if (output_file.empty()) throw std::runtime_error("no output file");
    if (!voice_file.empty()) load(voice_file, voice_);
    if (!carrier_file.empty()) load(carrier_file, carrier_);
    if (!voice_file.empty() && !carrier_file.empty() && voice_.sample_rate != carrier_.sample_rate)
      throw std::runtime_error("sample rate mismatch between carrier and voice");
    if (voice_file.empty() && carrier_file.empty()) throw std::runtime_error("must have at least one of carrier or voice");
    sample_rate_ = voice_file.empty() ? carrier_.sample_rate : voice_.sample_rate;
  }
  ~OfflineJob() { try { flush(); } catch (...) {} }
  int get_samplerate() const { return sample_rate_; }
  bool has_more_voice() const { return vpos_ < voice_.frames(); }
  bool has_more_carrier() const { return cpos_ < carrier_.frames(); }
  template <typename T> void read_voice(T &v) { v = has_more_voice() ? static_cast<T>(voice_.samples[0][vpos_++]) : T(0); }
  template <typename T> void read_carrier(T &v) { v = has_more_carrier() ? static_cast<T>(carrier_.samples[0][cpos_++]) : T(0); }
  int count_outputs() const { return 2; }
  void write_output(const float *v, int nSamples) { out_.insert(out_.end(), v, v + nSamples); }
  void flush() {
    if (flushed_) return;
    write_f32(out_path_, out_.data(), out_.size() / 2, 2, sample_rate_);
    flushed_ = true;
  }

private:
  static void load(const std::string &path, Data &d) {
    d = read(path);
    if (d.channels != 1) throw std::runtime_error("single channel only");
  }
  Data voice_, carrier_;
  size_t vpos_ = 0, cpos_ = 0;
  int sample_rate_ = 0;
  std::string out_path_;
  std::vector<float> out_;
  bool flushed_ = false;
};

} // namespace wav
} // namespace resynth_amd
