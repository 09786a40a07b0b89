// test helper: decode a WAV with cpp_audio_amd/host/wav.hpp and dump it as text header + raw float64 (deinterlaced)
#include <cstdio>
#include "../../cpp_audio_amd/host/wav.hpp"
int main(int argc, char **argv) {
  if (argc < 3) return 2;
  try {
    auto d = resynth_amd::wav::read(argv[1]);
    std::printf("%d %d %d %d %zu\n", d.sample_rate, d.channels, d.bits, d.is_float ? 1 : 0, d.frames());
    FILE *f = std::fopen(argv[2], "wb");
    for (auto &ch : d.samples) std::fwrite(ch.data(), sizeof(double), ch.size(), f);
    std::fclose(f);
    if (argc > 3) { // round trip through the float64 writer
      std::vector<double> inter(d.frames() * d.channels);
      for (size_t i = 0; i < d.frames(); ++i) for (int c = 0; c < d.channels; ++c) inter[i * d.channels + c] = d.samples[c][i];
      resynth_amd::wav::write_f64(argv[3], inter.data(), d.frames(), d.channels, d.sample_rate);
    }
  } catch (std::exception const &e) { std::printf("error: %s\n", e.what()); return 1; }
  return 0;
}
