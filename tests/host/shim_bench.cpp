// Throughput of the reference's real drop-in seam — RecordF + ComputeFunc (include/simple_audio_context.h:24-26, 80-102) — through the C ABI:
// per 16-frame tile every stream is fed, then every stream's compute runs into the caller's zeroed double buffer, as SimpleAudioOutContext::step
// does.  Prints one JSON line per stream count: x real time = audio seconds of EVERY stream / wall seconds.
//   shim_bench <streams> <seconds> [<streams> <seconds> ...]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../cpp_audio_amd/host/resynth_amd.hpp"

using namespace resynth_amd;

int main(int argc, char **argv) {
  try {
    for (int a = 1; a + 1 < argc; a += 2) {
      const int S = std::atoi(argv[a]);
      const double seconds = std::atof(argv[a + 1]);
      const int sr = 48000, n = int(seconds * sr);
      ra_config cfg = default_config();
      cfg.window_size = 4096; cfg.stride = 1024; cfg.n_channels = 64;
      cfg.analysis_volume = 1.f; cfg.stereo_spread = 0.f; cfg.carrier_volume = 0.f; cfg.seed = 7;
      Context ctx(cfg, S);
      std::vector<std::vector<float>> voice(S, std::vector<float>(n));
      for (int s = 0; s < S; ++s)
        for (int i = 0; i < n; ++i) {
          const double t = double(i) / sr;
          voice[s][i] = float(0.2 * std::sin(2 * M_PI * (220.0 + 17.0 * s) * t) + 0.1 * std::sin(2 * M_PI * (1200.0 + 300 * t) * t) + 0.05 * std::sin(2 * M_PI * 3000.7 * t));
        }
      std::vector<InputShim> in;
      std::vector<ComputeShim> comp;
      for (int s = 0; s < S; ++s) { in.push_back(InputShim{ctx.get(), nullptr, s}); comp.push_back(ComputeShim{ctx.get(), s}); }
      double energy = 0.0;
      const int warm = std::min(n / 4, sr / 2);
      auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < n; k += 16) {
        if (k == warm - warm % 16) t0 = std::chrono::steady_clock::now(); // (allocations of the first blocks are not the steady state)
        const int m = std::min(16, n - k);
        for (int s = 0; s < S; ++s) in[s](&voice[s][k], m);
        for (int s = 0; s < S; ++s) {
          double buf[32] = {0};
          if (!comp[s](buf, m)) return 3;
          energy += buf[0] * buf[0];
        }
      }
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const double audio = double(n - (warm - warm % 16)) / sr;
      std::printf("{\"streams\": %d, \"audio_seconds_per_stream\": %.3f, \"wall_seconds\": %.6f, \"x_realtime\": %.2f, \"us_per_compute\": %.3f, \"energy\": %.6g}\n", S, audio, dt,
                  audio / dt, dt * 1e6 / ((n - (warm - warm % 16)) / 16.0 * S), energy);
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "shim_bench: %s\n", e.what());
    return 1;
  }
  return 0;
}
