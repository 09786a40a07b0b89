// Driver (A) end to end on the HIP backend: WAV -> notes -> WAV (source/main.resynth.cpp:5-81 without the plot).
//   resynth_wav in.wav out.wav [half_window_size] [stride]   prints one line per rendered note: start end frequency amplitude
#include <cstdio>
#include <cstdlib>

#include "../../cpp_audio_amd/host/resynth_amd.hpp"

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: resynth_wav in.wav out.wav [half_window_size] [stride]\n"); return 2; }
  try {
    auto notes = resynth_amd::resynth_wav(argv[1], argv[2], argc > 3 ? std::atoi(argv[3]) : 400, argc > 4 ? std::atoi(argv[4]) : 400);
    for (auto const &n : notes) std::printf("%d %d %.17g %.17g\n", n.startFrame, n.endFrame, n.frequency, n.amplitude);
  } catch (std::exception const &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
