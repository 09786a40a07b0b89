// SURVEY.md section 8(e) in its own words — "one host thread + one HIP stream per device" — in ONE process: every thread owns a context on its
// device (thread t -> device t mod ra_device_count(): on a one-GPU box both land on device 0), launches on a HIP stream of its own and runs
// its contiguous shard of the streams, per-stream seeds = seed + global stream index.  No collective, no shared state between the threads
// beyond the library's per-device one-time function attributes (ra_once_per_device) and the thread-local error string.
//   device_threads <dir> <streams> <frames> <threads>     reads <dir>/voice.f32 [streams][frames], writes <dir>/out_<t>.f32 [shard][frames][2]
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "../../cpp_audio_amd/host/resynth_amd.hpp"

using namespace resynth_amd;

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  const std::string dir = argv[1];
  const int S = std::atoi(argv[2]), n = std::atoi(argv[3]), T = std::atoi(argv[4]);
  std::vector<float> voice((size_t)S * n);
  {
    std::ifstream f(dir + "/voice.f32", std::ios::binary);
    f.read(reinterpret_cast<char *>(voice.data()), (std::streamsize)(voice.size() * sizeof(float)));
    if (!f) { std::fprintf(stderr, "device_threads: cannot read voice.f32\n"); return 2; }
  }
  const int ndev = ra_device_count();
  if (ndev < 1) { std::fprintf(stderr, "device_threads: no HIP device\n"); return 2; }
  const int per = (S + T - 1) / T; // contiguous blocks of ceil(S / T) streams (cpp_audio_amd/sharding.py)
  std::vector<int> rc(T, 0);
  std::vector<std::string> err(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      try {
        const int lo = std::min(S, t * per), hi = std::min(S, lo + per);
        if (hi <= lo) return;
        const int device = t % ndev;
        if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice");
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("hipStreamCreate");
        ra_config cfg = default_config();
        cfg.window_size = 1024; cfg.stride = 256; cfg.n_channels = 32;
        cfg.analysis_volume = 1.f; cfg.stereo_spread = 0.f; cfg.carrier_volume = 0.f;
        cfg.seed = 1234 + (uint64_t)lo;
        {
          Context ctx(cfg, hi - lo, device);
          check(ra_set_hip_stream(ctx.get(), st));
          std::vector<float> out((size_t)(hi - lo) * n * 2), in, part;
          // two ragged blocks: the streams' state carries over inside the thread's context
          const int cut = n / 3 + 7 * t;
          int64_t t0 = 0;
          for (int64_t m : {(int64_t)cut, (int64_t)(n - cut)}) {
            in.resize((size_t)(hi - lo) * m); part.resize((size_t)(hi - lo) * m * 2);
            for (int s = lo; s < hi; ++s) std::copy(voice.begin() + (size_t)s * n + t0, voice.begin() + (size_t)s * n + t0 + m, in.begin() + (size_t)(s - lo) * m);
            ctx.process(in.data(), nullptr, m, part.data());
            for (int s = lo; s < hi; ++s) std::copy(part.begin() + (size_t)(s - lo) * m * 2, part.begin() + (size_t)(s - lo + 1) * m * 2, out.begin() + ((size_t)(s - lo) * n + t0) * 2);
            t0 += m;
          }
          if (ctx.stats().error_flags != 0) throw std::runtime_error("device error flags");
          std::ofstream f(dir + "/out_" + std::to_string(t) + ".f32", std::ios::binary);
          f.write(reinterpret_cast<const char *>(out.data()), (std::streamsize)(out.size() * sizeof(float)));
        }
        (void)hipStreamDestroy(st);
      } catch (const std::exception &e) { rc[t] = 1; err[t] = e.what(); }
    });
  for (auto &x : th) x.join();
  for (int t = 0; t < T; ++t) if (rc[t]) { std::fprintf(stderr, "device_threads: thread %d: %s\n", t, err[t].c_str()); return 1; }
  std::printf("devices %d threads %d\n", ndev, T);
  return 0;
}
