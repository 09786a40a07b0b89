// GPU test of the host shims (cpp_audio_amd/host/resynth_amd.hpp), driven the way the reference drives its own objects:
//  (a) batch Context::process
//  (b) the per-frame offline loop of rt.resynth.lib.cpp:1204-1234 with InputShim (input_f) + ComputeShim registered in a
//      minimal stand-in of SimpleAudioOutContext::step (zeroed double buffer, computes accumulate, cast to float)
//  (c) RtResynthAmd over a RtResynthOfflineJob-shaped WAV harness
//  (d) resynth.hpp's manual element loop (:17-99) over AmdElement, against render_notes
// Raw outputs are written next to argv[1] for the Python side to compare with the CPU oracle.
#include <cstdio>
#include <functional>
#include <memory>
#include "../../cpp_audio_amd/host/resynth_amd.hpp"

using namespace resynth_amd;

static std::vector<float> make_voice(int n, int sr) {
  std::vector<float> v(n);
  for (int i = 0; i < n; ++i) {
    double t = double(i) / sr;
    double g = (t > 0.02 && t < 0.2) ? 1.0 : 0.3;
    v[i] = float(0.25 * g * std::sin(2 * M_PI * 440.0 * t) + 0.1 * std::sin(2 * M_PI * (1200.0 + 300 * t) * t) + 0.05 * std::sin(2 * M_PI * 3000.7 * t));
  }
  return v;
}
static void dump(const std::string &path, const void *p, size_t bytes) {
  FILE *f = std::fopen(path.c_str(), "wb");
  std::fwrite(p, 1, bytes, f);
  std::fclose(f);
}
struct Note { int startFrame, endFrame; double frequency, amplitude; };

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  try {
    const int sr = 48000, n = 14000;
    ra_config cfg = default_config();
    cfg.window_size = 1024; cfg.stride = 256; cfg.n_channels = 16;
    cfg.analysis_volume = 1.f; cfg.stereo_spread = 0.f; cfg.carrier_volume = 0.f; cfg.seed = 99;
    std::vector<float> voice = make_voice(n, sr);
    dump(dir + "/voice.f32", voice.data(), voice.size() * 4);

    // (a)
    std::vector<float> out_a(2 * n);
    { Context ctx(cfg, 1); ctx.process(voice.data(), nullptr, n, out_a.data()); }
    dump(dir + "/out_a.f32", out_a.data(), out_a.size() * 4);

    // (b) the reference's callables
    std::vector<float> out_b(2 * n);
    {
      Context ctx(cfg, 1);
      std::function<void(const float *, int)> input_f = InputShim{ctx.get()};   // RecordF
      std::vector<std::function<bool(double *, int)>> computes;                  // SimpleAudioOutContext::computes
      computes.emplace_back(ComputeShim{ctx.get()});                             // registerSimpleCompute
      for (int k = 0; k < n; ++k) {
        input_f(&voice[k], 1);
        double buf[2] = {0, 0};
        for (auto &c : computes) if (!c(buf, 1)) return 3;
        out_b[2 * k] = float(buf[0]); out_b[2 * k + 1] = float(buf[1]);
      }
    }
    double err_ab = 0;
    for (int i = 0; i < 2 * n; ++i) err_ab = std::max(err_ab, std::fabs(double(out_a[i]) - out_b[i]));

    // (c) offline job over WAV files
    wav::write_f32(dir + "/voice.wav", voice.data(), n, 1, sr);
    {
      wav::OfflineJob job(dir + "/voice.wav", "", dir + "/job_out.wav");
      RtResynthAmd r(job, cfg, 0, 5000); // several blocks
      if (r.frames() != n) { std::printf("job frames %lld\n", (long long)r.frames()); return 4; }
      job.flush();
    }
    wav::Data jo = wav::read(dir + "/job_out.wav");
    double err_ac = 0;
    if (jo.channels != 2 || (int)jo.frames() != n || !jo.is_float) return 5;
    for (int i = 0; i < n; ++i) for (int c = 0; c < 2; ++c) err_ac = std::max(err_ac, std::fabs(jo.samples[c][i] - out_a[2 * i + c]));

    // (d) driver (A): batch renderer vs the reference's manual loop over the element duck type
    std::vector<Note> notes = {{0, 3, 440.0, -12.0}, {2, 2, 660.0, -20.0}, {2, 9, 95.0, -15.0}, {40, 41, 440.0, -18.0}, {41, 60, 3000.0, -25.0}};
    const int stride = 400, fs = 44100;
    std::vector<double> batch = render_notes(notes, stride, fs);
    std::vector<double> manual;
    {
      std::vector<std::unique_ptr<AmdElement>> pool;
      auto getAvailableOscillator = [&pool]() {
        for (auto &p : pool) if (!p->getEnvelope().isEnvelopeRTActive()) return p.get();
        pool.push_back(std::make_unique<AmdElement>());
        return pool.back().get();
      };
      auto recordFrame = [&]() {
        double res = 0.; bool empty = true;
        for (auto const &p : pool) { if (!p->getEnvelope().isEnvelopeRTActive()) continue; empty = false; p->step(); res += p->imag(); }
        manual.push_back(res);
        return empty;
      };
      std::vector<Note> byStart = notes;
      for (auto &nt : byStart) { nt.startFrame *= stride; nt.endFrame *= stride; }
      std::stable_sort(byStart.begin(), byStart.end(), [](Note const &a, Note const &b) { return a.startFrame < b.startFrame; });
      int frame = 0;
      for (auto const &note : byStart) {
        for (; frame < note.startFrame; ++frame) recordFrame();
        AmdElement *ae = getAvailableOscillator();
        ae->forgetPastSignals();
        ae->setAngleIncrements(2 * note.frequency / fs);
        ae->getAlgo().getVolumeAdjustment().setVolumeTarget(std::pow(10.0, note.amplitude / 20.0));
        ae->editEnvelope().setAHDSR(AHDSR{1000, itp::LINEAR, 0, 1000, itp::LINEAR, 10000, itp::LINEAR, 0.7f}, fs);
        ae->onKeyPressed(0);
        ae->onKeyReleased(stride + note.endFrame - note.startFrame);
      }
      while (!recordFrame()) {}
    }
    double err_d = 0;
    if (manual.size() != batch.size()) { std::printf("driver A length %zu vs %zu\n", manual.size(), batch.size()); return 6; }
    for (size_t i = 0; i < batch.size(); ++i) err_d = std::max(err_d, std::fabs(manual[i] - batch[i]));
    dump(dir + "/notes.f64", batch.data(), batch.size() * 8);

    std::printf("err_ab %.3e err_ac %.3e err_d %.3e frames %d notes_len %zu\n", err_ab, err_ac, err_d, n, batch.size());
    if (err_ab > 1e-6 || err_ac > 1e-6 || err_d > 1e-9) return 7; // block splits differ at rounding level in hop-parallel mode
  } catch (std::exception const &e) { std::printf("exception: %s\n", e.what()); return 1; }
  return 0;
}
