// K8 — post-processing `post: "limit"` of the offline job (reference source/rt.resynth.lib.cpp:1247-1258, source/rt.resynth.lib.params.cpp:152-176):
// Limiter<double>::feedOneFrame on every stereo frame of the double mix, before the mono publication (:1262-1272) and the float cast
// (include/simple_audio_context.h:105-109).
//
// The Limiter lives in the absent cpp.algorithms: PARITY UNPINNED.  Certain from the call sites: default-constructed (lib.cpp:908: no
// sample rate, so its time constants are in frames), fed one stereo frame at a time, in place; it "compresses and limits" (:1258) and
// exposes a target compression level (:891).  Decision, the same in the CPU restatement:
//   peak   = max(|L|, |R|)                                    (one gain for both channels: the stereo image is kept)
//   needed = peak > 0.95 ? 0.95 / peak : 1
//   target = needed < target ? needed : target + (1 - target) * 2e-5     (drops at once, recovers with a ~50 000-frame e-fold)
//   compression += (target - compression) * 0.01                         (follows its target with a ~100-frame lag: no gain steps)
//   out    = clamp(in * compression, -1, 1)                              (what the lag lets through is clipped: "limited")
// One wavefront per stream, frames in order (a recurrence through a comparison); a tile of 64 frames is loaded and stored coalesced, the
// recurrence runs on wave-uniform values.
#include "ra_internal.h"

namespace {

__device__ inline double readlane_f64(double v, int lane) { // lane is wave-uniform
  const long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffll), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ void __launch_bounds__(64) ra_limiter_kernel(RaParams p, RaBuffers b, int64_t r0, int64_t r1, int64_t n, float *__restrict__ out) {
  const int s = blockIdx.x, l = threadIdx.x;
  double comp = b.lim_state[2 * s], target = b.lim_state[2 * s + 1];
  const double2 *in = reinterpret_cast<const double2 *>(b.post_d) + (size_t)s * n;
  float *o = out ? out + (size_t)s * n * 2 : nullptr;
  float *fb = b.fb_block ? b.fb_block + (size_t)s * n : nullptr;
  auto fetch = [&](int64_t t) { return (t + l < r1) ? in[t + l] : make_double2(0.0, 0.0); };
  double2 nxt = fetch(r0);
  for (int64_t t = r0; t < r1; t += 64) {
    const int cnt = (int)min((int64_t)64, r1 - t);
    const double2 cur = nxt;
    nxt = fetch(t + 64); // requested before the tile's stores are issued
    double oL = 0.0, oR = 0.0;
    for (int i = 0; i < cnt; ++i) {
      const double a = readlane_f64(cur.x, i), c = readlane_f64(cur.y, i);
      const double peak = fmax(fabs(a), fabs(c));
      double needed = 1.0;
      if (peak > 0.95) needed = 0.95 / peak; // wave-uniform
      target = needed < target ? needed : target + (1.0 - target) * 2e-5;
      comp = comp + (target - comp) * 0.01;
      double x = a * comp, y = c * comp;
      x = fmin(1.0, fmax(-1.0, x)); y = fmin(1.0, fmax(-1.0, y));
      if (l == i) { oL = x; oR = y; }
    }
    if (l < cnt) {
      if (o) *reinterpret_cast<float2 *>(o + (t + l) * 2) = make_float2((float)oL, (float)oR);
      if (fb) fb[t + l] = (float)(oL + oR); // the mono publication sees the limited frame (lib.cpp:1262-1272)
    }
  }
  if (l == 0) { b.lim_state[2 * s] = comp; b.lim_state[2 * s + 1] = target; }
}

} // namespace

void ra_launch_limiter(const RaParams &p, const RaBuffers &b, int64_t r0, int64_t r1, int64_t n, float *d_out, hipStream_t st) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(ra_limiter_kernel, dim3((unsigned)p.n_streams), dim3(64), 0, st, p, b, r0, r1, n, d_out);
}
