// K3 — additive sine bank: one workgroup per stream, one lane per synth channel (partial).
//
// Replaces SimpleAudioOutContext::step's one-shot drain + ImplCRTP::compute (reference
// include/simple_audio_context.h:69-112, include/gen.crtp.h:350-378), the per-sample element chain
// (ra_chan.cuh) and the event side effects of ImplCRTP::onEvent (gen.crtp.h:382-622) incl. setPhase (:128-156).
//
// Each lane keeps its channel's whole state in registers for the block and runs the reference's serial
// recurrences in fp64.  Outputs of a 16-frame tile are transposed through LDS and summed over channels by
// 32 lanes (one per frame x output), so the cross-lane reduction costs ~2 LDS ops per channel-sample
// instead of a 64-lane butterfly per sample.  HBM traffic: 8 B/frame written, per-frame commands read once.
#include "ra_chan.cuh"

extern __shared__ __align__(16) unsigned char ra_smem_syn[];

namespace {

// LDS carve for a workgroup of Ppad lanes
struct SynLds {
  double *tile;   // [32][Ppad + 1]  row = out * 16 + frame
  double *x_inc;  // [Ppad] angleIncrements() of every channel, for setPhase
  double *x_cre;  // [Ppad]
  double *x_cim;  // [Ppad]
  int *x_act;     // [Ppad]
  int *ev_lane;   // [2 * Ppad] lane holding the event of a given order index (-1 = none)
};
__device__ inline SynLds carve(unsigned char *base, int Ppad) {
  SynLds s;
  s.tile = reinterpret_cast<double *>(base);
  s.x_inc = s.tile + 32 * (Ppad + 1);
  s.x_cre = s.x_inc + Ppad;
  s.x_cim = s.x_cre + Ppad;
  s.x_act = reinterpret_cast<int *>(s.x_cim + Ppad);
  s.ev_lane = s.x_act + Ppad;
  return s;
}

__global__ void ra_synth_kernel(RaParams p, RaBuffers b, int64_t t0, int64_t n, int64_t f_lo, int64_t n_frames_blk, float *__restrict__ out,
                                double *__restrict__ out_d) {
  const int s = blockIdx.x;
  const int l = threadIdx.x;
  const int Ppad = p.Ppad;
  const int stride1 = Ppad + 1;
  SynLds L = carve(ra_smem_syn, Ppad);

  double *sd = b.sd + (size_t)s * SD_COUNT * Ppad;
  int *si = b.si + (size_t)s * SI_COUNT * Ppad;
  Chan c;
  load_chan(c, sd, si, Ppad, l);

  const RaEnvParams ep = {p.envA, p.envH, p.envD, p.envR, p.SMinusOne};
  const double dur = (double)p.H;                       // ctrl.setup(stride), lib.cpp:110-111
  const double max_filter_inc = 2. / (double)p.H;       // setMaxFilterIncrement, lib.cpp:106
  const uint64_t seed = p.seed + (uint64_t)s;
  const double *mix = (p.voc_enabled && b.voc_mix) ? b.voc_mix + (size_t)s * n : nullptr;

  int64_t t = t0;
  for (int64_t fb = 0; fb <= n_frames_blk; ++fb) {
    const int64_t seg_end = (fb < n_frames_blk) ? ra_fire_sample(p.N, p.H, f_lo + fb) : t0 + n;

    // ---- render [t, seg_end) in tiles of <= 16 frames -------------------------------------------
    while (t < seg_end) {
      const int m = (int)min((int64_t)RA_N_FRAMES_PER_BUFFER, seg_end - t);
      for (int i = 0; i < m; ++i) {
        double v = 0.0;
        if (rt_active(c)) { // ImplCRTP::compute, gen.crtp.h:352-371
          chan_step(c, ep, dur, max_filter_inc, 0);
          if (rt_active(c)) v = c.vol * ((c.alias * c.cim) * c.value); // :1135 over :231 over :2424
        }
        L.tile[i * stride1 + l] = c.gL * v;        // StereoPanned::imag(0), :1382-1384
        L.tile[(16 + i) * stride1 + l] = c.gR * v;
      }
      __syncthreads();
      if (l < 64) {
        const int fr = l & 15, ch = (l >> 4) & 1, half = l >> 5;
        double acc = 0.0;
        if (fr < m) {
          const double *row = L.tile + (ch * 16 + fr) * stride1;
          const int c0 = half * (Ppad >> 1), c1 = c0 + (Ppad >> 1);
          for (int q = c0; q < c1; ++q) acc += row[q];
        }
        acc += __shfl_down(acc, 32);
        if (l < 32 && fr < m) {
          const int64_t rel = (t - t0) + fr;
          if (mix) acc += mix[rel];  // the vocoder compute runs after the synth compute (lib.cpp:1353,1377)
          if (out) out[((size_t)s * n + rel) * 2 + ch] = (float)acc; // simple_audio_context.h:105-109
          if (out_d) out_d[((size_t)s * n + rel) * 2 + ch] = acc;
        }
      }
      __syncthreads();
      t += m;
    }
    if (fb == n_frames_blk) break;

    // ---- the one-shots of frame fb run at the start of step(seg_end) ----------------------------
    const RaCmd cmd = b.cmd[((size_t)s * b.frames_cap + fb) * Ppad + l];
    for (int i = l; i < 2 * Ppad; i += Ppad) L.ev_lane[i] = -1;
    __syncthreads();
    if (cmd.type == RA_CMD_OFF) {
      // canHandleExplicitKeyReleaseNow(0) && onKeyReleased(0), gen.crtp.h:581-583
      if (env_can_release_now(c, 0)) env_on_key_released(c, 0);
    } else if (cmd.type == RA_CMD_CHANGE) { // gen.crtp.h:600-616
      float angleIncrements = 2 * cmd.freq / p.sample_rate;
      float angleIncrementsMult = angleIncrements * 1.f;
      c.target = (double)0.1f * (double)cmd.vel; c.flags |= FL_HAS_TARGET;
      chan_set_angle_increments(c, (double)angleIncrementsMult, 0);
    } else if (cmd.type == RA_CMD_ON) {
      L.ev_lane[cmd.ord] = l;
    }
    L.x_inc[l] = angle_increments(c, 0);
    L.x_cre[l] = c.cre;
    L.x_cim[l] = c.cim;
    L.x_act[l] = rt_active(c) ? 1 : 0;
    __syncthreads();
    for (int e = 0; e < 2 * Ppad; ++e) {
      const int el = L.ev_lane[e];
      if (el < 0) continue;
      if (el == l) {
        // -- onEvent(NoteOn), gen.crtp.h:389-449 --
        if (c.state != ENV_DONE2) atomicOr(b.error_flags, RA_ERR_BOOKKEEPING); // tryAcquire must succeed here
        c.state = ENV_SOON;
        chan_forget_past_signals(c);
        c.target = (double)0.1f * (double)cmd.vel; c.flags |= FL_HAS_TARGET;  // :425
        // ResynthElementInitializer, lib.cpp:96-118 (setAHDSR / max filter increment are uniform parameters here)
        c.cur = 0; // ctrl.setup: no f_result yet
        {
          float u = hash_uniform_pm1(seed, (uint64_t)cmd.noteid, 1);
          double pan = (double)(p.stereo_spread * u);
          pan = fmin(pan, 1.); pan = fmax(pan, -1.);
          double ang = (0.25 * M_PI) * (pan + 1.);
          c.gL = cos(ang); c.gR = sin(ang);   // stereo(), stereo.h:18-27
        }
        // setupAudioElement, gen.sine.h:9-21
        float inc_f = 2 * (cmd.freq * 1.f) / p.sample_rate;
        chan_set_angle_increments(c, (double)inc_f, 0);
        // -- the one-shot, gen.crtp.h:455-527: setPhase (:128-156) then onKeyPressed(0) --
        const double ai = angle_increments(c, 0);
        int src = -1;
        for (int q = 0; q < p.P; ++q) {
          if (q == l) continue;
          if (!L.x_act[q]) continue;
          if (L.x_inc[q] == ai) { src = q; break; }
        }
        if (src >= 0) { c.cre = L.x_cre[src]; c.cim = L.x_cim[src]; }
        else {
          float a = (p.phase_policy == RA_PHASE_HASH) ? hash_uniform_pm1(seed, (uint64_t)cmd.noteid, 0) : 0.f;
          double sn, cs;
          sincos(M_PI * (double)a, &sn, &cs);  // setStartAngle :2401-2403
          c.cre = cs; c.cim = sn;
        }
        env_on_key_pressed(c, 0);
        L.x_cre[l] = c.cre; L.x_cim[l] = c.cim; L.x_act[l] = 1; L.x_inc[l] = ai;
      }
      __syncthreads();
    }
  }
  store_chan(c, sd, si, Ppad, l);
}

} // namespace

void ra_launch_synth(const RaParams &p, const RaBuffers &b, int64_t t0, int64_t n, int64_t f_lo, int64_t n_frames_blk, float *d_out,
                     double *d_out_d, hipStream_t st) {
  if (n <= 0) return;
  size_t lds = (size_t)32 * (p.Ppad + 1) * 8 + (size_t)p.Ppad * (8 * 3 + 4 + 8) + 64;
  hipLaunchKernelGGL(ra_synth_kernel, dim3((unsigned)p.n_streams), dim3((unsigned)p.Ppad), lds, st, p, b, t0, n, f_lo, n_frames_blk, d_out,
                     d_out_d);
}
