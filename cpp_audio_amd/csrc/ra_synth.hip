// K3 — additive sine bank: one workgroup per stream, one lane per synth channel (partial).
//
// Replaces SimpleAudioOutContext::step's one-shot drain + ImplCRTP::compute (reference
// include/simple_audio_context.h:69-112, include/gen.crtp.h:350-378) and, inside it, the per-sample
// element chain ResynthElement<double> (source/rt.resynth.lib.cpp:6-19):
//   StereoPanned (include/audioelement.h:1319-1389) > VolumeAdjusted (:1110-1265) > Enveloped (:181-291)
//   > AHDSREnvelope (:665-1102) + FreqCtrl_ (:3055-3155) > InterpolatedFreq (:2705-2817) + SineOscillatorAlgo (:2365-2452)
// and the event side effects of ImplCRTP::onEvent (gen.crtp.h:382-622) incl. setPhase (:128-156).
//
// Each lane keeps its channel's whole state in registers for the block and runs the reference's serial
// recurrences in fp64.  Outputs of a 16-frame tile are transposed through LDS and summed over channels by
// 32 lanes (one per frame x output), so the cross-lane reduction costs ~2 LDS ops per channel-sample
// instead of a 64-lane butterfly per sample.  HBM traffic: 8 B/frame written, per-frame commands read once.
#include "ra_internal.h"

extern __shared__ __align__(16) unsigned char ra_smem_syn[];

namespace {

struct Chan {
  // AHDSREnvelope
  int state, counter, reldelay, ahdstate, ahdcounter, minch, nextminch;
  double value, top;
  // InterpolatedFreq
  double from, to, cur, C, fres;
  // SineOscillatorAlgo
  double cre, cim, mre, mim, alias, inc;
  // BaseVolumeAdjusted
  double vol, target, finit, k, y;
  // StereoPanned
  double gL, gR;
  int flags;
};

__device__ inline bool rt_active(const Chan &c) { return c.state != ENV_SOON && c.state != ENV_DONE2; } // audioelement.h:390-395

__device__ inline int env_max_counter(const Chan &c, const RaParams &p) { // :1066-1084
  switch (c.ahdstate) {
    case 0: return max(c.minch, p.envA);
    case 1: return p.envH;
    default: return (p.envD == 0) ? 0 : max(c.minch, p.envD);
  }
}
// onAHDStateChange (:1040-1053) and stepAHD (:1055-1064) call each other; unrolled into a loop
__device__ inline void env_on_ahd_change(Chan &c, const RaParams &p) {
  for (;;) {
    c.ahdcounter = 0;
    c.minch = c.nextminch;
    if (c.ahdstate < 0) return;
    if (env_max_counter(c, p) != 0) return;
    c.ahdstate = (c.ahdstate == 2) ? -1 : c.ahdstate + 1; // stepAHD with ahdCounter(0) >= maxCounter(0)
  }
}
__device__ inline void env_step_ahd(Chan &c, const RaParams &p) {
  if (c.ahdcounter < env_max_counter(c, p)) return;
  c.ahdstate = (c.ahdstate == 2) ? -1 : c.ahdstate + 1;
  env_on_ahd_change(c, p);
}
__device__ inline double itp_linear(double t, double b, double cc, double d) { return cc * t / d + b; }

__device__ inline void env_step(Chan &c, const RaParams &p) { // EnvelopeCRT::step :702-756
  if (c.reldelay >= 0) {
    if (c.reldelay == 0) { // startReleased :836-841
      c.minch = c.nextminch;
      c.state = ENV_RELEASED;
      c.top = c.value;
      c.counter = 0;
    }
    --c.reldelay;
  }
  ++c.counter;
  switch (c.state) {
    case ENV_PRESSED: {
      if (c.counter < 1) return;
      if (c.counter == 1) { c.ahdstate = 0; env_on_ahd_change(c, p); } // startPressed :955-958
      // stepPressed :960-1000
      if (c.ahdstate >= 0) { env_step_ahd(c, p); ++c.ahdcounter; }
      if (c.ahdstate < 0) { c.value = 1.0 + (double)p.SMinusOne; }
      else {
        double from, tmf;
        switch (c.ahdstate) {
          case 0: from = 0; tmf = 1; break;
          case 1: from = 1; tmf = 0; break;
          default: from = 1; tmf = (double)p.SMinusOne; break;
        }
        c.value = itp_linear((double)c.ahdcounter, from, tmf, (double)env_max_counter(c, p));
      }
    } break;
    case ENV_RELEASED: {
      int rt = max(c.minch, p.envR); // getReleaseTime :1002-1004
      c.value = itp_linear((double)c.counter, c.top, -c.top, (double)rt);
      if (c.counter >= rt) { c.state = ENV_DONE1; c.counter = 0; c.value = 0; }
    } break;
    case ENV_DONE1:
      if (c.counter > RA_N_FRAMES_PER_BUFFER) { c.state = ENV_DONE2; c.counter = 0; }
      c.value = 0;
      break;
    default: c.value = 0; break;
  }
}

__device__ inline double ctrl_prop(double from, double to) { return (to == from) ? (double)1.f : -log(from / to) / (to - from); } // :2804-2816

__device__ inline double ctrl_step(Chan &c, double dur) { // InterpolatedFreq::step :2764-2788
  if ((c.flags & FL_HAS_FRES) && c.fres == c.to) return c.to;
  if (c.cur + (double).5f > dur) c.cur = dur;
  c.fres = c.from + (c.to - c.from) * (c.cur / dur);
  c.flags |= FL_HAS_FRES;
  if (c.cur < dur) {
    double f = c.from + (c.to - c.from) * (c.cur + (double).5f) / dur;
    c.cur += c.C * f;
  }
  return c.fres;
}

__device__ inline double alias_mult(double inc) { // freqAliasingMultiplicator :466-483
  if (inc == 0.) return 1.;
  double halfN = 1. / inc;
  if (halfN < 1.) return 0.;
  if (halfN > 4.) return 1.;
  return (halfN - 1.) * (1. / 3.);
}

__device__ inline void osc_set_inc(Chan &c, double f) { // SineOscillatorAlgo::setAngleIncrements :2404-2411
  if (c.inc == f) return;
  c.inc = f;
  double sn, cs;
  sincos(M_PI * f, &sn, &cs);
  c.mre = cs;
  c.mim = sn;
  c.alias = alias_mult(f);
}

// FreqCtrl_::angleIncrements :3117-3124 (mode 0) / SineOscillatorAlgo::angleIncrements (driver A, mode 1)
__device__ inline double angle_increments(const Chan &c, int mode) {
  if (mode == 1) return c.inc;
  return c.inc != 0. ? c.inc : c.from;
}

// StereoPanned::step -> BaseVolumeAdjusted::step (:1195-1216) -> Enveloped::step (:197-200) -> FreqCtrl_::step (:3126-3129)
__device__ inline void chan_step(Chan &c, const RaParams &p, double dur, double max_filter_inc, int mode) {
  env_step(c, p);
  if (mode == 0) osc_set_inc(c, ctrl_step(c, dur));
  { // SineOscillatorAlgo::step :2413-2421 with approx_normalize :2443-2447
    double re = c.cre * c.mre - c.cim * c.mim;
    double im = c.cre * c.mim + c.cim * c.mre;
    double nn = re * re + im * im;
    double g = 0.5 * (3.0 - nn);
    c.cre = re * g;
    c.cim = im * g;
  }
  if (!(c.flags & FL_HAS_VOL)) {
    c.finit = fmin(max_filter_inc, fabs(angle_increments(c, mode)));
    double a = M_PI * c.finit; c.k = a / (1.0 + a); // one-pole low-pass coefficient (DESIGN.md §5, L0 decision)
    c.y = c.target;
  } else if (c.vol != c.target) {
    double inc = fmin(max_filter_inc, fabs(angle_increments(c, mode)));
    if (c.finit != inc) { c.finit = inc; double a = M_PI * inc; c.k = a / (1.0 + a); }
  }
  c.y = c.y + c.k * (c.target - c.y);
  c.vol = c.y;
  c.flags |= FL_HAS_VOL;
}

// Enveloped::setAngleIncrements :218-225 -> FreqCtrl_::setAngleIncrements :3103-3108 -> InterpolatedFreq::setAngleIncrements :2752-2762
__device__ inline void chan_set_angle_increments(Chan &c, double a, int mode) {
  double period = (a == 0.) ? 1. : 2. / fabs(a);
  c.nextminch = (int)(0.5f + 2.5 * period);
  if (mode == 1) { osc_set_inc(c, a); return; }
  c.cur = 0;
  c.to = a;
  c.from = (c.flags & FL_HAS_FRES) ? c.fres : a;
  c.C = ctrl_prop(c.from, c.to);
}

__device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// declared replacement of the reference's mersenne<> draws (gen.crtp.h:152, lib.cpp:116): see DESIGN.md §5
__device__ inline float hash_uniform_pm1(uint64_t seed, uint64_t noteid, uint64_t salt) {
  uint64_t h = splitmix64(seed ^ splitmix64(noteid * 2 + salt));
  uint32_t r = (uint32_t)(h >> 40);
  return -1.0f + 2.0f * ((float)r * (1.0f / 16777216.0f));
}

__device__ inline void load_chan(Chan &c, const double *sd, const int *si, int Ppad, int l) {
  c.value = sd[SD_ENV_VALUE * Ppad + l]; c.top = sd[SD_ENV_TOP * Ppad + l];
  c.from = sd[SD_CTRL_FROM * Ppad + l]; c.to = sd[SD_CTRL_TO * Ppad + l]; c.cur = sd[SD_CTRL_CUR * Ppad + l];
  c.C = sd[SD_CTRL_C * Ppad + l]; c.fres = sd[SD_CTRL_FRES * Ppad + l];
  c.cre = sd[SD_OSC_CRE * Ppad + l]; c.cim = sd[SD_OSC_CIM * Ppad + l]; c.mre = sd[SD_OSC_MRE * Ppad + l];
  c.mim = sd[SD_OSC_MIM * Ppad + l]; c.alias = sd[SD_OSC_ALIAS * Ppad + l]; c.inc = sd[SD_OSC_INC * Ppad + l];
  c.vol = sd[SD_VOL * Ppad + l]; c.target = sd[SD_VOL_TARGET * Ppad + l]; c.finit = sd[SD_VOL_FINIT * Ppad + l];
  c.k = sd[SD_VOL_K * Ppad + l]; c.y = sd[SD_VOL_Y * Ppad + l]; c.gL = sd[SD_GAIN_L * Ppad + l]; c.gR = sd[SD_GAIN_R * Ppad + l];
  c.state = si[SI_ENV_STATE * Ppad + l]; c.counter = si[SI_ENV_COUNTER * Ppad + l]; c.reldelay = si[SI_ENV_RELDELAY * Ppad + l];
  c.ahdstate = si[SI_ENV_AHDSTATE * Ppad + l]; c.ahdcounter = si[SI_ENV_AHDCOUNTER * Ppad + l]; c.minch = si[SI_ENV_MINCH * Ppad + l];
  c.nextminch = si[SI_ENV_NEXTMINCH * Ppad + l]; c.flags = si[SI_FLAGS * Ppad + l];
}
__device__ inline void store_chan(const Chan &c, double *sd, int *si, int Ppad, int l) {
  sd[SD_ENV_VALUE * Ppad + l] = c.value; sd[SD_ENV_TOP * Ppad + l] = c.top;
  sd[SD_CTRL_FROM * Ppad + l] = c.from; sd[SD_CTRL_TO * Ppad + l] = c.to; sd[SD_CTRL_CUR * Ppad + l] = c.cur;
  sd[SD_CTRL_C * Ppad + l] = c.C; sd[SD_CTRL_FRES * Ppad + l] = c.fres;
  sd[SD_OSC_CRE * Ppad + l] = c.cre; sd[SD_OSC_CIM * Ppad + l] = c.cim; sd[SD_OSC_MRE * Ppad + l] = c.mre;
  sd[SD_OSC_MIM * Ppad + l] = c.mim; sd[SD_OSC_ALIAS * Ppad + l] = c.alias; sd[SD_OSC_INC * Ppad + l] = c.inc;
  sd[SD_VOL * Ppad + l] = c.vol; sd[SD_VOL_TARGET * Ppad + l] = c.target; sd[SD_VOL_FINIT * Ppad + l] = c.finit;
  sd[SD_VOL_K * Ppad + l] = c.k; sd[SD_VOL_Y * Ppad + l] = c.y; sd[SD_GAIN_L * Ppad + l] = c.gL; sd[SD_GAIN_R * Ppad + l] = c.gR;
  si[SI_ENV_STATE * Ppad + l] = c.state; si[SI_ENV_COUNTER * Ppad + l] = c.counter; si[SI_ENV_RELDELAY * Ppad + l] = c.reldelay;
  si[SI_ENV_AHDSTATE * Ppad + l] = c.ahdstate; si[SI_ENV_AHDCOUNTER * Ppad + l] = c.ahdcounter; si[SI_ENV_MINCH * Ppad + l] = c.minch;
  si[SI_ENV_NEXTMINCH * Ppad + l] = c.nextminch; si[SI_FLAGS * Ppad + l] = c.flags;
}

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
          chan_step(c, p, dur, max_filter_inc, 0);
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
      // canHandleExplicitKeyReleaseNow(0) && onKeyReleased(0), gen.crtp.h:581-583, audioelement.h:769-781,802-811
      if (c.state == ENV_PRESSED && c.reldelay == -1 && -c.counter <= 0) {
        if (c.counter + 0 <= 0) c.state = ENV_DONE1; else c.reldelay = 0;
      }
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
        // forgetPastSignals: envelope :694-700, ctrl :2725-2727, volume :1168-1172
        c.counter = 0; c.minch = c.nextminch;
        c.flags = 0;
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
        c.state = ENV_PRESSED; c.counter = 0; c.reldelay = -1; c.value = 0; // onKeyPressed(0) :762-767
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
