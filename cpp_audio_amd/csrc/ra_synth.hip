// K3 — additive sine bank.  One lane = one synth channel (partial) of one stream.
//
// Replaces SimpleAudioOutContext::step's one-shot drain + ImplCRTP::compute (reference
// include/simple_audio_context.h:69-112, include/gen.crtp.h:350-378), the per-sample element chain
// (ra_chan.cuh) and the event side effects of ImplCRTP::onEvent (gen.crtp.h:382-622) incl. setPhase (:128-156).
//
// Default pipeline (two launches):
//   K3a propagate : one workgroup per stream walks the block's analysis frames in order; per hop it stores every
//                   channel's state, advances it by the hop length in closed form (chan_jump) and applies the frame's
//                   commands (note on / off / change).  ~O(frames) work, latency bound, tiny.
//   K3b render    : one workgroup per (stream, hop) reloads its start state and runs the reference's per-sample
//                   recurrences for that hop only.  All hops of all streams are independent, so the grid is
//                   streams x frames workgroups instead of `streams` (256 waves cannot fill 1024 SIMDs; 120 k can).
// ra_config.reference_order = 1 selects the sample-serial single-launch kernel instead (bit-identical to the CPU
// restatement's operation order; used by the exactness tests).
//
// Outputs of a 16-frame tile are transposed through LDS and summed over channels by 32 lanes (one per frame x output),
// so the cross-lane reduction costs ~2 LDS ops per channel-sample instead of a 64-lane butterfly per sample.
#include "ra_chan.cuh"
#include <hip/hip_ext.h>
#include "ra_lane.cuh"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

extern __shared__ __align__(16) unsigned char ra_smem_syn[];

// register budgets (waves per SIMD the compiler must leave room for); overridable for tuning builds
#ifndef RA_PROP_EU
#define RA_PROP_EU 2
#endif
#ifndef RA_REND_EU
#define RA_REND_EU 3
#endif
#if RA_PROP_EU > 0
#define RA_PROP_OCC __attribute__((amdgpu_waves_per_eu(RA_PROP_EU)))
#else
#define RA_PROP_OCC
#endif
#if RA_REND_EU > 0
#define RA_REND_OCC __attribute__((amdgpu_waves_per_eu(RA_REND_EU)))
#else
#define RA_REND_OCC
#endif

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
size_t syn_lds_bytes(int Ppad) { return (size_t)32 * (Ppad + 1) * 8 + (size_t)Ppad * (8 * 3 + 4 + 8) + 64; }

struct SynConst {
  RaEnvParams ep;
  double dur, max_filter_inc;
  uint64_t seed;
};
__device__ inline SynConst syn_const(const RaParams &p, int s) {
  SynConst k;
  k.ep = {p.envA, p.envH, p.envD, p.envR, p.SMinusOne};
  k.dur = (double)p.H;                  // ctrl.setup(stride), lib.cpp:110-111
  k.max_filter_inc = 2. / (double)p.H;  // setMaxFilterIncrement, lib.cpp:106
  k.seed = p.seed + (uint64_t)s;
  return k;
}

// samples [t, seg_end) of stream s: step every active channel, mix, write.  Workgroup-uniform control flow.
template <bool FAST>
__device__ inline void render_range(Chan &c, const SynLds &L, const SynConst &k, const RaParams &p, int s, int l, int64_t t, int64_t seg_end, int64_t t0,
                                    int64_t n, const double *__restrict__ mix, float *__restrict__ out, double *__restrict__ out_d,
                                    float *__restrict__ fb_mono = nullptr) {
  const int Ppad = p.Ppad, stride1 = Ppad + 1;
  const double inv_dur = 1.0 / k.dur;
  while (t < seg_end) {
    const int m = (int)min((int64_t)RA_N_FRAMES_PER_BUFFER, seg_end - t);
    for (int i = 0; i < m; ++i) {
      double v = 0.0;
      if (rt_active(c)) { // ImplCRTP::compute, gen.crtp.h:352-371
        if (FAST) chan_step_fast(c, k.ep, k.dur, inv_dur, k.max_filter_inc);
        else chan_step(c, k.ep, k.dur, k.max_filter_inc, 0);
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
      const bool w = l < 32 && fr < m;
      const int64_t rel = (t - t0) + fr;
      if (w && mix) acc += mix[rel];  // the vocoder compute runs after the synth compute (lib.cpp:1353,1377)
      const double other = __shfl_xor(acc, 16); // the same frame's other output channel
      if (w) {
        if (out) out[((size_t)s * n + rel) * 2 + ch] = (float)acc; // simple_audio_context.h:105-109
        if (out_d) out_d[((size_t)s * n + rel) * 2 + ch] = acc;
        if (fb_mono && ch == 0) fb_mono[rel] = (float)(acc + other); // the post processor's publication, lib.cpp:1262-1272, pubsub.cpp:92
      }
    }
    __syncthreads();
    t += m;
  }
}

// ---- regime-based renderer (hop-parallel path) ------------------------------------------------------------------------
// Between two transitions (envelope stage change, glide end, first step of a note) a channel's step is a handful of
// multiply-adds with per-stretch constants.  Each lane caches those constants; the wave takes the general step only on
// samples where one of its lanes is at a transition (~1 % of samples), which keeps the common path free of the state
// machine's branches, divisions and sincos (213 -> ~50 VALU instructions per sample).
struct Regime {
  double dv;                        // envelope value increment per step
  double base, slope, alpha, beta;  // glide: inc = base + slope * cur ; cur' = alpha * cur + beta  (steady: slope 0, alpha 1, beta 0)
  double rf;                        // hi: running reciprocal of the increment (refined by Newton steps as the glide moves)
  double f_end;                     // the increment a glide clamps to
  int steps_left;                   // envelope steps before its next transition
  int pending;                      // steps taken since the envelope counters were last brought up to date
  bool active, gliding, irregular;  // irregular: the next step must be the general one
  bool hi;                          // the glide passes above fs/8: the anti-aliasing gain follows the increment
  bool kvar;                        // the glide passes below the volume filter's maximum increment: its coefficient follows the increment
};

// envelope half of the regime: ramp increment and distance to the next stage change
__device__ inline void regime_env(const Chan &c, const SynConst &k, Regime &r) {
  r.pending = 0;
  r.active = rt_active(c);
  r.dv = 0; r.steps_left = 0x7fffffff;
  if (!r.active) return;
  r.steps_left = env_regular_steps(c, k.ep);
  switch (c.state) {
    case ENV_PRESSED:
      if (c.ahdstate >= 0) {
        double tmf = c.ahdstate == 0 ? 1.0 : c.ahdstate == 1 ? 0.0 : (double)k.ep.SMinusOne;
        r.dv = tmf / (double)env_max_counter(c, k.ep);
      }
      break;
    case ENV_RELEASED: r.dv = -c.top / (double)max(c.minch, k.ep.R); break;
    default: break;
  }
}

// The render kernel never takes the general step: the first step of a note (no filter state, no increment yet) is taken by the
// propagate kernel before it stores the hop's start state (FL_STEPPED0), envelope stage changes and glide ends have lean forms below,
// and a filter coefficient that the next step would recompute (BaseVolumeAdjusted::step, audioelement.h:1195-1216) is recomputed here.
// `irregular` is left for states the path cannot produce (non-positive increments, an unstarted control): they raise RA_ERR_BOOKKEEPING.
__device__ inline void regime_setup(Chan &c, const SynConst &k, double inv_dur, Regime &r) {
  regime_env(c, k, r);
  r.base = c.inc; r.slope = 0; r.alpha = 1; r.beta = 0; r.gliding = false; r.irregular = false; r.hi = false; r.kvar = false; r.rf = 0; r.f_end = c.inc;
  if (!r.active) return;
  if (!(c.flags & FL_HAS_VOL) || !(c.flags & FL_HAS_FRES)) { r.irregular = true; return; }
  const bool steady = c.fres == c.to;
  const double f_end = c.from + (c.to - c.from) * (k.dur / k.dur);
  const bool clamped = !steady && c.cur == k.dur && c.fres == f_end;
  double lo;
  if (steady || clamped) { r.base = c.fres; r.f_end = c.fres; lo = fabs(c.fres); }
  else {
    if (c.to == c.from || c.from <= 0. || c.to <= 0.) { r.irregular = true; return; } // increments are positive for every event the path produces
    r.gliding = true;
    r.f_end = f_end;
    r.slope = (c.to - c.from) * inv_dur;
    r.base = c.from;
    r.alpha = 1.0 + c.C * r.slope;
    r.beta = c.C * (c.from + 0.5 * r.slope);
    lo = fmin(fabs(c.from + r.slope * c.cur), fabs(f_end));
    r.hi = fmax(c.from, c.to) > 0.25;
    if (r.hi) r.rf = 1.0 / (c.from + r.slope * c.cur);
  }
  if (c.vol != c.target) { // the volume filter's coefficient is constant over the stretch, or follows the glide step by step
    const double want = fmin(k.max_filter_inc, lo);
    if (r.gliding && lo < k.max_filter_inc) r.kvar = true;
    else if (c.finit != want) { c.finit = want; const double a = M_PI * want; c.k = a / (1.0 + a); }
  }
}

// tuning aid: -DRA_ABL=1 leaves out the straight-line steps of a clean span (its set-up stays), 2 the channel sum and output, 4 the
// checked steps (bit mask; instruction-count ablations with rocprofv3 --pmc SQ_INSTS_VALU, scripts/build_variants.sh); 0 = the product
#ifndef RA_ABL
#define RA_ABL 0
#endif
#ifndef RA_TILE_DEPTH
#define RA_TILE_DEPTH 32 // samples per tile: 8, 16 or 32 (the per-tile set-up, the span cutting and the channel sums are amortised over them)
#endif
#ifndef RA_TILE_FLOAT
#define RA_TILE_FLOAT 1  // the LDS tile holds a channel's samples as floats (r03): a 32-sample tile then takes the LDS a 16-sample tile of doubles took
#endif
// A tile entry is one channel's contribution to one frame, |v| <= the channel's volume (a few 1e-2 on the BASELINE shapes); the sum over the
// channels is formed in double and rounded to the float output.  Rounding the ENTRY to float first moves it by 6e-8 of itself — a few 1e-9
// on the sum, below the 1e-8 the path has against the restatement anyway.  With 32-sample tiles of doubles the tile buffer halved the waves
// per SIMD and the kernel got slower (r02); as floats the buffer keeps its size and the set-up of a span is paid half as often.
#if RA_TILE_FLOAT
using tile_t = float;
#else
using tile_t = double;
#endif
constexpr int TD = RA_TILE_DEPTH; // tile depth of the regime renderer
static_assert(TD == 8 || TD == 16 || TD == 32, "tile depth");
constexpr int TROW_MONO = TD + 1; // the LDS tile is lane-major: [Ppad][TROW_MONO] entries, one value per frame (odd row length: conflict-free both ways)
// bytes of one wave group's tile, and — stereo — of the channels' two gains [Ppad] double2 behind it
__host__ __device__ inline size_t ra_tile_bytes(int Ppad, bool mono) { return (size_t)Ppad * TROW_MONO * sizeof(tile_t) + (mono ? 0 : (size_t)Ppad * sizeof(double2)); }

// LEN steps (a whole tile, or half of one) of a lane whose wave has no transition inside them: straight-line code, no state machine.
// GLIDE: some lane of the wave glides (the others run the same code with the identity rotation, alpha = 1, beta = 0);
// HI: some gliding lane passes above fs/8, where the anti-aliasing gain follows the increment.
// The phasor is renormalised once per tile instead of once per step (drift < 1e-15 per tile; every hop restarts from the angle).
#ifndef RA_TILE_UNROLL
#define RA_TILE_UNROLL 16 // (tile depth 8: measured on configs[1]: 8 (straight-line tile) 14.9 ms, 1 (rolled) 15.4 ms, 2: 16.5 ms per step)
#endif
#define RA_PRAGMA(x) _Pragma(#x)
#define RA_UNROLL(n) RA_PRAGMA(unroll n)
// LEN = 0: the number of steps is the runtime argument n (the pieces of a tile cut at a transition), the loop stays rolled.
// Rotation of `mult` while gliding: the first step turns by an arbitrary angle (cd0, sd0); each later one by the span's small mean angle,
// applied as (1, tan) — two multiply-adds instead of four operations.  That turns by exactly the angle and stretches `mult` by 1 / cos,
// i.e. by d^2 / 2 ~ 1e-10 .. 1e-8 per step: `mult` is renormalised at the end of the span like the phasor.
template <int LEN, bool GLIDE, bool HI, bool KVAR, bool MONO>
__device__ inline void clean_tile(Chan &c, Regime &r, double max_filter_inc, double cd0, double sd0, double tn, double al_da, double al_dda, tile_t *mine,
                                  int n) {
  const int len = LEN ? LEN : n;
  double cur = c.cur, mre = c.mre, mim = c.mim, cre = c.cre, cim = c.cim, y = c.y, value = c.value, alias = c.alias, f = c.inc;
  double kk = c.k, fin = c.finit, rr = 0.0, a_used = c.alias;
  double tnr = tn; // the rotation of step j + 1 is that of step j times alpha (the time steps are a geometric sequence): one multiplication per step
  const double target = c.target;
  if (KVAR) rr = 1.0 / (1.0 + M_PI * c.finit);
  // w_i = (alias + i al_da) (value + (i + 1) dv), the product at step i; va holds w_{i-1} before step i
  (void)al_dda; (void)a_used;
  double va = alias * value - (HI ? al_da * value : 0.0); // = w_{-1} = (alias - al_da) * value
  double d1 = HI ? fma(al_da, value, (alias - al_da) * r.dv) + al_da * r.dv : alias * r.dv; // w_0 - w_{-1}
  const double d2 = HI ? 2.0 * (al_da * r.dv) : 0.0;
  // A full tile whose filter coefficient does not follow the increment needs the glide's time `cur` only at its end: the affine step
  // cur' = alpha cur + beta is then composed with itself by doubling (A_2n = A_n^2, B_2n = A_n B_n + B_n) and applied as 16 + 8 + 4 + 2 + 1
  // steps (the increment of the last step wants cur one step before the end), 15 instructions instead of 33 per tile.
  constexpr bool CLOSED = GLIDE && !KVAR && (LEN == 8 || LEN == 16 || LEN == 32);
  auto step = [&](auto first_tag) -> double {
    constexpr bool FIRST = decltype(first_tag)::value;
    if (GLIDE) {
      if (!CLOSED) {
        f = fma(r.slope, cur, r.base);
        cur = fma(r.alpha, cur, r.beta);
      }
      double mr, mi;
      if (FIRST) { mr = fma(mre, cd0, -(mim * sd0)); mi = fma(mre, sd0, mim * cd0); }
      else { mr = fma(-tnr, mim, mre); mi = fma(tnr, mre, mim); tnr *= r.alpha; }
      mre = mr; mim = mi;
      if (KVAR) { // BaseVolumeAdjusted::step, audioelement.h:1195-1216: k = a / (1 + a), a = pi min(max increment, |inc|); 1 / (1 + a) by Newton
        if (r.kvar && y != target) {
          const double w = fmin(max_filter_inc, fabs(f));
          if (w != fin) {
            const double a1 = 1.0 + M_PI * w;
            rr = rr * fma(-a1, rr, 2.0); rr = rr * fma(-a1, rr, 2.0);
            kk = (M_PI * w) * rr; fin = w;
          }
        }
      }
    }
    const double re = fma(cre, mre, -(cim * mim)), im = fma(cre, mim, cim * mre);
    cre = re; cim = im;
    y = fma(kk, target - y, y);
    // anti-aliasing gain times envelope: both move linearly over the span (HI: the gain along the secant of freqAliasingMultiplicator,
    // audioelement.h:466-483, between the span's first and last increment — set up by the caller; else it is constant), so their product
    // is followed by its forward differences: va += d1 (; d1 += d2)
    va += d1;
    if (HI) d1 += d2;
    return y * (cim * va);
  };
  auto put = [&](int i, double v) { mine[i] = (tile_t)v; }; // (stereo: the gains are applied in the channel sum)
  if (RA_ABL & 1) return;
  put(0, step(std::true_type{}));
  if (LEN) {
    RA_UNROLL(RA_TILE_UNROLL)
    for (int i = 1; i < LEN; ++i) put(i, step(std::false_type{}));
  } else {
    for (int i = 1; i < len; ++i) put(i, step(std::false_type{}));
  }
  if (CLOSED) {
    constexpr int LG = LEN == 32 ? 5 : LEN == 16 ? 4 : 3;
    double A[LG], B[LG];
    A[0] = r.alpha; B[0] = r.beta;
#pragma unroll
    for (int q = 1; q < LG; ++q) { A[q] = A[q - 1] * A[q - 1]; B[q] = fma(A[q - 1], B[q - 1], B[q - 1]); }
#pragma unroll
    for (int q = LG - 1; q >= 0; --q) cur = fma(A[q], cur, B[q]); // LEN - 1 steps
    f = fma(r.slope, cur, r.base);
    cur = fma(r.alpha, cur, r.beta);
  }
  const double g = 0.5 * (3.0 - fma(cre, cre, cim * cim));
  c.cre = cre * g; c.cim = cim * g;
  c.y = y; c.vol = y; c.value = fma((double)len, r.dv, value);
  if (GLIDE) {
    const double gm = 0.5 * (3.0 - fma(mre, mre, mim * mim));
    c.mre = mre * gm; c.mim = mim * gm; c.cur = cur; c.alias = HI ? fma((double)(len - 1), al_da, alias) : alias; // the gain the last step used
    if (r.gliding) { c.fres = f; c.inc = f; }
    if (KVAR) { c.k = kk; c.finit = fin; }
  }
}

template <bool MONO, bool PACKED>
__device__ inline void render_range_fast(Chan &c, tile_t *tile, const SynConst &k, const RaParams &p, int s, int l, int64_t t, int64_t seg_end,
                                         int64_t t0, int64_t n, const double *__restrict__ mix, float *__restrict__ out, double *__restrict__ out_d,
                                         int32_t *b_error_flags, float *__restrict__ fb_mono, RaPhaseClock &clk, const int Ppad, const int64_t dB) {
  constexpr bool packed = PACKED;
  // packed (Ppad = 64, one wave): lanes 0..31 carry the sounding channels of the hop that starts at t, lanes 32..63 those of the hop that
  // starts at t + dB and has the same length; the channel sums then stop at 32 lanes and each half writes its own frames.
  const double inv_dur = 1.0 / k.dur;
  const double lim = k.dur - 0.5; // a glide step that finds cur > lim clamps (InterpolatedFreq::step, audioelement.h:2768-2770)
  bool stepped0 = (c.flags & FL_STEPPED0) != 0; // this hop's first step was taken by the propagate kernel: sample 0 only reads the state
  c.flags &= ~FL_STEPPED0;
  Regime r;
  regime_setup(c, k, inv_dur, r);
  clk.mark(21);
  constexpr int ROW = TROW_MONO;
  tile_t *mine = tile + l * ROW;                                                // this lane's row
  // stereo: every channel has its own two gains (StereoPanned, stereo.h:18-27); the lanes that sum a frame read them from LDS, so the tile
  // holds one value per frame and channel as in the mono kernel (half the LDS of a tile of gL v, gR v pairs: three waves per SIMD)
  double2 *gains = reinterpret_cast<double2 *>(tile + (size_t)Ppad * ROW);
  if (!MONO) gains[l] = make_double2(c.gL, c.gR);
  // reduction role of lanes 0..63: frame fr of output ch over a quarter of the channels (MONO: frame fr over an eighth)
  // the 64 reducing lanes: output value o (frame fr of channel ch; entries of a row are laid out [ch][fr]) over one of NPART channel groups
  constexpr int NOUT = TD, NPART = 64 / NOUT;
  const int o = l & (NOUT - 1), fr = o & (TD - 1), part = (l & 63) / NOUT;
  const tile_t *col = tile + (part * (Ppad / NPART)) * ROW + o;
  double gLm = 0.0, gRm = 0.0;
  if (MONO) { const double ang = (0.25 * M_PI) * (0.0 + 1.); gLm = cos(ang); gRm = sin(ang); } // stereo(0), stereo.h:18-27, as at note-on

  // One checked step (tiles in which some lane meets a transition).  An envelope stage change runs the envelope's own state machine
  // (its counters are first brought up to date) and re-derives the envelope half of the regime; a glide that has reached its end clamps
  // (InterpolatedFreq::step, audioelement.h:2768-2785) and the lane turns steady; everything else is the regular recurrence with the
  // exact per-step rotation update pi (inc_i - inc_{i-1}).
  auto checked_step = [&]() -> double {
    const bool env_now = r.steps_left == 0;
    if (env_now) {
      if (r.pending) env_jump(c, k.ep, r.pending);
      env_step(c, k.ep);
      regime_env(c, k, r);
      if (!r.active) return 0.0; // the envelope ended on this step: the element is skipped from here on (gen.crtp.h:356-362)
    }
    if (r.gliding) {
      const bool ended = c.cur > lim;
      const double f = ended ? r.f_end : fma(r.slope, c.cur, r.base);
      c.cur = ended ? k.dur : fma(r.alpha, c.cur, r.beta);
      const double d = M_PI * (f - c.inc), d2 = d * d;
      const double sd = d * fma(d2, fma(d2, 1.0 / 120.0, -1.0 / 6.0), 1.0), cd = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
      const double mr = fma(c.mre, cd, -(c.mim * sd)), mi = fma(c.mre, sd, c.mim * cd);
      c.mre = mr; c.mim = mi; c.inc = f; c.fres = f;
      if (r.hi) { c.alias = alias_mult(f); r.rf = 1.0 / f; }
      if ((r.kvar || ended) && c.y != c.target) {
        const double w = fmin(k.max_filter_inc, fabs(f));
        if (w != c.finit) { c.finit = w; const double a = M_PI * w; c.k = a / (1.0 + a); }
      }
      if (ended) { r.gliding = false; r.hi = false; r.kvar = false; r.base = f; r.slope = 0; r.alpha = 1; r.beta = 0; }
    }
    const double re = fma(c.cre, c.mre, -(c.cim * c.mim));
    const double im = fma(c.cre, c.mim, c.cim * c.mre);
    const double g = 0.5 * (3.0 - fma(re, re, im * im));
    c.cre = re * g; c.cim = im * g;
    c.y = fma(c.k, c.target - c.y, c.y);
    c.vol = c.y;
    if (!env_now) { c.value += r.dv; ++r.pending; --r.steps_left; }
    return c.vol * ((c.alias * c.cim) * c.value);
  };
  if (__any(r.active && r.irregular)) atomicOr(b_error_flags, RA_ERR_BOOKKEEPING); // see regime_setup

  // LEN (compile time; 0: n_rt) consecutive samples of the tile, written to row entries off .. off + LEN - 1, by the straight-line code —
  // if no lane of the wave has a transition inside them; otherwise nothing is produced and the caller cuts the span shorter.
  auto clean_span = [&](auto len_tag, int off, int n_rt) -> bool {
    constexpr int LENC = decltype(len_tag)::value;
    const int LEN = LENC ? LENC : n_rt;
    tile_t *row = mine + off;
    // a span is clean for a lane if no transition can fall inside it: then the loop needs no per-sample checks
    double step0 = 0.0;
    if (r.active && r.gliding) step0 = fma(r.alpha - 1.0, c.cur, r.beta); // cur increment of the span's first step
    bool clean = !r.active || (!stepped0 && r.steps_left >= LEN &&
                               (!r.gliding || (fabs(r.alpha - 1.0) < 1e-2 && c.cur + (LEN - 1) * fabs(step0) * 1.1 <= lim)));
    // Anti-aliasing gain of a glide above fs/8, (1 / inc - 1) / 3 between inc = 1/4 and 1 (freqAliasingMultiplicator, audioelement.h:466-483):
    // over a span it follows the secant between its values at the span's first and last increment (1 / inc from the running reciprocal,
    // two Newton steps each; the function's curvature over a span leaves < 3e-7 of the gain).  A span in which the increment crosses 1/4
    // or 1 is not clean (the checked step evaluates the function).
    double al0 = c.alias, al_da = 0.0;
    const double al_dda = 0.0;
    if (clean && r.active && r.gliding && r.hi) {
      const double f0 = fma(r.slope, c.cur, r.base), df = r.slope * step0, fl = fma((double)(LEN - 1), df, f0); // first and last increment
      const double flo = f0 < fl ? f0 : fl, fhi = f0 < fl ? fl : f0;
      double rf = r.rf * fma(-f0, r.rf, 2.0);
      rf = rf * fma(-f0, rf, 2.0);
      r.rf = rf;
      if (fhi <= 0.25) al0 = 1.0;
      else if (flo >= 1.0) al0 = 0.0;
      else if (flo >= 0.25 && fhi <= 1.0) {
        double rl = rf * fma(-fl, rf, 2.0);
        rl = rl * fma(-fl, rl, 2.0);
        al0 = fma(rf, 1. / 3., -1. / 3.);
        const double inv = LENC == TD ? 1.0 / (TD - 1) : (LEN > 1 ? 1.0 / (double)(LEN - 1) : 0.0);
        al_da = ((rl - rf) * (1. / 3.)) * inv;
      } else clean = false;
    }
    if (!__all(clean)) return false;
    const bool any_glide = __any(r.active && r.gliding);
    if (r.active) {
      c.alias = al0; // (its own value for lanes whose gain does not move)
      if (!any_glide) { clk.tally<0>(); clean_tile<LENC, false, false, false, MONO>(c, r, k.max_filter_inc, 1.0, 0.0, 0.0, 0.0, 0.0, row, LEN); }
      else {
        // Rotation of `mult` per step while gliding: pi (inc_i - inc_{i-1}).  The span's first step turns by d0 = pi (inc_0 - inc),
        // whatever the oscillator's increment currently is (0 right after a note change); the later steps turn by
        // pi slope (cur_i - cur_{i-1}), a geometric sequence with ratio alpha (|alpha - 1| ~ 1e-5): their mean serves all of them,
        // as its tangent (clean_tile).
        double cd0 = 1.0, sd0 = 0.0, tn = 0.0;
        if (r.gliding) {
          const double d0 = M_PI * (fma(r.slope, c.cur, r.base) - c.inc);
          const double dA = (M_PI * r.slope) * step0; // the second step's angle; every later one is alpha times its predecessor's (clean_tile)
          double d2 = d0 * d0;
          sd0 = d0 * fma(d2, fma(d2, 1.0 / 120.0, -1.0 / 6.0), 1.0);
          cd0 = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
          d2 = dA * dA;
          tn = dA * fma(d2, 1.0 / 3.0, 1.0); // tan: |dA| ~ 1e-5 .. 1e-4 (the cubic term is kept for the first angle; it then shrinks or grows by alpha^3 per step: < 1e-13)
        }
        if (__any(r.active && r.kvar)) clk.tally<3>(); else if (__any(r.active && r.gliding && r.hi)) clk.tally<2>(); else clk.tally<1>();
        if (__any(r.active && r.kvar)) clean_tile<LENC, true, true, true, MONO>(c, r, k.max_filter_inc, cd0, sd0, tn, al_da, al_dda, row, LEN);
        else if (__any(r.active && r.gliding && r.hi)) clean_tile<LENC, true, true, false, MONO>(c, r, k.max_filter_inc, cd0, sd0, tn, al_da, al_dda, row, LEN);
        else clean_tile<LENC, true, false, false, MONO>(c, r, k.max_filter_inc, cd0, sd0, tn, 0.0, 0.0, row, LEN);
      }
      r.pending += LEN; r.steps_left -= LEN;
    } else {
      for (int i = 0; i < LEN; ++i) row[i] = (tile_t)0;
    }
    clk.lap<0>();
    return true;
  };
  // how many of the next `remaining` samples this lane can take before its next transition (an estimate that errs on the short side for
  // the glide's end: the exact test is clean_span's)
  auto safe_len = [&](int remaining) -> int {
    if (!r.active) return remaining;
    if (stepped0) return 0;
    int n = min(remaining, r.steps_left);
    if (r.gliding) {
      if (!(fabs(r.alpha - 1.0) < 1e-2)) return 0;
      const float step = (float)(fabs(fma(r.alpha - 1.0, c.cur, r.beta)) * 1.1), room = (float)(lim - c.cur);
      if (!(room >= 0.f)) return 0;
      if (step > 0.f) { const float q = 0.999f * room / step; n = q < 1.0e6f ? min(n, (int)q + 1) : n; }
    }
    return n;
  };
  auto wave_min = [&](int v) -> int {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
  };

  while (t < seg_end) {
    const int m = (int)min((int64_t)TD, seg_end - t);
    // The whole tile in one unrolled go if no lane has a transition in it (~9 of 10 tiles).  Otherwise it is cut at the transitions: clean
    // pieces of whatever length the wave allows (rolled loop), and at a sample where some lane changes regime one checked step for the
    // wave — an envelope stage change, the end of a glide, a note's first sample (already taken by the propagate kernel).
    int pos = (m == TD && clean_span(std::integral_constant<int, TD>{}, 0, TD)) ? TD : 0;
    while (pos < m) {
      const int nn = wave_min(safe_len(m - pos));
      if (nn >= 1 && clean_span(std::integral_constant<int, 0>{}, pos, nn)) { pos += nn; continue; }
      double v = 0.0;
      if (r.active && !(RA_ABL & 4)) {
        if (stepped0) v = c.vol * ((c.alias * c.cim) * c.value); // already stepped (propagate kernel): StereoPanned::imag of the state as loaded
        else v = checked_step();
      }
      stepped0 = false;
      mine[pos] = (tile_t)v;
      ++pos;
      clk.lap<1>();
    }
    // one wave: its LDS operations execute in order in hardware; the fence and the wave barrier cost no instruction and keep the compiler
    // from moving a lane's `mine[pos] = v` below the other lanes' column reads
    if (Ppad > 64) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    if (l < 64 && !(RA_ABL & 2)) { // NOUT outputs x NPART channel groups
      double acc = 0.0, accR = 0.0; // mono: the sum; stereo: left and right
      if (fr < m) {
        const tile_t *q = col;
        constexpr int NJ = 64 / NPART; // channels of a group per 64-channel block
        const double2 *g = gains + part * (Ppad / NPART);
        for (int blk = 0; blk < (Ppad >> 6); ++blk, q += NJ * ROW, g += NJ) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (MONO) acc += (double)q[j * ROW];
            else { const double v = (double)q[j * ROW]; const double2 gg = g[j]; acc += gg.x * v; accR += gg.y * v; }
          }
        }
      }
      if (!packed) acc += lane_xor32_low(acc);
      if (NPART >= 4) acc += lane_xor16_even(acc);
      if (NPART >= 8) acc += lane_ror8(acc);
      if (!MONO) {
        if (!packed) accR += lane_xor32_low(accR);
        if (NPART >= 4) accR += lane_xor16_even(accR);
        if (NPART >= 8) accR += lane_ror8(accR);
      }
      const int lo_lane = packed ? (l & 31) : l; // packed: lanes 32.. hold the second hop's frames (even rows of 16: where the swaps are valid)
      const int64_t t_mine = (packed && l >= 32) ? t + dB : t;
      if (lo_lane < NOUT && fr < m) {
        const int64_t rel = (t_mine - t0) + fr;
        double oL = MONO ? gLm * acc : acc, oR = MONO ? gRm * acc : accR;
        if (mix) { oL += mix[rel]; oR += mix[rel]; }
        if (out) { // one 8-byte store per frame when the caller's buffer allows it (two 4-byte stores make partial-line writes)
          float *op = out + ((size_t)s * n + rel) * 2;
          if ((reinterpret_cast<uintptr_t>(out) & 7) == 0) *reinterpret_cast<float2 *>(op) = make_float2((float)oL, (float)oR);
          else { op[0] = (float)oL; op[1] = (float)oR; }
        }
        if (out_d) { out_d[((size_t)s * n + rel) * 2] = oL; out_d[((size_t)s * n + rel) * 2 + 1] = oR; }
        if (fb_mono) fb_mono[rel] = (float)(oL + oR); // what the post processor publishes to the analysis (lib.cpp:1262-1272, pubsub.cpp:92)
      }
    }
    if (Ppad > 64) __syncthreads();
    clk.lap<2>();
    t += m;
  }
  clk.flush(24);
}

// the one-shots of one analysis frame, in the order the reference queued them.  Workgroup-uniform control flow.
// PHASE: the oscillator is carried as its angle theta (propagate kernel) instead of the phasor (cre, cim).
template <bool PHASE>
__device__ inline void apply_frame_events(Chan &c, const RaCmd &cmd, int n_on, const SynLds &L, const SynConst &k, const RaParams &p, const RaBuffers &b, int l) {
  const int ctype = cmd.type & 0xff;
  if (ctype == RA_CMD_OFF) {
    // canHandleExplicitKeyReleaseNow(0) && onKeyReleased(0), gen.crtp.h:581-583
    if (env_can_release_now(c, 0)) env_on_key_released(c, 0);
  } else if (ctype == RA_CMD_CHANGE) { // gen.crtp.h:600-616
    float angleIncrements = 2 * cmd.freq / p.sample_rate;
    float angleIncrementsMult = angleIncrements * 1.f;
    c.target = (double)0.1f * (double)cmd.vel; c.flags |= FL_HAS_TARGET;
    chan_set_angle_increments(c, (double)angleIncrementsMult, 0);
  } else if (ctype == RA_CMD_ON) {
    L.ev_lane[cmd.type >> 8] = l; // rank among the frame's note-ons, in the reference's one-shot order
  }
  if (n_on == 0) return; // nothing reads the exchange arrays (workgroup-uniform)
  L.x_inc[l] = angle_increments(c, 0);
  L.x_cre[l] = PHASE ? c.theta : c.cre;
  L.x_cim[l] = c.cim;
  L.x_act[l] = rt_active(c) ? 1 : 0;
  __syncthreads();
  // scratch for the broadcast of one note-on's increment and the lowest matching lane (the tile is idle between hops)
  double *bc_ai = L.tile;
  int *bc_src = reinterpret_cast<int *>(L.tile + 1);
  for (int e = 0; e < n_on; ++e) {
    const int el = L.ev_lane[e];
    double ai = 0.0;
    if (el == l) {
      // -- onEvent(NoteOn), gen.crtp.h:389-449 --
      if (c.state != ENV_DONE2) atomicOr(b.error_flags, RA_ERR_BOOKKEEPING); // tryAcquire must succeed here
      c.state = ENV_SOON;
      chan_forget_past_signals(c);
      c.target = (double)0.1f * (double)cmd.vel; c.flags |= FL_HAS_TARGET;  // :425
      // ResynthElementInitializer, lib.cpp:96-118 (setAHDSR / max filter increment are uniform parameters here)
      c.cur = 0; // ctrl.setup: no f_result yet
      {
        float u = hash_uniform_pm1(k.seed, (uint64_t)cmd.noteid, 1);
        double pan = (double)(p.stereo_spread * u);
        pan = fmin(pan, 1.); pan = fmax(pan, -1.);
        double ang = (0.25 * M_PI) * (pan + 1.);
        c.gL = cos(ang); c.gR = sin(ang);   // stereo(), stereo.h:18-27
      }
      // setupAudioElement, gen.sine.h:9-21
      float inc_f = 2 * (cmd.freq * 1.f) / p.sample_rate;
      chan_set_angle_increments(c, (double)inc_f, 0);
      ai = angle_increments(c, 0);
      *bc_ai = ai;
      *bc_src = 0x7fffffff;
    }
    __syncthreads();
    // -- the one-shot, gen.crtp.h:455-527: setPhase (:128-156): the first RT-active channel (in channel order) whose increment
    //    equals this one's donates its phase; every lane tests itself, the lowest match wins
    {
      const bool match = l != el && l < p.P && L.x_act[l] && L.x_inc[l] == *bc_ai;
      if (match) atomicMin(bc_src, l);
    }
    __syncthreads();
    if (el == l) {
      const int src = *bc_src;
      if (src != 0x7fffffff) { // synchronizeAngles :2397-2399
        if (PHASE) c.theta = L.x_cre[src]; else { c.cre = L.x_cre[src]; c.cim = L.x_cim[src]; }
      } else {
        float a = (p.phase_policy == RA_PHASE_HASH) ? hash_uniform_pm1(k.seed, (uint64_t)cmd.noteid, 0) : 0.f;
        if (PHASE) c.theta = (double)a;      // setStartAngle :2401-2403: cur = polar(pi a)
        else {
          double sn, cs;
          sincos(M_PI * (double)a, &sn, &cs);
          c.cre = cs; c.cim = sn;
        }
      }
      env_on_key_pressed(c, 0);
      L.x_cre[l] = PHASE ? c.theta : c.cre; L.x_cim[l] = c.cim; L.x_act[l] = 1; L.x_inc[l] = ai;
    }
    __syncthreads();
  }
}

// ---- reference-order kernel: one workgroup per stream, samples strictly in order ----------------------------------
__global__ void __launch_bounds__(RA_MAX_CHANNELS) ra_synth_serial_kernel(RaParams p, RaBuffers b, RaSpan sp, float *__restrict__ out, double *__restrict__ out_d) {
  const int s = blockIdx.x, l = threadIdx.x, Ppad = p.Ppad;
  SynLds L = carve(ra_smem_syn, Ppad);
  double *sd = b.sd + (size_t)s * SD_COUNT * Ppad;
  int *si = b.si + (size_t)s * SI_COUNT * Ppad;
  Chan c;
  load_chan(c, sd, si, Ppad, l);
  const SynConst k = syn_const(p, s);
  const double *mix = (p.voc_enabled && b.voc_mix) ? b.voc_mix + (size_t)s * sp.n : nullptr;
  int64_t t = sp.ct0;
  for (int64_t i = 0; i <= sp.nfb; ++i) {
    const int64_t seg_end = (i < sp.nfb) ? ra_fire_sample(p.N, p.H, sp.f_lo + sp.fb0 + i) : sp.ct1;
    render_range<false>(c, L, k, p, s, l, t, seg_end, sp.t0, sp.n, mix, out, out_d, (b.fb_block && !p.post_limit) ? b.fb_block + (size_t)s * sp.n : nullptr);
    t = seg_end;
    if (i == sp.nfb) break;
    const RaCmd cmd = b.cmd[((size_t)s * b.frames_cap + sp.fb0 + i) * Ppad + l];
    apply_frame_events<false>(c, cmd, b.cmd_non[(size_t)s * b.frames_cap + sp.fb0 + i], L, k, p, b, l);
  }
  store_chan(c, sd, si, Ppad, l);
}

// ---- K3a: state at the start of every segment (segment i of a chunk = samples between its frames i-1 and i) -----------
__global__ void __launch_bounds__(RA_MAX_CHANNELS) RA_PROP_OCC ra_synth_propagate_kernel(RaParams p, RaBuffers b, RaSpan sp) {
  __builtin_amdgcn_s_setprio(3); // a dependency chain a few hundred waves wide: let it win issue arbitration against the throughput kernels it runs beside
  const int s = blockIdx.x, l = threadIdx.x, Ppad = p.Ppad;
  SynLds L = carve(ra_smem_syn, Ppad);
  double *sd = b.sd + (size_t)s * SD_COUNT * Ppad;
  int *si = b.si + (size_t)s * SI_COUNT * Ppad;
  Chan c;
  load_chan(c, sd, si, Ppad, l);
  const SynConst k = syn_const(p, s);
  const bool hop_mono = ra_hop_mono(p); // the render kernel that follows applies the centre gains after the channel sum: no gains handed over
  RaPhaseClock clk(b.dbg_clk, l == 0);
  int64_t t = sp.ct0;
  for (int64_t i = 0; i <= sp.nfb; ++i) {
    const int64_t seg_end = (i < sp.nfb) ? ra_fire_sample(p.N, p.H, sp.f_lo + sp.fb0 + i) : sp.ct1;
    const size_t seg = (size_t)s * b.seg_cap + sp.seg0 + i;
    // The first step of a note has no closed form (no filter state, no increment yet): it is taken here, before the hop's start state
    // is stored, so that the render kernel never needs the general step; the render outputs that sample from the stored state.
    int64_t len = seg_end - t;
    if (p.render_mode == 0 && len > 0 && rt_active(c) && (!(c.flags & FL_HAS_VOL) || !(c.flags & FL_HAS_FRES))) {
      chan_step_phase(c, k.ep, k.dur, 1.0 / k.dur, k.max_filter_inc);
      c.flags |= FL_STEPPED0;
      --len;
    }
    // the hop's start state for the render kernel: a channel that does not sound hands over its envelope state word only (the render
    // never looks at the rest: half of the 12.8 KB per hop and stream on the BASELINE shapes)
    if (rt_active(c)) store_hop(c, b.seg_sd + seg * SD_HOP_COUNT * Ppad, b.seg_si + seg * SI_HOP_COUNT * Ppad, Ppad, l, hop_mono);
    else (b.seg_si + seg * SI_HOP_COUNT * Ppad)[SI_ENV_STATE * Ppad + l] = c.state;
    c.flags &= ~FL_STEPPED0;
    clk.lap<0>();
    chan_jump_phase(c, k.ep, k.dur, k.max_filter_inc, len);
    clk.lap<1>();
    t = seg_end;
    if (i == sp.nfb) break;
    const RaCmd cmd = b.cmd[((size_t)s * b.frames_cap + sp.fb0 + i) * Ppad + l];
    const int n_on = b.cmd_non[(size_t)s * b.frames_cap + sp.fb0 + i];
    if (clk.dst && cmd.type + n_on == -12345) clk.acc[3] += 1; // tuning aid: makes the lap below wait for the loads
    clk.lap<2>();
    apply_frame_events<true>(c, cmd, n_on, L, k, p, b, l);
    clk.lap<3>();
  }
  store_chan(c, sd, si, Ppad, l);
  clk.flush(40);
}

// ---- K3b: render one segment ----------------------------------------------------------------------------------------
template <bool MONO>
__device__ inline void render_diag_segment(const RaParams &p, const RaBuffers &b, const RaSpan &sp, float *__restrict__ out, double *__restrict__ out_d) {
  const int i = blockIdx.x, s = blockIdx.y, l = threadIdx.x, Ppad = p.Ppad;
  if (i > sp.nfb) return;
  const int64_t seg_begin = (i == 0) ? sp.ct0 : ra_fire_sample(p.N, p.H, sp.f_lo + sp.fb0 + i - 1);
  const int64_t seg_end = (i < sp.nfb) ? ra_fire_sample(p.N, p.H, sp.f_lo + sp.fb0 + i) : sp.ct1;
  if (seg_end <= seg_begin) return;
  const size_t seg = (size_t)s * b.seg_cap + sp.seg0 + i;
  Chan c;
  {
    const int stv = ra_hop_state((b.seg_si + seg * SI_HOP_COUNT * Ppad)[SI_ENV_STATE * Ppad + l]);
    if (stv != ENV_SOON && stv != ENV_DONE2) { load_hop(c, b.seg_sd + seg * SD_HOP_COUNT * Ppad, b.seg_si + seg * SI_HOP_COUNT * Ppad, Ppad, l, false); chan_from_phase(c); }
    else { chan_init(c); c.state = stv; }
  }
  const SynConst k = syn_const(p, s);
  const double *mix = (p.voc_enabled && b.voc_mix) ? b.voc_mix + (size_t)s * sp.n : nullptr;
  SynLds L = carve(ra_smem_syn, Ppad);
  if (p.render_mode == 2) render_range<false>(c, L, k, p, s, l, seg_begin, seg_end, sp.t0, sp.n, mix, out, out_d);
  else render_range<true>(c, L, k, p, s, l, seg_begin, seg_end, sp.t0, sp.n, mix, out, out_d);
}

// A workgroup renders two consecutive hops.  Channels are handed out first-free, so a stream that plays ~30 partials leaves half of a
// 64-lane wave (and every further wave of a workgroup) idle: measured on the BASELINE shapes, 54 % of the hops of configs[1] and 97 % of
// those of configs[3] (128 channels) have at most 32 sounding channels.  When both hops of the pair do and have the same length, their
// sounding channels are compacted into the two halves of ONE wave (the workgroup's other waves leave) and the pair costs what one hop
// did; otherwise the hops are rendered one after the other by the whole workgroup as before.
// (r03: packing by the SUM of the two counts — an uneven split of the wave, every summing lane keeping one masked sum per hop — reaches 52 %
// instead of 48 % of configs[1]'s hops and costs every packed tile its selects: 13.6 against 12.9 ms per step on one box; not kept.)
template <bool MONO>
__device__ inline void render_pair(const RaParams &p, const RaBuffers &b, const RaSpan &sp, float *__restrict__ out, double *__restrict__ out_d) {
  // blockDim.y > 1 (RA_RENDER_GROUP, single-wave streams only): wave y of the workgroup renders its own pair, nothing is shared between the
  // waves (a tuning aid: grouping makes the renderer release wave slots on all SIMDs of a CU at once; measured slower, not the default).
  const int wy = __builtin_amdgcn_readfirstlane((int)threadIdx.y); // wave-uniform (blockDim.x is a multiple of 64): keeps the hop's loops scalar
  // One workgroup per hop, as before.  Hop 0 of a span is a partial one (it starts where the span does) and goes alone; the pairs are hops
  // (1, 2), (3, 4), ...: the workgroup of the first hop of a pair renders both if they can be packed and the workgroup of the second one,
  // which takes the same decision from the same data, leaves; otherwise each renders its own hop.
  const int ih = blockIdx.x * blockDim.y + wy, s = blockIdx.y, l = threadIdx.x, Ppad = p.Ppad;
  if (ih > sp.nfb) return;
  const bool second = ih > 0 && !(ih & 1);            // the second hop of a pair
  const int i0 = second ? ih - 1 : ih;                // the pair's first hop (or the lone hop)
  const bool two = i0 > 0 && i0 + 1 <= sp.nfb;
  auto hop_begin = [&](int i) -> int64_t { return (i == 0) ? sp.ct0 : ra_fire_sample(p.N, p.H, sp.f_lo + sp.fb0 + i - 1); };
  auto hop_end = [&](int i) -> int64_t { return (i < sp.nfb) ? ra_fire_sample(p.N, p.H, sp.f_lo + sp.fb0 + i) : sp.ct1; };
  const int64_t beg0 = hop_begin(i0), end0 = hop_end(i0), beg1 = two ? hop_begin(i0 + 1) : 0, end1 = two ? hop_end(i0 + 1) : 0;
  const size_t seg0 = (size_t)s * b.seg_cap + sp.seg0 + i0;
  RaPhaseClock clk(b.dbg_clk, l == 0);
  RaClockStamp stamp(b.dbg_stamp, l == 0);
  const unsigned long long wg_t0 = clk.t;
  const SynConst k = syn_const(p, s);
  const double *mix = (p.voc_enabled && b.voc_mix) ? b.voc_mix + (size_t)s * sp.n : nullptr;
  tile_t *tile = reinterpret_cast<tile_t *>(ra_smem_syn + (size_t)wy * ra_tile_bytes(Ppad, MONO));
  float *fb_mono = (b.fb_block && !p.post_limit) ? b.fb_block + (size_t)s * sp.n : nullptr; // with post "limit" the limiter publishes

  // sounding channels of both hops: every wave looks at all of them (Ppad / 64 state words per lane and hop), so the decision needs no barrier
  bool pack = end0 > beg0 && end1 > beg1 && (end0 - beg0) == (end1 - beg1) && p.render_pack && !(RA_ABL & 8);
  int src = -1; // packed: the channel this lane renders (lanes 0..31: first hop, 32..63: second), -1: none
  if (pack) {
    const int lw = l & 63;
    int cnt[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int32_t *st = b.seg_si + (seg0 + h) * SI_HOP_COUNT * Ppad + SI_ENV_STATE * Ppad;
      for (int blk = 0; blk < (Ppad >> 6); ++blk) {
        const int stv = ra_hop_state(st[blk * 64 + lw]);
        const bool act = stv != ENV_SOON && stv != ENV_DONE2; // rt_active
        const unsigned long long m = __ballot(act);
        const int rank = cnt[h] + __popcll(m & ((1ull << lw) - 1ull));
        // the lane that will render this channel is (h * 32 + rank): hand the channel index over through the wave's own tile
        if (act && rank < 32 && l < 64) reinterpret_cast<int *>(tile)[h * 32 + rank] = blk * 64 + lw;
        cnt[h] += __popcll(m);
      }
    }
    pack = cnt[0] <= 32 && cnt[1] <= 32;
    if (pack) {
      if (l >= 64) return; // the other waves of the workgroup have nothing to do (no barrier follows on this path)
      const int h = l >> 5;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if ((l & 31) < cnt[h]) src = reinterpret_cast<int *>(tile)[l]; // (one wave: its LDS operations execute in order; the fence pins the compiler's order)
    }
  }
  if (pack && second) return; // rendered by the workgroup of the pair's first hop
  if (clk.dst) atomicAdd(clk.dst + 64 + (pack ? 1 : 0), 1ull); // hops rendered alone / pairs rendered packed
  Chan c;
  if (pack) {
    chan_init(c);
    if (src >= 0) {
      const size_t seg = seg0 + (l >> 5);
      load_hop(c, b.seg_sd + seg * SD_HOP_COUNT * Ppad, b.seg_si + seg * SI_HOP_COUNT * Ppad, Ppad, src, MONO);
      chan_from_phase(c);
    }
    clk.mark(20);
    render_range_fast<MONO, true>(c, tile, k, p, s, l, beg0, end0, sp.t0, sp.n, mix, out, out_d, b.error_flags, fb_mono, clk, 64, beg1 - beg0);
  } else {
    const int64_t bh = second ? beg1 : beg0, eh = second ? end1 : end0;
    if (eh <= bh) return;
    const size_t seg = seg0 + (second ? 1 : 0);
    { // (a silent channel's hop state is its envelope state word, see the propagate kernel)
      const int stv = ra_hop_state((b.seg_si + seg * SI_HOP_COUNT * Ppad)[SI_ENV_STATE * Ppad + l]);
      if (stv != ENV_SOON && stv != ENV_DONE2) { load_hop(c, b.seg_sd + seg * SD_HOP_COUNT * Ppad, b.seg_si + seg * SI_HOP_COUNT * Ppad, Ppad, l, MONO); chan_from_phase(c); }
      else { chan_init(c); c.state = stv; }
    }
    clk.mark(20);
    render_range_fast<MONO, false>(c, tile, k, p, s, l, bh, eh, sp.t0, sp.n, mix, out, out_d, b.error_flags, fb_mono, clk, Ppad, 0);
  }
  stamp.finish(4);
  if (clk.dst) { // workgroup lifetime: maximum and a histogram in steps of 200 k ticks
    const unsigned long long d = __builtin_readcyclecounter() - wg_t0;
    atomicMax(clk.dst + 39, d);
    atomicAdd(clk.dst + 32 + (int)min(d / 200000ull, 6ull), 1ull);
  }
}
__global__ void __launch_bounds__(RA_MAX_CHANNELS) RA_REND_OCC ra_synth_render_kernel(RaParams p, RaBuffers b, RaSpan sp, float *__restrict__ out, double *__restrict__ out_d) {
  render_pair<false>(p, b, sp, out, out_d);
}
// stereo_spread 0 (every channel panned to the centre): the two equal-for-all gains are applied after the channel sum
__global__ void __launch_bounds__(RA_MAX_CHANNELS) RA_REND_OCC ra_synth_render_mono_kernel(RaParams p, RaBuffers b, RaSpan sp, float *__restrict__ out, double *__restrict__ out_d) {
  render_pair<true>(p, b, sp, out, out_d);
}
// tuning aid (ra_config.reference_order 2 / 3): hop-parallel like the product kernel, but every sample takes the general step
__global__ void __launch_bounds__(RA_MAX_CHANNELS) ra_synth_render_diag_kernel(RaParams p, RaBuffers b, RaSpan sp, float *__restrict__ out, double *__restrict__ out_d) {
  render_diag_segment<false>(p, b, sp, out, out_d);
}

} // namespace

void ra_launch_synth_serial(const RaParams &p, const RaBuffers &b, const RaSpan &sp, float *d_out, double *d_out_d, hipStream_t st) {
  if (sp.ct1 <= sp.ct0) return;
  hipLaunchKernelGGL(ra_synth_serial_kernel, dim3((unsigned)p.n_streams), dim3((unsigned)p.Ppad), syn_lds_bytes(p.Ppad), st, p, b, sp, d_out, d_out_d);
}
void ra_launch_synth_propagate(const RaParams &p, const RaBuffers &b, const RaSpan &sp, hipStream_t st) {
  if (sp.ct1 <= sp.ct0) return;
  hipLaunchKernelGGL(ra_synth_propagate_kernel, dim3((unsigned)p.n_streams), dim3((unsigned)p.Ppad), syn_lds_bytes(p.Ppad), st, p, b, sp);
}
void ra_launch_synth_render(const RaParams &p, const RaBuffers &b, const RaSpan &sp, float *d_out, double *d_out_d, hipStream_t st) {
  if (sp.ct1 <= sp.ct0) return;
  if (p.render_mode) {
    const dim3 grid((unsigned)(sp.nfb + 1), (unsigned)p.n_streams), block((unsigned)p.Ppad);
    hipLaunchKernelGGL(ra_synth_render_diag_kernel, grid, block, syn_lds_bytes(p.Ppad), st, p, b, sp, d_out, d_out_d);
    return;
  }
  // hops per workgroup (single-wave streams).  The chip starts ~16 M workgroups per second: with packed pairs half of the one-wave workgroups
  // leave at once and the 17 k workgroups of a chunk took longer to start than to run.  Two hops per workgroup, measured on configs[1] with
  // packing: render alone 9.2 -> 8.3 ms, step 14.9 -> 13.7 ms; four: 16.3 ms (a four-wave workgroup wants a slot on every SIMD at once).
  static const int group_env = std::getenv("RA_RENDER_GROUP") ? std::atoi(std::getenv("RA_RENDER_GROUP")) : 0;
  const int group_want = group_env > 0 ? group_env : 2;
  const unsigned group = (p.Ppad == 64) ? (unsigned)std::min(std::max(group_want, 1), RA_MAX_CHANNELS / 64) : 1u; // hops per workgroup
  const dim3 grid((unsigned)((sp.nfb + 1 + group - 1) / group), (unsigned)p.n_streams), block((unsigned)p.Ppad, group);
  const size_t lds = (size_t)group * ra_tile_bytes(p.Ppad, ra_hop_mono(p)); // the tiles (+ the gains)
  hipEvent_t ea = nullptr, eb = nullptr; // profiling: the launch stamps its own begin and end (RaExtTiming)
  if (ra_ext_timing) { ea = ra_ext_timing->a; eb = ra_ext_timing->b; ra_ext_timing->used = true; }
  if (ra_hop_mono(p)) hipExtLaunchKernelGGL(ra_synth_render_mono_kernel, grid, block, (uint32_t)lds, st, ea, eb, 0, p, b, sp, d_out, d_out_d);
  else hipExtLaunchKernelGGL(ra_synth_render_kernel, grid, block, (uint32_t)lds, st, ea, eb, 0, p, b, sp, d_out, d_out_d);
}
