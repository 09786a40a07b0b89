// K9 — gen.voice: voice::Impl_<..., MODE, ...> (reference include/gen.voice.h:1087-1190) = ImplCRTP::compute / onEvent
// (include/gen.crtp.h:350-622) over VolumeAdjusted<SoundEngine<MODE, 2, ...>> (include/soundengine.h:240-1127), MODE = SWEEP, BIRDS, ROBOTS,
// the engine's ramps being
//   StereoPanned<VolumeAdjusted<Enveloped<FreqCtrl_<Mix<LoudnessVolumeAdjusted<SineOscillatorAlgo<double>>>, LogRamp<double>>, AHDSREnvelope>>>.
//
// What a ramp is, once the host has run the note's Markov chain and the orchestration's schedule (ra_voice_host.hpp): a JOB — a LogRamp
// spec, a start frame, an envelope given by three integers — whose samples depend on nothing but the job:
//   * playNextSpec (soundengine.h:776-803) makes the ramp's state afresh: the control is copied from the spec, forgetPastSignals, start
//     angle 0 (DefaultStartPhase::Zero), an AHDSR{xfade_len, 0, 0, xfade_len, 1} whose minimum change duration nobody sets;
//   * the ramp's own VolumeAdjusted and the element's outer one are fed a constant target from their first step on: y + k (target - y)
//     stays the target exactly, whatever k — they are constants (spec volume; Element::baseVolume * velocity);
//   * the envelope's value is a function of the step index j: j / A up to 1 while pressed (T steps), top (1 - m / R) after the release;
//   * the oscillator's phasor after step j is polar(pi * (inc_1 + ... + inc_j)) up to its renormalisation: the phase is a running sum;
//   * what does carry state from sample to sample: LogRamp's warped time `cur` (audioelement.h:2522-2542), that phase sum, and the
//     loudness volume's one-pole, whose coefficient follows the increment (LoudnessVolumeAdjusted, :1275-1317 over :1195-1216).
// So a block runs as two kernels:
//   pre-pass  one lane per job, samples in order, only that three-number recurrence (no oscillator, no envelope, no output): it leaves
//             the state at every tile boundary the job crosses (tile = 256 frames) and at the block's end (carried to the next block);
//   render    one wavefront per group of (stream, tile) units; a lane = one SLOT of a unit — the host packs a stream's jobs into
//             g = 2^k slots such that the jobs of a slot never overlap — steps the slot's jobs through the tile from the stored states with
//             the very same recurrence code, adds oscillator, envelope and gains, and the wave sums the slots of each unit in slot order:
//             every output sample is written once, by a fixed-order sum (bit-reproducible; no atomics).
// Reference behaviour kept as it is: Mix::angleIncrements() never exceeds 0 (audioelement.h:1856-1865) — it only ever reaches the two
// constant volume filters; a note-off silences the element at once (soundengine.h:345-353) — the host cuts the note's jobs there;
// LogRamp's constant is 1.f when from == to (:2570).
#include "ra_internal.h"
#include "ra_lane.cuh"
#include <algorithm>
#include <cstdlib>

extern __shared__ __align__(16) unsigned char ra_smem_voice[];

namespace {

constexpr int TS = 8; // samples a lane parks in LDS before the wave sums the slots

struct Ramp { // the registers of a lane while it steps a job
  double cur, from, to, dur, inv_dur, l2r;    // LogRamp; l2r = log2(to / from) (PROPORTIONAL_VALUE_DERIVATIVE only)
  double al, be, be_o;                        // its time step as one multiply-add, cur' = al cur + be; when the ramp turns: al' = 2 - al, be' = be_o
  double theta, y;                            // phase sum (units of pi, kept in [0, 2)); the loudness volume
  float lo_f, hi_f, inv_w, lo_e, hi_e;        // the loudness table interval the ramp is in: its two frequencies, 1 / width, its two levels
  int interp, loud_k, swapped;
};

// loudness::equal_loudness_volume_from_freq, include/loudness.h:246-274 over :157-186 (float arithmetic).  The reference finds the table
// interval by bisection on every call; a ramp moves through the 29 frequencies in order, so the interval it is in — freqs[k] <= freq <
// freqs[k + 1], k clamped to 0 .. 27 — is kept in registers and walked when the frequency leaves it (28 times per sweep, not per sample):
// the same interval, the same interpolation ratio * elv[k + 1] + (1 - ratio) * elv[k] (at the interval's ends and outside the table the
// ratio is 0 or 1: the reference's three cases).  The final powf(10, log_ratio (db - max_db) / 20) runs as one exp2f: float ulps on a gain.
__device__ inline void loudness_seek(Ramp &r, const float *__restrict__ lf, const float *__restrict__ le, float freq) {
  int k = min(max(r.loud_k, 0), 27);
  while (k < 27 && freq >= lf[k + 1]) ++k;
  while (k > 0 && freq < lf[k]) --k;
  r.loud_k = k;
  r.lo_f = lf[k]; r.hi_f = lf[k + 1]; r.inv_w = 1.f / (r.hi_f - r.lo_f); r.lo_e = le[k]; r.hi_e = le[k + 1];
}
__device__ inline float loudness_volume(Ramp &r, const RaVoiceParams &p, const float *__restrict__ lf, const float *__restrict__ le, float freq) {
  if ((freq < r.lo_f && r.loud_k > 0) || (freq >= r.hi_f && r.loud_k < 27)) loudness_seek(r, lf, le, freq);
  const float ratio = fminf(fmaxf((freq - r.lo_f) * r.inv_w, 0.f), 1.f);
  const float db = ratio * r.hi_e + (1.f - ratio) * r.lo_e;
  if (db > p.max_db) return 1.f;
  return exp2f(p.db_to_log2 * (db - p.max_db));
}

__device__ inline double rcp_newton(double d) { // 1 / d: v_rcp_f64 and two Newton steps instead of the IEEE division (an ulp)
  double r = __builtin_amdgcn_rcp(d);
  r = r * fma(-d, r, 2.0);
  r = r * fma(-d, r, 2.0);
  return r;
}
__device__ inline double one_pole_k(double inc_abs) { // a / (1 + a), a = pi inc (Filter<LOW_PASS>)
  const double a = M_PI * inc_abs;
  return a * rcp_newton(1.0 + a);
}

// itp::interpolate(kind, t, b, c, d) as decided (Penner's equations): every one of them is b + c g(t / d)
__device__ __noinline__ double ease01(int kind, double u) {
  switch (kind) {
    case RA_ITP_EASE_IN_QUAD: return u * u;
    case RA_ITP_EASE_OUT_QUAD: return -u * (u - 2.0);
    case RA_ITP_EASE_INOUT_QUAD: { const double v = 2.0 * u; if (v < 1.0) return 0.5 * v * v; const double w = v - 1.0; return -0.5 * (w * (w - 2.0) - 1.0); }
    case RA_ITP_EASE_IN_CUBIC: return u * u * u;
    case RA_ITP_EASE_OUT_CUBIC: { const double w = u - 1.0; return w * w * w + 1.0; }
    case RA_ITP_EASE_INOUT_CUBIC: { const double v = 2.0 * u; if (v < 1.0) return 0.5 * v * v * v; const double w = v - 2.0; return 0.5 * (w * w * w + 2.0); }
    case RA_ITP_EASE_IN_QUART: { const double u2 = u * u; return u2 * u2; }
    case RA_ITP_EASE_OUT_QUART: { const double w = u - 1.0, w2 = w * w; return -(w2 * w2 - 1.0); }
    case RA_ITP_EASE_INOUT_QUART: { const double v = 2.0 * u; if (v < 1.0) { const double v2 = v * v; return 0.5 * v2 * v2; } const double w = v - 2.0, w2 = w * w; return -0.5 * (w2 * w2 - 2.0); }
    case RA_ITP_EASE_IN_QUINT: { const double u2 = u * u; return u2 * u2 * u; }
    case RA_ITP_EASE_OUT_QUINT: { const double w = u - 1.0, w2 = w * w; return w2 * w2 * w + 1.0; }
    case RA_ITP_EASE_INOUT_QUINT: { const double v = 2.0 * u; if (v < 1.0) { const double v2 = v * v; return 0.5 * v2 * v2 * v; } const double w = v - 2.0, w2 = w * w; return 0.5 * (w2 * w2 * w + 2.0); }
    case RA_ITP_EASE_IN_SINE: return 1.0 - cospi(0.5 * u);
    case RA_ITP_EASE_OUT_SINE: return sinpi(0.5 * u);
    case RA_ITP_EASE_INOUT_SINE: return -0.5 * (cospi(u) - 1.0);
    case RA_ITP_EASE_IN_EXPO: return (u == 0.0) ? 0.0 : exp2(10.0 * (u - 1.0));
    case RA_ITP_EASE_OUT_EXPO: return (u == 1.0) ? 1.0 : 1.0 - exp2(-10.0 * u);
    case RA_ITP_EASE_INOUT_EXPO: {
      if (u == 0.0) return 0.0;
      if (u == 1.0) return 1.0;
      const double v = 2.0 * u;
      if (v < 1.0) return 0.5 * exp2(10.0 * (v - 1.0));
      return 0.5 * (2.0 - exp2(-10.0 * (v - 1.0)));
    }
    case RA_ITP_EASE_IN_CIRC: return -(sqrt(1.0 - u * u) - 1.0);
    case RA_ITP_EASE_OUT_CIRC: { const double w = u - 1.0; return sqrt(1.0 - w * w); }
    case RA_ITP_EASE_INOUT_CIRC: { const double v = 2.0 * u; if (v < 1.0) return -0.5 * (sqrt(1.0 - v * v) - 1.0); const double w = v - 2.0; return 0.5 * (sqrt(1.0 - w * w) + 1.0); }
    default: return u;
  }
}

// LogRamp::do_step (audioelement.h:2522-2542) in two halves.  ramp_tick: the wrap at the ramp's end, then `cur` moves on by C times the
// increment half a sample later; it returns the time of this step.  ramp_value: the increment at that time.
// The time step cur + C (from + (to - from) (cur + 1/2) / dur) is affine in cur: it runs as ONE multiply-add with the two constants formed
// once per ramp (al = 1 + C (to - from) / dur, be = C (from + (to - from) / (2 dur)); swapped ends: 2 - al (exact for al >= 1) and
// C (to - (to - from) / (2 dur))).
// All three kernels step time with this very code, so their states agree bit for bit; against the reference's operation order the
// trajectory differs by rounding (1e-16 relative per step, a few 1e-12 of the duration after the longest ramp).
__device__ inline bool ramp_turn(Ramp &r) { // the ramp has reached its end: it restarts from the other end (periodic ramp)
  const bool w = r.cur + 0.5 > r.dur;
  if (__builtin_amdgcn_ballot_w64(w) == 0) return false; // (a wave-uniform branch: the turn is one step of a ramp's thousands — without it the selects below run every step)
  if (w) {
    r.cur = 0.0;
    double t = r.from; r.from = r.to; r.to = t;
    r.al = 2.0 - r.al;
    t = r.be; r.be = r.be_o; r.be_o = t;
    r.l2r = -r.l2r; r.swapped ^= 1;
  }
  return w;
}
__device__ inline double ramp_tick(Ramp &r) {
  ramp_turn(r);
  const double c0 = r.cur;
  r.cur = fma(r.al, c0, r.be);
  return c0;
}
__device__ inline double ramp_value(const Ramp &r, double c0) {
  const double d = r.to - r.from, u = c0 * r.inv_dur;
  if (r.interp == RA_ITP_LINEAR) return fma(d, u, r.from);
  if (r.interp == RA_ITP_PROPORTIONAL_VALUE_DERIVATIVE) return r.from * exp2(u * r.l2r);
  return fma(d, ease01(r.interp, u), r.from);
}

// One step of what carries state besides `cur`: the phase sum and the loudness volume (LoudnessVolumeAdjusted::setAngleIncrements
// :1291-1298, BaseVolumeAdjusted::step :1195-1216: the filter starts at its first target; its coefficient is re-derived from the
// increment whenever the volume still differs from the target — when it does not, the coefficient does not matter).
// k_out / tgt_out: the step as the affine map y -> (1 - k) y + k tgt, for the pre-pass' composition.
__device__ inline double ramp_advance(Ramp &r, const RaVoiceParams &p, const float *__restrict__ lf, const float *__restrict__ le, bool first, double &k_out, double &tgt_out) {
  const double inc = ramp_value(r, ramp_tick(r));
  const double tgt = (double)loudness_volume(r, p, lf, le, (float)(inc * 0.5 * (double)p.sample_rate));
  const double k = first ? 1.0 : one_pole_k(fabs(inc));
  k_out = k; tgt_out = tgt;
  return inc;
}
__device__ inline double wrap2(double th) { return th - 2.0 * floor(th * 0.5); }

__device__ inline void ramp_setup(Ramp &r, const RaVoiceJob &J) {
  r.from = J.from; r.to = J.to; r.dur = J.dur; r.inv_dur = rcp_newton(J.dur);
  { // J.C: get_linear_proportionality_constant :2560-2571, from the host (symmetric in from / to: a swap keeps it)
    const double g = (J.to - J.from) * r.inv_dur;
    r.al = fma(J.C, g, 1.0); r.be = J.C * fma(0.5, g, J.from);
    r.be_o = J.C * fma(-0.5, g, J.to);
  }
  r.interp = J.interp;
  r.l2r = J.l2r;
  r.cur = J.cur0; r.theta = 0.0; r.y = 0.0; r.loud_k = 14; r.swapped = 0;
}
__device__ inline void ramp_swap_if(Ramp &r, int swapped) {
  if (swapped != r.swapped) {
    double t = r.from; r.from = r.to; r.to = t;
    r.al = 2.0 - r.al;
    t = r.be; r.be = r.be_o; r.be_o = t;
    r.l2r = -r.l2r; r.swapped = swapped;
  }
}
__device__ inline void ramp_load(Ramp &r, const RaVoiceJob &J, const RaVoiceState *st, const float *__restrict__ lf, const float *__restrict__ le) { // st == nullptr: the job's first step is next
  ramp_setup(r, J);
  if (st) { r.cur = st->cur; r.theta = st->theta; r.y = st->y; r.loud_k = st->loud_k; ramp_swap_if(r, st->swapped); }
  r.lo_f = 1.f; r.hi_f = 0.f; r.inv_w = 0.f; r.lo_e = 0.f; r.hi_e = 0.f; // (an empty interval: the first step seeks)
  loudness_seek(r, lf, le, lf[min(max(r.loud_k, 0), 27)]);
}

__device__ inline void load_tables(const RaVoiceParams &p, float *lf, float *le) { // (a dynamically indexed kernel-argument array would live in scratch)
  for (int j = threadIdx.x; j < 29; j += blockDim.x) { lf[j] = p.loud_freqs[j]; le[j] = p.loud_elv[j]; }
  __syncthreads();
}

// ---- pre-pass: the states a block's render needs, in three launches -------------------------------------------------------------------
// A job's states sit in consecutive slots: one per tile boundary it crosses, then one for the block's end if it goes on.  Slot i closes
// the job's i-th SEGMENT (from the previous slot's frame — or the job's first frame of the block — to its own).
//   A  one lane per job, samples in order: LogRamp's time alone (two dependent FMAs per sample) -> `cur` and the swap flag of every slot,
//      and which job a slot belongs to;
//   B  one lane per slot: its segment's samples from the segment's start time — the phase the segment adds, and the loudness volume's
//      one-pole over the segment as ONE affine map y -> a y + b (each sample is one: y + k (tgt - y));
//   C  one lane per job: the running phase and volume through its slots.
// B is where the work is (the increment, the loudness gain and the coefficient of every sample), and it is as parallel as the render.
__global__ void __launch_bounds__(64) ra_voice_pre_a_kernel(const RaVoiceJob *__restrict__ jobs, const int32_t *__restrict__ order, int n_work, const RaVoiceState *__restrict__ st_prev,
                                                             RaVoiceState *__restrict__ st_cur, RaVoiceSeg *__restrict__ seg, int L) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_work) return;
  const int ji = order[w];
  const RaVoiceJob J = jobs[ji];
  Ramp r;
  ramp_setup(r, J);
  if (J.state_in >= 0) { r.cur = st_prev[J.state_in].cur; ramp_swap_if(r, st_prev[J.state_in].swapped); }
  const int last = J.begin + J.n; // (exclusive)
  const int steps = (J.state_out >= 0) ? J.n : ((last - 1) / L) * L - J.begin; // up to the last state somebody reads
  int next_b = (J.begin / L + 1) * L, slot = J.state_tiles;
  // only time and the turn flag are carried here (ramp_tick without the ends): the chain per step is one multiply-add and a select
  double cur = r.cur, al = r.al, be = r.be, be_o = r.be_o;
  const double dur = r.dur;
  int sw = r.swapped;
  for (int i = 0; i < steps; ++i) {
    if (J.begin + i == next_b) { st_cur[slot].cur = cur; st_cur[slot].swapped = sw; seg[slot].job = ji; ++slot; next_b += L; }
    const bool w = cur + 0.5 > dur;
    if (__builtin_amdgcn_ballot_w64(w) != 0) { // (wave-uniform branch around the turn, see ramp_turn)
      const double b2 = w ? be_o : be;
      be_o = w ? be : be_o;
      be = b2;
      al = w ? 2.0 - al : al;
      sw ^= w ? 1 : 0;
      cur = w ? 0.0 : cur;
    }
    cur = fma(al, cur, be);
  }
  if (J.begin + steps == next_b && J.begin + steps < last) { st_cur[slot].cur = cur; st_cur[slot].swapped = sw; seg[slot].job = ji; }
  if (J.state_out >= 0) { st_cur[J.state_out].cur = cur; st_cur[J.state_out].swapped = sw; seg[J.state_out].job = ji; }
}

__global__ void __launch_bounds__(256) ra_voice_pre_b_kernel(RaVoiceParams p, const RaVoiceJob *__restrict__ jobs, int n_slots, const RaVoiceState *__restrict__ st_prev,
                                                              const RaVoiceState *__restrict__ st_cur, RaVoiceSeg *__restrict__ seg, int L) {
  __shared__ float lf[29], le[29];
  load_tables(p, lf, le);
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_slots) return;
  const RaVoiceJob J = jobs[seg[slot].job];
  const int s = slot - J.state_tiles;                       // the segment's index within the job
  const int first_b = (J.begin / L + 1) * L;                // the job's first tile boundary
  const int f0 = (s == 0) ? J.begin : first_b + (s - 1) * L;
  const int f1 = (slot == J.state_out) ? J.begin + J.n : first_b + s * L;
  Ramp r;
  ramp_setup(r, J);
  if (s == 0) { if (J.state_in >= 0) { r.cur = st_prev[J.state_in].cur; r.loud_k = st_prev[J.state_in].loud_k; ramp_swap_if(r, st_prev[J.state_in].swapped); } }
  else { r.cur = st_cur[slot - 1].cur; ramp_swap_if(r, st_cur[slot - 1].swapped); }
  r.lo_f = 1.f; r.hi_f = 0.f; r.inv_w = 0.f; r.lo_e = 0.f; r.hi_e = 0.f;
  r.loud_k = 14;
  loudness_seek(r, lf, le, lf[14]);
  double dth = 0.0, a = 1.0, b = 0.0;
  int j = J.j0 + (f0 - J.begin);
  for (int f = f0; f < f1; ++f, ++j) {
    double k, tgt;
    dth += ramp_advance(r, p, lf, le, j == 1, k, tgt);
    a = a - k * a;             // (1 - k) a
    b = fma(k, tgt - b, b);    // (1 - k) b + k tgt
  }
  seg[slot].dtheta = dth; seg[slot].a = a; seg[slot].b = b; seg[slot].loud_k = r.loud_k;
}

__global__ void __launch_bounds__(64) ra_voice_pre_c_kernel(const RaVoiceJob *__restrict__ jobs, const int32_t *__restrict__ order, int n_work, const RaVoiceState *__restrict__ st_prev,
                                                             RaVoiceState *__restrict__ st_cur, const RaVoiceSeg *__restrict__ seg, int L) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_work) return;
  const RaVoiceJob J = jobs[order[w]];
  double th = 0.0, y = 0.0;
  if (J.state_in >= 0) { th = st_prev[J.state_in].theta; y = st_prev[J.state_in].y; }
  const int tiles = (J.begin + J.n - 1) / L - J.begin / L;
  const int n_slots = tiles + (J.state_out >= 0 ? 1 : 0);
  for (int i = 0; i < n_slots; ++i) {
    const int slot = J.state_tiles + i;
    th = wrap2(th + seg[slot].dtheta);
    y = fma(seg[slot].a, y, seg[slot].b);
    st_cur[slot].theta = th; st_cur[slot].y = y; st_cur[slot].loud_k = seg[slot].loud_k;
  }
}

// ---- render --------------------------------------------------------------------------------------------------------------------------
// d = a b + c as the three-address instruction.  (The compiler turns a Horner chain whose coefficients live in VGPRs into v_mov_b64 + v_fmac_f64
// pairs — two issue slots per term; the chain below is a fifth of the render's instructions.)
__device__ inline double fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// sin(pi x), x in [0, 2]: k = the nearest integer, r = x - k in [-1/2, 1/2], sin(pi x) = (-1)^k sin(pi r); sin(pi r) / r as a polynomial of
// degree 6 in r^2 (a Chebyshev fit on [0, 1/4]: within 8e-14 of the function — the output is a float — against 4e-16 for degree 7; every
// coefficient is a VGPR pair held through the loop), the sign as a flip of the sign bit
__device__ inline double osc_imag(double theta) {
  const double kf = rint(theta), r = theta - kf;
  const double z = r * r;
  double v = 0.00044749586935114105;
  v = fma3(v, z, -0.007364008149410389);
  v = fma3(v, z, 0.08214479537170938);
  v = fma3(v, z, -0.5992644340182135);
  v = fma3(v, z, 2.55016403591087);
  v = fma3(v, z, -5.167712779988022);
  v = fma3(v, z, 3.141592653589636);
  v *= r;
  const unsigned flip = ((unsigned)(int)kf & 1u) << 31;
  return __hiloint2double(__double2hiint(v) ^ (int)flip, __double2loint(v));
}
__device__ inline double alias_gain(double inc) { // freqAliasingMultiplicator, audioelement.h:466-483
  if (inc > 0.0 && inc <= 0.25) return 1.0;
  if (inc == 0.0) return 1.0;
  const double half_n = rcp_newton(inc);
  if (half_n < 1.0) return 0.0;
  if (half_n > 4.0) return 1.0;
  return (half_n - 1.0) * (1.0 / 3.0);
}

struct Seg { // the piece of a job a lane renders inside its tile
  Ramp r;
  double cL, cR, inv_A, inv_R, top; // cL / cR = the job's output factor times the ramp's volume times the Mix gain (constants of the job)
  int lo, hi, j, T, A;
};

__device__ inline void seg_open(Seg &g, const RaVoiceJob &J, int f_lo, int f_hi, int q, int L, const RaVoiceState *__restrict__ st_prev, const RaVoiceState *__restrict__ st_cur,
                                const float *__restrict__ lf, const float *__restrict__ le, float sine_gain) {
  g.lo = max(J.begin, f_lo);
  g.hi = min(J.begin + J.n, f_hi);
  const RaVoiceState *st;
  if (g.lo == J.begin) st = (J.state_in >= 0) ? st_prev + J.state_in : nullptr;
  else st = st_cur + J.state_tiles + (q - (J.begin / L + 1));
  ramp_load(g.r, J, st, lf, le);
  g.j = J.j0 + (g.lo - J.begin);
  g.T = J.T; g.A = J.A;
  g.inv_A = rcp_newton((double)J.A); g.inv_R = rcp_newton((double)J.R);
  g.top = (J.T < J.A) ? (double)J.T * g.inv_A : 1.0; // the envelope's value when the key is released (startReleased, audioelement.h:836-841)
  g.cL = J.fL * (J.vol * (double)sine_gain); g.cR = J.fR * (J.vol * (double)sine_gain);
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT) ra_voice_render_kernel(RaVoiceParams p, const RaVoiceJob *__restrict__ jobs, const int32_t *__restrict__ slot_begin,
                                                                const RaVoiceState *__restrict__ st_prev, const RaVoiceState *__restrict__ st_cur, int n_streams, int n,
                                                                int L, int n_tiles, int g_log2, float *__restrict__ out) {
  const int l = threadIdx.x, nt = blockDim.x, g = 1 << g_log2;
  double *tile = reinterpret_cast<double *>(ra_smem_voice); // [nt][2 TS + 1]: L then R samples of the lane
  float *lf = reinterpret_cast<float *>(tile + (size_t)nt * (2 * TS + 1)), *le = lf + 29;
  load_tables(p, lf, le);
  const int U = nt >> g_log2;                       // (stream, tile) units of this workgroup
  const int unit0 = (int)blockIdx.x * U, n_units = n_streams * n_tiles; // (the launcher refuses more than 2^31 - 1 units)
  const int u = l >> g_log2, s = l & (g - 1);
  const int unit = unit0 + u;
  const bool valid = unit < n_units;
  const int stream = valid ? unit / n_tiles : 0, q = valid ? unit - stream * n_tiles : 0;
  const int f_lo = q * L, f_hi = min(n, f_lo + L);
  // the slot's jobs are disjoint and in time order: the first one that ends after the tile's begin
  int jb = 0, je = 0;
  if (valid) {
    jb = slot_begin[(size_t)stream * g + s];
    je = slot_begin[(size_t)stream * g + s + 1];
    int lo = jb, hi = je;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (jobs[mid].begin + jobs[mid].n > f_lo) hi = mid; else lo = mid + 1;
    }
    jb = lo;
  }
  Seg sg;
  bool have = false;
  constexpr int kNever = 0x7fffffff;
  int next_lo = kNever; // the frame at which the slot's next job starts to sound inside this tile
  if (jb < je) { const int b = jobs[jb].begin; if (b < f_hi) next_lo = max(b, f_lo); }
  double *mine = tile + (size_t)l * (2 * TS + 1);
  const int n_out = U * TS * 2;
  for (int base = 0; base < L; base += TS) { // (every unit of the workgroup has the same tile length but the block's last tile may be cut: f_hi)
    const int fb = f_lo + base;
    const bool work = (have || next_lo < fb + TS) && fb < f_hi;
    if (!__syncthreads_or(work ? 1 : 0)) {
      for (int t = l; t < n_out; t += nt) {
        const int uu = t / (TS * 2), i = (t >> 1) % TS, ch = t & 1;
        const int un = unit0 + uu;
        if (un >= n_units) continue;
        const int st2 = un / n_tiles, q2 = un - st2 * n_tiles;
        const int f = q2 * L + base + i;
        if (f < min(n, q2 * L + L)) out[((size_t)st2 * n + f) * 2 + ch] = 0.f;
      }
      continue;
    }
#pragma unroll 1
    for (int i = 0; i < TS; ++i) {
      const int f = fb + i;
      double vL = 0.0, vR = 0.0;
      if (f == next_lo) { // (the one place a job is opened: at the tile's first frame if it is already sounding, else where it begins)
        seg_open(sg, jobs[jb], f_lo, f_hi, q, L, st_prev, st_cur, lf, le, p.sine_gain);
        have = true;
        next_lo = kNever;
      }
      if (have) {
        // ---- one step of a sounding ramp: Enveloped::step, then the imag() chain of the element (audioelement.h:1382-1384 -> 1126-1129 -> 227-231) ----
        double k, tgt;
        const double inc = ramp_advance(sg.r, p, lf, le, sg.j == 1, k, tgt);
        sg.r.theta = wrap2(sg.r.theta + inc);
        sg.r.y = fma(k, tgt - sg.r.y, sg.r.y);
        const int j = sg.j;
        const double env = (j <= sg.T) ? ((j < sg.A) ? (double)j * sg.inv_A : 1.0)                       // attack then sustain 1 (stepPressed :960-1000)
                                       : fma(-sg.top, (double)(j - sg.T) * sg.inv_R, sg.top);         // release: top + (-top) m / R (:732-737)
        const double osc = alias_gain(inc) * osc_imag(sg.r.theta);                                       // SineOscillatorAlgo::imag :2424
        // Mix::imag (gain * (loudness volume * osc), :1867-1873) under the envelope, the ramp's volume, the stereo gain and the element's
        // volume: the constant factors are folded into cL / cR (an ulp from the reference's order of the multiplications)
        const double mono = (sg.r.y * osc) * env;
        vL = sg.cL * mono; vR = sg.cR * mono;
        ++sg.j;
        if (f + 1 == sg.hi) { // the segment ends: the slot's next job may begin inside this tile
          have = false;
          ++jb;
          if (jb < je) { const int b = jobs[jb].begin; if (b < f_hi) next_lo = b; }
        }
      }
      mine[i] = vL; mine[TS + i] = vR;
    }
    __syncthreads();
    for (int t = l; t < n_out; t += nt) { // buffer[2 i + ch] += elem.imag(ch): the unit's slots in slot order, then the float cast of SimpleAudioOutContext::step
      const int uu = t / (TS * 2), i = (t >> 1) % TS, ch = t & 1;
      const int un = unit0 + uu;
      if (un >= n_units) continue;
      const double *col = tile + (size_t)(uu << g_log2) * (2 * TS + 1) + ch * TS + i;
      double acc = 0.0;
      for (int k = 0; k < g; ++k) acc += col[(size_t)k * (2 * TS + 1)];
      const int st2 = un / n_tiles, q2 = un - st2 * n_tiles;
      const int f = q2 * L + base + i;
      if (f < min(n, q2 * L + L)) out[((size_t)st2 * n + f) * 2 + ch] = (float)acc;
    }
    __syncthreads();
  }
}

#include "ra_wind.cuh"

} // namespace

// One chunk [c0, c0 + clen) of a WIND block in two halves, so that the host can run the controls of chunk k + 1 on a queue of their own beside the filter
// chains of chunk k (ra_wind.cuh): the controls (+ the sine voice) write the coefficients and the sine's output; the chains and the mix read them.
void ra_wind_launch_controls(const RaVoiceParams &vp, const RaWindParams &wp, const RaWindJob *jobs, int n_jobs, const double *pink, RaWindState *states, const RaWindScratch &sc, int c0,
                             int clen, hipStream_t st) {
  if (clen <= 0 || n_jobs <= 0) return;
  hipLaunchKernelGGL(ra_wind_ctl_kernel, dim3((unsigned)n_jobs, 3), dim3(64), 0, st, vp, wp, jobs, pink, states, sc, c0, clen);
}
void ra_wind_launch_chains(const RaWindParams &wp, const RaWindJob *jobs, int n_jobs, const int32_t *stream_begin, int n_streams, const double *pink, double *fstates,
                           const RaWindScratch &sc, int c0, int clen, int64_t n, float *d_out, hipStream_t st) {
  if (clen <= 0) return;
  if (n_jobs > 0) {
    static const int forced_m = std::getenv("RA_WIND_SECTIONS_PER_LANE") ? std::atoi(std::getenv("RA_WIND_SECTIONS_PER_LANE")) : 0; // tuning aid
    static const bool no_scan = std::getenv("RA_WIND_NO_SCAN") != nullptr; // tuning aid: the wave pipeline for short chains too
    auto chains = [&](int first, int count, int sections) {
      const dim3 grid((unsigned)n_jobs, (unsigned)count);
      // A chain that fits one wave runs as scans over time (ra_wind_scan_kernel).  Longer ones as a pipeline of waves, one section per lane while that keeps a
      // SIMD's few waves busy: a step's instructions are one dependent chain, and with two sections per lane the chain is twice as long (measured at 1024 ramps
      // of 178 sections: 3.6 ms per chunk against 4.7; at 4096 ramps two per lane win by 5 %).  Orders above 160 need two per lane (five waves of LDS rings).
      const int w1 = (sections + 63) / 64, w2 = (sections + 127) / 128;
      const bool two = forced_m ? forced_m == 2 : (sections > 320 || (long long)n_jobs * count * w2 >= 8192);
      if (sections <= 64 && !no_scan) hipLaunchKernelGGL(ra_wind_scan_kernel, grid, dim3(64), 0, st, wp, jobs, pink, fstates, sc, c0, clen, first);
      else if ((!two || sections <= 64) && sections <= 320)
        hipLaunchKernelGGL(ra_wind_chain_kernel<1>, grid, dim3((unsigned)(64 * w1)), (size_t)w1 * kWindWaveLds * sizeof(double), st, wp, jobs, pink, fstates, sc, c0, clen, first);
      else hipLaunchKernelGGL(ra_wind_chain_kernel<2>, grid, dim3((unsigned)(64 * w2)), (size_t)w2 * kWindWaveLds * sizeof(double), st, wp, jobs, pink, fstates, sc, c0, clen, first);
    };
    if (wp.g_bp != 0.f) chains(0, 1, 2 * wp.order);
    if (wp.g_br != 0.f) chains(1, 2, wp.order);
  }
  hipLaunchKernelGGL(ra_wind_mix_kernel, dim3((unsigned)((clen + 255) / 256), (unsigned)n_streams), dim3(256), 0, st, wp, jobs, stream_begin, sc, c0, clen, n, d_out);
}

void ra_voice_launch_pre(const RaVoiceParams &p, const RaVoiceJob *jobs, const int32_t *order, int n_work, int n_slots, const RaVoiceState *st_prev, RaVoiceState *st_cur,
                         RaVoiceSeg *seg, int L, hipStream_t st) {
  if (n_work <= 0 || n_slots <= 0) return;
  hipLaunchKernelGGL(ra_voice_pre_a_kernel, dim3((unsigned)((n_work + 63) / 64)), dim3(64), 0, st, jobs, order, n_work, st_prev, st_cur, seg, L);
  hipLaunchKernelGGL(ra_voice_pre_b_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, p, jobs, n_slots, st_prev, st_cur, seg, L);
  hipLaunchKernelGGL(ra_voice_pre_c_kernel, dim3((unsigned)((n_work + 63) / 64)), dim3(64), 0, st, jobs, order, n_work, st_prev, st_cur, seg, L);
}

size_t ra_voice_render_lds(int threads) { return (size_t)threads * (2 * TS + 1) * sizeof(double) + 58 * sizeof(float); }

int ra_voice_launch_render(const RaVoiceParams &p, const RaVoiceJob *jobs, const int32_t *slot_begin, const RaVoiceState *st_prev, const RaVoiceState *st_cur, int n_streams,
                           int64_t n, int L, int g_log2, float *d_out, hipStream_t st) {
  if (n <= 0) return 0;
  const int g = 1 << g_log2;
  const int threads = std::max(64, g);
  const int U = threads / g;
  const int n_tiles = (int)((n + L - 1) / L);
  const long long units = (long long)n_streams * n_tiles;
  if (units > 0x7fffffffll - 1024) return -2;
  const size_t lds = ra_voice_render_lds(threads);
  if (threads == 64) {
    hipLaunchKernelGGL(ra_voice_render_kernel<64>, dim3((unsigned)((units + U - 1) / U)), dim3(64), lds, st, p, jobs, slot_begin, st_prev, st_cur, n_streams, (int)n, L, n_tiles, g_log2,
                       d_out);
    return 0;
  }
  // more than 64 ramps of a stream at once (n_channels > 21): one unit per workgroup of g lanes.  Function attributes are per device and a
  // process may hold voices on several GPUs from several host threads: once per device, under the site's lock
  static unsigned long long attr_done = 0;
  static int attr_failed = 0;
  ra_once_per_device(attr_done, [] {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voice_render_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ra_voice_render_lds(1024)) != hipSuccess) attr_failed = 1;
  });
  if (attr_failed) return -1;
  hipLaunchKernelGGL(ra_voice_render_kernel<1024>, dim3((unsigned)((units + U - 1) / U)), dim3((unsigned)threads), lds, st, p, jobs, slot_begin, st_prev, st_cur, n_streams, (int)n, L,
                     n_tiles, g_log2, d_out);
  return 0;
}
