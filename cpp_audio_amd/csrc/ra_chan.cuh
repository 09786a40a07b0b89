// One synth channel = one audio element of the reference, held in registers by one lane.
//
// Restates, in the reference's operation order (fp64), the per-sample chain of
//   ResynthElement<double> (source/rt.resynth.lib.cpp:6-19) =
//   StereoPanned (include/audioelement.h:1319-1389) > VolumeAdjusted (:1110-1265) > Enveloped (:181-291)
//   > AHDSREnvelope (:665-1102) + FreqCtrl_ (:3055-3155) > InterpolatedFreq (:2705-2817) + SineOscillatorAlgo (:2365-2452)
// (mode 0), and of driver (A)'s Enveloped<VolumeAdjusted<SineOscillatorAlgo<double>>, AHDSREnvelope> (include/resynth.hpp:13-15)
// (mode 1: no frequency glide, the increment reaches the oscillator directly).
#pragma once
#include "ra_internal.h"

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
  double theta; // phase-domain oscillator (propagate kernel): angle in rad/pi, cur = e^{i pi theta}
  int flags;
};

__device__ inline void chan_init(Chan &c) { // a freshly constructed element
  c.state = ENV_DONE2; c.counter = 0; c.reldelay = -1; c.ahdstate = -1; c.ahdcounter = 0; c.minch = 0; c.nextminch = 0;
  c.value = 0; c.top = 0;
  c.from = c.to = c.cur = c.C = c.fres = 0;
  c.cre = 1; c.cim = 0; c.mre = 1; c.mim = 0; c.alias = 1; c.inc = 0;
  c.vol = c.target = c.finit = c.k = c.y = 0;
  c.gL = c.gR = 1;
  c.theta = 0;
  c.flags = 0;
}

__device__ inline bool rt_active(const Chan &c) { return c.state != ENV_SOON && c.state != ENV_DONE2; } // audioelement.h:390-395

__device__ inline int env_max_counter(const Chan &c, const RaEnvParams &p) { // :1066-1084
  switch (c.ahdstate) {
    case 0: return max(c.minch, p.A);
    case 1: return p.H;
    default: return (p.D == 0) ? 0 : max(c.minch, p.D);
  }
}
// onAHDStateChange (:1040-1053) and stepAHD (:1055-1064) call each other; unrolled into a loop
__device__ inline void env_on_ahd_change(Chan &c, const RaEnvParams &p) {
  for (;;) {
    c.ahdcounter = 0;
    c.minch = c.nextminch;
    if (c.ahdstate < 0) return;
    if (env_max_counter(c, p) != 0) return;
    c.ahdstate = (c.ahdstate == 2) ? -1 : c.ahdstate + 1; // stepAHD with ahdCounter(0) >= maxCounter(0)
  }
}
__device__ inline void env_step_ahd(Chan &c, const RaEnvParams &p) {
  if (c.ahdcounter < env_max_counter(c, p)) return;
  c.ahdstate = (c.ahdstate == 2) ? -1 : c.ahdstate + 1;
  env_on_ahd_change(c, p);
}
__device__ inline double itp_linear(double t, double b, double cc, double d) { return cc * t / d + b; }

__device__ inline void env_step(Chan &c, const RaEnvParams &p) { // EnvelopeCRT::step :702-756
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
      int rt = max(c.minch, p.R); // getReleaseTime :1002-1004
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
__device__ inline void env_on_key_pressed(Chan &c, int delay) { c.state = ENV_PRESSED; c.counter = -delay; c.reldelay = -1; c.value = 0; } // :762-767
__device__ inline void env_on_key_released(Chan &c, int delay) { // :769-781
  if (c.state != ENV_PRESSED) return;
  if (c.counter + delay <= 0) c.state = ENV_DONE1; else c.reldelay = delay;
}
__device__ inline bool env_can_release_now(const Chan &c, int delay) { // :802-811
  return c.state == ENV_PRESSED && c.reldelay == -1 && (-c.counter <= delay);
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

// FreqCtrl_::angleIncrements :3117-3124 (mode 0) / SineOscillatorAlgo::angleIncrements (mode 1)
__device__ inline double angle_increments(const Chan &c, int mode) {
  if (mode == 1) return c.inc;
  return c.inc != 0. ? c.inc : c.from;
}

// StereoPanned::step -> BaseVolumeAdjusted::step (:1195-1216) -> Enveloped::step (:197-200) -> FreqCtrl_::step (:3126-3129)
__device__ inline void chan_step(Chan &c, const RaEnvParams &p, double dur, double max_filter_inc, int mode) {
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
  { // BaseVolumeAdjusted::step, audioelement.h:1195-1216.  Written with selects: conditional stores to c.k / c.y in the two branches get
    // merged by the compiler into one store through a pointer, which pins both fields in scratch memory for the whole kernel.
    const bool first = !(c.flags & FL_HAS_VOL);
    const double inc = fmin(max_filter_inc, fabs(angle_increments(c, mode)));
    const bool new_k = first || (c.vol != c.target && c.finit != inc);
    const double a = M_PI * inc; // one-pole low-pass coefficient (DESIGN.md §5, L0 decision)
    const double k_new = a / (1.0 + a);
    c.finit = new_k ? inc : c.finit;
    c.k = new_k ? k_new : c.k;
    const double y0 = first ? c.target : c.y;
    c.y = y0 + c.k * (c.target - y0);
    c.vol = c.y;
    c.flags |= FL_HAS_VOL;
  }
}

// ---- the same step with cheaper arithmetic, for the hop-parallel render kernel (not reference order) ------------------
// * divisions by the constant glide duration become multiplications by its reciprocal
// * while a note glides, the oscillator's per-sample rotation e^{i pi inc} is updated by the (tiny) change of inc with a
//   5th-order series instead of a full sincos per sample; every hop restarts from an exactly evaluated rotation
//   (chan_jump / note events), so the series error (< 1e-20 per step) and rounding drift (~1e-14 per hop) cannot build up
// * the anti-aliasing gain needs its division only above fs/8
// Differences to chan_step are at rounding level; parity bar: 1e-5 RMS against the CPU restatement.
__device__ inline void osc_set_inc_fast(Chan &c, double f) {
  if (c.inc == f) return;
  const double d = M_PI * (f - c.inc);
  if (fabs(d) < 1.0e-3) {
    const double d2 = d * d;
    const double sd = d * (1.0 + d2 * (-1.0 / 6.0 + d2 * (1.0 / 120.0)));
    const double cd = 1.0 + d2 * (-0.5 + d2 * (1.0 / 24.0));
    const double mr = c.mre * cd - c.mim * sd, mi = c.mre * sd + c.mim * cd;
    c.mre = mr; c.mim = mi;
  } else {
    double sn, cs;
    sincospi(f, &sn, &cs); // e^{i pi f}: no large-argument reduction to carry around (|f| < 1 on this path)
    c.mre = cs; c.mim = sn;
  }
  c.inc = f;
  c.alias = (f > 0. && f <= 0.25) ? 1.0 : alias_mult(f);
}

__device__ inline void chan_step_fast(Chan &c, const RaEnvParams &p, double dur, double inv_dur, double max_filter_inc) {
  env_step(c, p);
  { // InterpolatedFreq::step with x / dur -> x * inv_dur
    double f;
    if ((c.flags & FL_HAS_FRES) && c.fres == c.to) f = c.to;
    else {
      if (c.cur + 0.5 > dur) c.cur = dur;
      const double slope = (c.to - c.from) * inv_dur;
      c.fres = (c.cur == dur) ? c.from + (c.to - c.from) * (dur / dur) : c.from + slope * c.cur;
      c.flags |= FL_HAS_FRES;
      if (c.cur < dur) c.cur += c.C * (c.from + slope * (c.cur + 0.5));
      f = c.fres;
    }
    osc_set_inc_fast(c, f);
  }
  {
    double re = c.cre * c.mre - c.cim * c.mim;
    double im = c.cre * c.mim + c.cim * c.mre;
    double g = 0.5 * (3.0 - (re * re + im * im));
    c.cre = re * g;
    c.cim = im * g;
  }
  { // BaseVolumeAdjusted::step, audioelement.h:1195-1216.  Written with selects: conditional stores to c.k / c.y in the two branches get
    // merged by the compiler into one store through a pointer, which pins both fields in scratch memory for the whole kernel.
    const bool first = !(c.flags & FL_HAS_VOL);
    const double inc = fmin(max_filter_inc, fabs(angle_increments(c, 0)));
    const bool new_k = first || (c.vol != c.target && c.finit != inc);
    const double a = M_PI * inc;
    const double k_new = a / (1.0 + a);
    c.finit = new_k ? inc : c.finit;
    c.k = new_k ? k_new : c.k;
    const double y0 = first ? c.target : c.y;
    c.y = y0 + c.k * (c.target - y0);
    c.vol = c.y;
    c.flags |= FL_HAS_VOL;
  }
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

// forgetPastSignals: envelope :694-700, ctrl :2725-2727, volume :1168-1172
__device__ inline void chan_forget_past_signals(Chan &c) {
  c.counter = 0;
  c.minch = c.nextminch;
  c.flags = 0;
}

// ---------------------------------------------------------------------------------------------------
// chan_jump: the state after n calls of "if (rt_active) chan_step" (mode 0), without running them.
//
// Between two analysis frames nothing touches a channel, and each of its recurrences has a closed form:
//   envelope   : integer counters; the value is recomputed from them (bit-exact with stepping)
//   freq glide : cur' = (1+x) cur + beta  (InterpolatedFreq::step is affine in cur_sample)  -> expm1 / log1p
//   oscillator : unit phasor; phase += pi * sum(inc_m), and the sum telescopes: inc_m = (cur_{m+1} - cur_m)/C - x/(2C)...
//   volume     : one-pole towards a constant target with constant k -> (1-k)^n
// Steps at which a regime changes (stage rotation, release start, glide end, first step of a note, ...) are
// taken one at a time with chan_step itself, so only regular stretches are jumped.  Differences to stepping are
// rounding-level (1e-13 relative), far inside the 1e-5 RMS budget; ra_config.reference_order selects the
// sample-serial kernel instead.
// ---------------------------------------------------------------------------------------------------
__device__ inline int env_regular_steps(const Chan &c, const RaEnvParams &p) { // steps that only bump counters (0 = next step is a transition)
  int lim = 0x7fffffff;
  if (c.reldelay >= 0) lim = c.reldelay; // the step that finds reldelay == 0 starts the release
  switch (c.state) {
    case ENV_PRESSED: {
      if (c.counter < 1) return 0;
      if (c.ahdstate < 0) return lim;
      int a = c.ahdcounter, M = env_max_counter(c, p);
      return a < M ? min(lim, M - a) : 0;
    }
    case ENV_RELEASED: { int rt = max(c.minch, p.R); return max(0, min(lim, rt - 1 - c.counter)); }
    case ENV_DONE1: return max(0, min(lim, RA_N_FRAMES_PER_BUFFER - c.counter));
    default: return 0;
  }
}
__device__ inline void env_jump(Chan &c, const RaEnvParams &p, int j) { // j <= env_regular_steps
  if (c.reldelay >= 0) c.reldelay -= j;
  c.counter += j;
  switch (c.state) {
    case ENV_PRESSED:
      if (c.ahdstate < 0) c.value = 1.0 + (double)p.SMinusOne;
      else {
        c.ahdcounter += j;
        double from, tmf;
        switch (c.ahdstate) {
          case 0: from = 0; tmf = 1; break;
          case 1: from = 1; tmf = 0; break;
          default: from = 1; tmf = (double)p.SMinusOne; break;
        }
        c.value = itp_linear((double)c.ahdcounter, from, tmf, (double)env_max_counter(c, p));
      }
      break;
    case ENV_RELEASED: c.value = itp_linear((double)c.counter, c.top, -c.top, (double)max(c.minch, p.R)); break;
    default: c.value = 0; break;
  }
}

// small-argument elementary functions (the glide's per-step growth x = ln(to/from)/dur is ~1e-5): short series instead of
// libm calls keep the latency-bound propagate kernel short; outside their range they defer to libm
__device__ inline double expm1_small(double z) {
  if (!(fabs(z) < 0.25)) return expm1(z);
  double r = 1.0; // z (1 + z/2 (1 + z/3 (... (1 + z/14))))   truncation < 0.25^15/15! ~ 1e-21
#pragma unroll
  for (int k = 14; k >= 2; --k) r = fma(z * (1.0 / k), r, 1.0);
  return z * r;
}
__device__ inline double log1p_small(double x) {
  if (!(fabs(x) < 0.1)) return log1p(x);
  const double s = x / (2.0 + x), s2 = s * s; // log1p(x) = 2 atanh(x / (2 + x))
  double q = 1.0 / 13.0;
  q = fma(q, s2, 1.0 / 11.0); q = fma(q, s2, 1.0 / 9.0); q = fma(q, s2, 1.0 / 7.0); q = fma(q, s2, 1.0 / 5.0); q = fma(q, s2, 1.0 / 3.0);
  q = fma(q, s2, 1.0);
  return 2.0 * s * q;
}

__device__ inline void osc_rotate(Chan &c, double sum_inc) { // cur *= e^{i pi sum_inc}
  double r = sum_inc - 2.0 * rint(0.5 * sum_inc); // exact reduction to [-1, 1]
  double sn, cs;
  sincospi(r, &sn, &cs);
  double re = c.cre * cs - c.cim * sn, im = c.cre * sn + c.cim * cs;
  double g = 0.5 * (3.0 - (re * re + im * im));
  c.cre = re * g; c.cim = im * g;
}
__device__ inline void osc_set_inc_jump(Chan &c, double f) { // osc_set_inc with the exact-pi sincospi
  if (c.inc == f) return;
  c.inc = f;
  double sn, cs;
  sincospi(f, &sn, &cs);
  c.mre = cs; c.mim = sn;
  c.alias = (f > 0. && f <= 0.25) ? 1.0 : alias_mult(f);
}

__device__ inline void chan_jump(Chan &c, const RaEnvParams &p, double dur, double max_filter_inc, int64_t n) {
  const double inv_dur = 1.0 / dur;
  while (n > 0 && rt_active(c)) {
    bool serial = !(c.flags & FL_HAS_VOL);
    int j = env_regular_steps(c, p);
    if (j == 0) serial = true;
    // regime of the frequency control
    const bool steady = (c.flags & FL_HAS_FRES) && c.fres == c.to;
    const double f_end = c.from + (c.to - c.from) * (dur / dur);
    const bool clamped = !steady && (c.flags & FL_HAS_FRES) && c.cur == dur && c.fres == f_end;
    const bool gliding = !steady && !clamped;
    if (gliding && (c.to == c.from || c.cur + 0.5 > dur || !(c.flags & FL_HAS_FRES))) serial = true;
    if (serial) { chan_step_fast(c, p, dur, inv_dur, max_filter_inc); --n; continue; }
    if ((int64_t)j > n) j = (int)n;

    double sum_inc, fres_last = c.fres, u_j = c.cur;
    if (!gliding) {
      sum_inc = (double)j * c.fres; // constant increment (== to when steady)
    } else {
      const double slope = (c.to - c.from) * inv_dur;
      const double x = c.C * slope;                            // alpha - 1
      const double beta = c.C * c.from + 0.5 * x;
      const double Lg = log1p_small(x);
      const double A = c.cur + beta / x;
      const double lim = dur - 0.5;                            // a step with cur > lim clamps
      // E_m = alpha^m - 1 ; u_m = cur + E_m A
      double Ej = expm1_small((double)j * Lg);
      double Ejm1 = (Ej - x) / (1.0 + x);
      if (c.cur + Ejm1 * A > lim) { // the glide ends inside: jump only up to the first clamped step
        double est = log1p_small((lim - c.cur) / A) / Lg;
        int ms = (est == est && est < 2.0e9 && est >= 0.0) ? (int)est + 1 : 1;
        ms = max(1, min(ms, j));
        double Em = expm1_small((double)ms * Lg), Emm1 = (Em - x) / (1.0 + x);
        while (ms > 1 && c.cur + Emm1 * A > lim) { --ms; Em = Emm1; Emm1 = (Em - x) / (1.0 + x); }
        while (ms < j && !(c.cur + Em * A > lim)) { ++ms; Emm1 = Em; Em = fma(Em, x, Em + x); }
        j = ms; Ej = Em; Ejm1 = Emm1;
      }
      const double u_jm1 = c.cur + Ejm1 * A;
      u_j = c.cur + Ej * A;
      fres_last = c.from + slope * u_jm1;
      // inc_m = (u_{m+1} - u_m) / C - (to - from) / (2 dur)   (InterpolatedFreq::step, audioelement.h:2781-2785)
      sum_inc = (u_j - c.cur) / c.C - (double)j * 0.5 * slope;
    }
    // volume filter: needs a constant coefficient over the stretch
    if (c.vol != c.target) {
      double lo = gliding ? fmin(fabs(c.from + (c.to - c.from) * (c.cur * inv_dur)), fabs(fres_last)) : fabs(c.fres);
      double want = fmin(max_filter_inc, lo);
      bool ok = (c.finit == want) && (!gliding || lo >= max_filter_inc);
      if (!ok) { chan_step_fast(c, p, dur, inv_dur, max_filter_inc); --n; continue; }
      double q = exp((double)j * log1p_small(-c.k));
      c.y = c.target + (c.y - c.target) * q;
      c.vol = c.y;
    }
    env_jump(c, p, j);
    if (gliding) { c.cur = u_j; c.fres = fres_last; osc_set_inc_jump(c, fres_last); }
    osc_rotate(c, sum_inc);
    n -= j;
  }
}

// ---- phase-domain propagation ---------------------------------------------------------------------------------------------
// The propagate kernel is a dependency chain over hops; it carries the oscillator as its angle theta (cur = e^{i pi theta}), so
// a step is "theta += inc" and a jump "theta += sum(inc)": no trigonometry on the chain.  The render kernel rebuilds the phasor,
// the per-sample rotation and the anti-aliasing gain from (theta, inc) when it loads a hop's start state.
__device__ inline void chan_step_phase(Chan &c, const RaEnvParams &p, double dur, double inv_dur, double max_filter_inc) {
  env_step(c, p);
  {
    double f;
    if ((c.flags & FL_HAS_FRES) && c.fres == c.to) f = c.to;
    else {
      if (c.cur + 0.5 > dur) c.cur = dur;
      const double slope = (c.to - c.from) * inv_dur;
      c.fres = (c.cur == dur) ? c.from + (c.to - c.from) * (dur / dur) : c.from + slope * c.cur;
      c.flags |= FL_HAS_FRES;
      if (c.cur < dur) c.cur += c.C * (c.from + slope * (c.cur + 0.5));
      f = c.fres;
    }
    c.inc = f;
  }
  c.theta += c.inc;
  { // BaseVolumeAdjusted::step, audioelement.h:1195-1216.  Written with selects: conditional stores to c.k / c.y in the two branches get
    // merged by the compiler into one store through a pointer, which pins both fields in scratch memory for the whole kernel.
    const bool first = !(c.flags & FL_HAS_VOL);
    const double inc = fmin(max_filter_inc, fabs(angle_increments(c, 0)));
    const bool new_k = first || (c.vol != c.target && c.finit != inc);
    const double a = M_PI * inc;
    const double k_new = a / (1.0 + a);
    c.finit = new_k ? inc : c.finit;
    c.k = new_k ? k_new : c.k;
    const double y0 = first ? c.target : c.y;
    c.y = y0 + c.k * (c.target - y0);
    c.vol = c.y;
    c.flags |= FL_HAS_VOL;
  }
}

// chan_jump in the phase domain; a glide that ends inside the stretch is taken in one go (glide part, clamp step, constant tail)
__device__ inline void chan_jump_phase(Chan &c, const RaEnvParams &p, double dur, double max_filter_inc, int64_t n) {
  const double inv_dur = 1.0 / dur;
  double k_cached = -1.0, lk = 0.0;
  while (n > 0 && rt_active(c)) {
    bool serial = !(c.flags & FL_HAS_VOL);
    int j = env_regular_steps(c, p);
    if (j == 0) serial = true;
    const bool steady = (c.flags & FL_HAS_FRES) && c.fres == c.to;
    const double f_end = c.from + (c.to - c.from) * (dur / dur);
    const bool clamped = !steady && (c.flags & FL_HAS_FRES) && c.cur == dur && c.fres == f_end;
    const bool gliding = !steady && !clamped;
    if (gliding && (c.to == c.from || c.cur + 0.5 > dur || !(c.flags & FL_HAS_FRES))) serial = true;
    if (serial) { chan_step_phase(c, p, dur, inv_dur, max_filter_inc); --n; continue; }
    if ((int64_t)j > n) j = (int)n;

    double sum_inc, new_cur = c.cur, new_fres = c.fres;
    if (!gliding) {
      sum_inc = (double)j * c.fres;
    } else {
      const double slope = (c.to - c.from) * inv_dur;
      const double x = c.C * slope;                            // alpha - 1
      const double beta = c.C * c.from + 0.5 * x;
      const double Lg = log1p_small(x);
      const double A = c.cur + beta / x;
      const double lim = dur - 0.5;                            // a step that finds cur > lim clamps
      const double inv_1px = 1.0 / (1.0 + x);
      double Ej = expm1_small((double)j * Lg);                 // E_m = alpha^m - 1 ; u_m = cur + E_m A
      double Ejm1 = (Ej - x) * inv_1px;
      if (c.cur + Ejm1 * A > lim) { // the glide ends inside the stretch
        double est = log1p_small((lim - c.cur) / A) / Lg;
        int ms = (est == est && est < 2.0e9 && est >= 0.0) ? (int)est + 1 : 1;
        ms = max(1, min(ms, j));
        double Em = expm1_small((double)ms * Lg), Emm1 = (Em - x) * inv_1px;
        while (ms > 1 && c.cur + Emm1 * A > lim) { --ms; Em = Emm1; Emm1 = (Em - x) * inv_1px; }
        while (ms < j && !(c.cur + Em * A > lim)) { ++ms; Emm1 = Em; Em = fma(Em, x, Em + x); }
        // steps 0 .. ms-1 glide; if ms < j step ms clamps and every later step keeps f_end
        const double u_ms = c.cur + Em * A;
        sum_inc = (u_ms - c.cur) / c.C - (double)ms * 0.5 * slope;
        if (ms < j) { sum_inc += (double)(j - ms) * f_end; new_cur = dur; new_fres = f_end; }
        else { new_cur = u_ms; new_fres = c.from + slope * (c.cur + Emm1 * A); }
      } else {
        new_cur = c.cur + Ej * A;
        new_fres = c.from + slope * (c.cur + Ejm1 * A);
        // inc_m = (u_{m+1} - u_m) / C - (to - from) / (2 dur)   (InterpolatedFreq::step, audioelement.h:2781-2785)
        sum_inc = (new_cur - c.cur) / c.C - (double)j * 0.5 * slope;
      }
    }
    if (c.vol != c.target) { // volume filter: needs a constant coefficient over the stretch
      double lo = gliding ? fmin(fabs(c.from + (c.to - c.from) * (c.cur * inv_dur)), fabs(new_fres)) : fabs(c.fres);
      double want = fmin(max_filter_inc, lo);
      bool ok = (c.finit == want) && (!gliding || lo >= max_filter_inc);
      if (!ok && gliding && c.from > 0. && c.to > 0.) {
        // a glide below the filter's maximum increment: the coefficient k = a / (1 + a), a = pi min(max_filter_inc, |inc|), follows the
        // increment step by step (BaseVolumeAdjusted::step, audioelement.h:1195-1216).  Only the filter is stepped (a lean loop; the
        // reciprocal tracks 1 + a by two Newton steps per sample, exact to rounding since a moves by ~1e-6 per step); glide, envelope
        // and angle keep their closed forms.
        const double slope = (c.to - c.from) * inv_dur, alpha = 1.0 + c.C * slope, beta = c.C * (c.from + 0.5 * slope), lim = dur - 0.5;
        double cur = c.cur, y = c.y, fin = c.finit, kk = c.k, rr = 1.0 / (1.0 + M_PI * c.finit);
        for (int i = 0; i < j; ++i) {
          const bool ended = cur > lim;
          const double f = ended ? f_end : fma(slope, cur, c.from);
          cur = ended ? dur : fma(alpha, cur, beta);
          if (y != c.target) {
            const double w = fmin(max_filter_inc, fabs(f));
            if (w != fin) {
              const double a1 = 1.0 + M_PI * w;
              rr = rr * (2.0 - a1 * rr); rr = rr * (2.0 - a1 * rr);
              kk = (M_PI * w) * rr; fin = w;
            }
          }
          y = y + kk * (c.target - y);
        }
        c.y = y; c.vol = y; c.finit = fin; c.k = kk;
      } else {
        if (!ok) { chan_step_phase(c, p, dur, inv_dur, max_filter_inc); --n; continue; }
        if (c.k != k_cached) { k_cached = c.k; lk = log1p_small(-c.k); }
        double q = exp((double)j * lk);
        c.y = c.target + (c.y - c.target) * q;
        c.vol = c.y;
      }
    }
    env_jump(c, p, j);
    if (gliding) { c.cur = new_cur; c.fres = new_fres; c.inc = new_fres; }
    c.theta += sum_inc;
    n -= j;
  }
  c.theta -= 2.0 * rint(0.5 * c.theta); // keep the angle in [-1, 1]
}

// hop start in the render kernel: phasor, per-sample rotation and anti-aliasing gain from (theta, inc)
__device__ inline void chan_from_phase(Chan &c) {
  double sn, cs;
  sincospi(c.theta, &sn, &cs);
  c.cre = cs; c.cim = sn;
  sincospi(c.inc, &sn, &cs);
  c.mre = cs; c.mim = sn;
  c.alias = (c.inc > 0. && c.inc <= 0.25) ? 1.0 : alias_mult(c.inc);
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
  c.theta = sd[SD_THETA * Ppad + l];
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
  sd[SD_THETA * Ppad + l] = c.theta;
  si[SI_ENV_STATE * Ppad + l] = c.state; si[SI_ENV_COUNTER * Ppad + l] = c.counter; si[SI_ENV_RELDELAY * Ppad + l] = c.reldelay;
  si[SI_ENV_AHDSTATE * Ppad + l] = c.ahdstate; si[SI_ENV_AHDCOUNTER * Ppad + l] = c.ahdcounter; si[SI_ENV_MINCH * Ppad + l] = c.minch;
  si[SI_ENV_NEXTMINCH * Ppad + l] = c.nextminch; si[SI_FLAGS * Ppad + l] = c.flags;
}
// A hop's start state as the propagate kernel hands it to the render kernel: the planes below SD_HOP_COUNT only (no gains when the render
// applies the centre gains after the sum).  The phasor, the rotation and the anti-aliasing gain are derived from (theta, inc) by
// chan_from_phase at the hop's start, vol is a copy of y, and the AHD state and the flags ride in the state word: 144 (128) instead of
// 200 bytes per sounding channel and hop.
__device__ inline void load_hop(Chan &c, const double *sd, const int *si, int Ppad, int l, bool mono) {
  c.value = sd[SD_ENV_VALUE * Ppad + l]; c.top = sd[SD_ENV_TOP * Ppad + l];
  c.from = sd[SD_CTRL_FROM * Ppad + l]; c.to = sd[SD_CTRL_TO * Ppad + l]; c.cur = sd[SD_CTRL_CUR * Ppad + l];
  c.C = sd[SD_CTRL_C * Ppad + l]; c.fres = sd[SD_CTRL_FRES * Ppad + l]; c.inc = sd[SD_OSC_INC * Ppad + l];
  c.target = sd[SD_VOL_TARGET * Ppad + l]; c.finit = sd[SD_VOL_FINIT * Ppad + l];
  c.k = sd[SD_VOL_K * Ppad + l]; c.y = sd[SD_VOL_Y * Ppad + l]; c.vol = c.y;
  c.theta = sd[SD_THETA * Ppad + l];
  if (!mono) { c.gL = sd[SD_GAIN_L * Ppad + l]; c.gR = sd[SD_GAIN_R * Ppad + l]; } else { c.gL = 0.0; c.gR = 0.0; }
  c.cre = 1.0; c.cim = 0.0; c.mre = 1.0; c.mim = 0.0; c.alias = 1.0; // chan_from_phase follows
  const int w = si[SI_ENV_STATE * Ppad + l];
  c.state = ra_hop_state(w); c.ahdstate = ((w >> 4) & 15) - 1; c.flags = w >> 8;
  c.counter = si[SI_ENV_COUNTER * Ppad + l]; c.reldelay = si[SI_ENV_RELDELAY * Ppad + l];
  c.ahdcounter = si[SI_ENV_AHDCOUNTER * Ppad + l]; c.minch = si[SI_ENV_MINCH * Ppad + l]; c.nextminch = si[SI_ENV_NEXTMINCH * Ppad + l];
}
__device__ inline void store_hop(const Chan &c, double *sd, int *si, int Ppad, int l, bool mono) {
  sd[SD_ENV_VALUE * Ppad + l] = c.value; sd[SD_ENV_TOP * Ppad + l] = c.top;
  sd[SD_CTRL_FROM * Ppad + l] = c.from; sd[SD_CTRL_TO * Ppad + l] = c.to; sd[SD_CTRL_CUR * Ppad + l] = c.cur;
  sd[SD_CTRL_C * Ppad + l] = c.C; sd[SD_CTRL_FRES * Ppad + l] = c.fres; sd[SD_OSC_INC * Ppad + l] = c.inc;
  sd[SD_VOL_TARGET * Ppad + l] = c.target; sd[SD_VOL_FINIT * Ppad + l] = c.finit;
  sd[SD_VOL_K * Ppad + l] = c.k; sd[SD_VOL_Y * Ppad + l] = c.y;
  sd[SD_THETA * Ppad + l] = c.theta;
  if (!mono) { sd[SD_GAIN_L * Ppad + l] = c.gL; sd[SD_GAIN_R * Ppad + l] = c.gR; }
  si[SI_ENV_STATE * Ppad + l] = ra_hop_word(c.state, c.ahdstate, c.flags);
  si[SI_ENV_COUNTER * Ppad + l] = c.counter; si[SI_ENV_RELDELAY * Ppad + l] = c.reldelay;
  si[SI_ENV_AHDCOUNTER * Ppad + l] = c.ahdcounter; si[SI_ENV_MINCH * Ppad + l] = c.minch; si[SI_ENV_NEXTMINCH * Ppad + l] = c.nextminch;
}
