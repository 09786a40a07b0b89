// K9 — gen.voice, SWEEP mode (first slice): voice::Impl_<..., Mode::SWEEP, ...> (reference include/gen.voice.h:1087-1190)
// = ImplCRTP::compute / onEvent (include/gen.crtp.h:350-622; SynchronizePhase::No, DefaultStartPhase::Zero) over
// VolumeAdjusted<SoundEngine<SWEEP, 2, ...>> (include/soundengine.h:240-1127), its ramps being
//   StereoPanned<VolumeAdjusted<Enveloped<FreqCtrl_<Mix<LoudnessVolumeAdjusted<SineOscillatorAlgo<double>>>, LogRamp<double>>, AHDSREnvelope>>>.
//
// The same skeleton as the other synth kernels: one workgroup per stream, lane = channel, samples in order in 16-sample tiles, the
// channels summed in channel order.  What SWEEP reduces the sound engine to (the CPU restatement keeps the general shape):
//   * the Markov chain of create_sweep (soundengine.h:697-720) is one certain transition whose LEAVE lambda builds the single ramp spec,
//     and both mersenne<> draws on the way are inert (min_exp == max_exp): initialize_sweep is deterministic arithmetic, done at note-on;
//   * of the engine's three ramps only one ever plays: get_ramps (:275-290) hands playNextSpec the LAST finished ramp, all three are
//     finished when a note starts, and a SWEEP note has one spec — a lane carries that one ramp;
//   * reference behaviour kept as it is: Mix::angleIncrements() never exceeds 0 (audioelement.h:1856-1865), so the volume filters see
//     the control's `from`; a note-off silences the element at once and leaves its ramp unfinished (soundengine.h:345-353), which
//     makes the channel unavailable for good (isEnvelopeFinished_internal, :457-467).
// Per sample a ramp costs a sincos (the increment moves every sample), a float pow (the equal-loudness volume, include/loudness.h:246-274)
// and three one-pole updates with re-derived coefficients: ~300 instructions on one wave per stream — correct first, not yet fast.
#include "ra_chan.cuh"
#include <algorithm>
#include <cstdlib>

extern __shared__ __align__(16) unsigned char ra_smem_voice[];

namespace {

struct VoiceChan {
  Chan c;   // the ramp: envelope, sine oscillator (cre, cim, mre, mim, alias, inc), its own volume filter (vol, target, finit, k, y, flags), gains
  // LogRamp (audioelement.h:2464-2572)
  double lr_from, lr_to, lr_cur, lr_dur, lr_C, lr_inv; // lr_inv = 1 / lr_dur
  // the loudness adjustment's BaseVolumeAdjusted
  double l_vol, l_target, l_finit, l_k, l_y;
  // the outer VolumeAdjusted<SoundEngine>
  double o_vol, o_target, o_finit, o_k, o_y;
  // the single ramp spec (RampSpecs, soundengine.h:1046-1120)
  double sp_from, sp_to, sp_dur, sp_cur, sp_C;
  double start_angle;
  int64_t noteid;
  int oddOn, rt_active, spec_it, spec_end, remaining_silence, l_has, o_has, loud_k; // loud_k: the loudness table interval the sweep is in
};

__device__ inline bool ramp_rt_active(const VoiceChan &v) { return rt_active(v.c); }
__device__ inline bool engine_finished(const VoiceChan &v) { return !(v.oddOn & 1) && !ramp_rt_active(v); } // isEnvelopeFinished_internal
__device__ inline bool elem_rt_active(const VoiceChan &v) { return v.rt_active && !engine_finished(v); }

// loudness::equal_loudness_volume_from_freq, include/loudness.h:246-274 over :157-186 (float arithmetic).  The reference finds the table
// interval by bisection on every call; a sweep moves through the 29 frequencies in order, so the interval of the previous sample is
// carried in `k` (freqs[k] <= freq < freqs[k + 1], clamped to 0 .. 27) and walked: the same interval, a step or two per note.  The final
// powf(10, x) runs as exp2f(x log2 10): a float ulp or two from the reference's pow on a gain (tolerance 1e-5).
__device__ inline float loudness_volume(const RaVoiceParams &p, const float *__restrict__ lf, const float *__restrict__ le, float freq, int &k) {
  k = min(max(k, 0), 27);
  while (k < 27 && freq >= lf[k + 1]) ++k;
  while (k > 0 && freq < lf[k]) --k;
  const int min_ = k, max_ = k + 1; // what the bisection of loudness.h:231-244 ends with
  int kk; float ratio;
  if (freq <= lf[min_]) { kk = min_; ratio = 1.f; }
  else if (freq >= lf[max_]) { kk = max_; ratio = 1.f; }
  else { kk = max_; ratio = (freq - lf[min_]) / (lf[max_] - lf[min_]); }
  const float db = (ratio == 1.f) ? le[kk] : ratio * le[kk] + (1.f - ratio) * le[kk - 1];
  if (db > p.max_db) return 1.f;
  return exp2f((p.log_ratio * (db - p.max_db) / 20.f) * 3.32192809488736f);
}

// FreqCtrl_::angleIncrements (audioelement.h:3117-3124) over Mix::angleIncrements (:1856-1865)
__device__ inline double ramp_angle_increments(const RaVoiceParams &p, const VoiceChan &v) {
  double minAngle = 0;
  if (p.sine_gain != 0.f) minAngle = fmin(minAngle, v.c.inc);
  if (minAngle != 0.) return minAngle;
  return v.lr_from;
}
__device__ inline double one_pole_k(double inc_abs) { // a / (1 + a), a = pi inc: v_rcp_f64 and two Newton steps instead of the IEEE division (an ulp)
  const double a = M_PI * inc_abs, d = 1.0 + a;
  double r = __builtin_amdgcn_rcp(d);
  r = r * fma(-d, r, 2.0);
  r = r * fma(-d, r, 2.0);
  return a * r;
}
__device__ inline void one_pole_volume(int &has, double &vol, double target, double &finit, double &k, double &y, double inc_abs) { // BaseVolumeAdjusted::step :1195-1216
  if (!has) {
    finit = inc_abs;
    k = one_pole_k(finit);
    y = target;
    has = 1;
  } else if (vol != target) {
    if (finit != inc_abs) { finit = inc_abs; k = one_pole_k(finit); }
  }
  y = y + k * (target - y);
  vol = y;
}

// SineOscillatorAlgo::setAngleIncrements (audioelement.h:2404-2411) on a sweep: the increment moves a little every sample, so the rotation
// (cos, sin)(pi f) follows it by the small angle pi (f - f_prev) (fifth-order series, |d| ~ 1e-4: exact to rounding) instead of a sincos
// per sample, and the anti-aliasing gain (:466-483) only divides where it is not 0 or 1.
__device__ inline void voice_osc_set_inc(Chan &c, double f) {
  if (c.inc == f) return;
  const double d = M_PI * (f - c.inc);
  if (c.inc != 0. && fabs(d) < 1e-2) {
    const double d2 = d * d;
    const double sd = d * fma(d2, fma(d2, 1.0 / 120.0, -1.0 / 6.0), 1.0), cd = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
    const double mr = fma(c.mre, cd, -(c.mim * sd)), mi = fma(c.mre, sd, c.mim * cd);
    c.mre = mr; c.mim = mi;
  } else {
    double sn, cs;
    sincos(M_PI * f, &sn, &cs);
    c.mre = cs; c.mim = sn;
  }
  c.inc = f;
  c.alias = (f == 0.) ? 1. : (f > 1.) ? 0. : (f < 0.25 && f > 0.) ? 1. : alias_mult(f);
}

// One step of a sounding element (ImplCRTP::compute, gen.crtp.h:352-371, for one channel): SoundEngine::step, the outer volume, imag.
__device__ inline void voice_step(VoiceChan &v, const RaVoiceParams &p, const float *__restrict__ lf, const float *__restrict__ le, const RaEnvParams &ep, double &vL,
                                  double &vR) {
 // ImplCRTP::compute, gen.crtp.h:352-371
  // ---- SoundEngine::step: orchestrate_algos (soundengine.h:741-773) ----
  if (v.spec_it != v.spec_end) {
    const bool has_cur = v.spec_it >= 0 && v.spec_it < v.spec_end; // (silenceAfter is true for a SWEEP spec)
    bool play = true;
    if (v.c.state == ENV_PRESSED) {
      const int remaining = v.c.reldelay; // countStepsToRelease
      if (remaining > 0) play = false;
      else if (remaining == 0 && has_cur) v.remaining_silence = 0; // articulative_pause_length: never set in SWEEP mode, taken as 0
    }
    if (play && has_cur && v.remaining_silence > 0) { --v.remaining_silence; play = false; }
    if (play && (v.oddOn & 1)) { // playNextSpec :775-801: a finished ramp exists (two of the three never play)
      ++v.spec_it;
      if (v.spec_it != v.spec_end) {
        v.lr_from = v.sp_from; v.lr_to = v.sp_to; v.lr_dur = v.sp_dur; v.lr_cur = v.sp_cur; v.lr_C = v.sp_C; v.lr_inv = 1.0 / v.sp_dur;
        chan_forget_past_signals(v.c); v.l_has = 0;     // new_ramp->forgetPastSignals()
        { double sn, cs; sincos(M_PI * v.start_angle, &sn, &cs); v.c.cre = cs; v.c.cim = sn; } // setStartAngle
        v.c.target = (double)1.f;                       // spec volume (setVolume never called in SWEEP: 1.f)
        v.c.state = ENV_SOON;                           // tryAcquire
        const int time_to_release = (int)(.5f + v.lr_dur) - p.xfade_len;
        env_on_key_pressed(v.c, 0);
        env_on_key_released(v.c, max(time_to_release, 0));
      }
    }
  }
  // ---- step_algos (:423-436): the ramp steps unless it is finished ----
  if (v.c.state != ENV_DONE2) {
    env_step(v.c, ep);
    double inc;
    { // LogRamp::do_step, audioelement.h:2522-2542
      if (v.lr_cur + (double).5f > v.lr_dur) { v.lr_cur = 0; const double tmp = v.lr_from; v.lr_from = v.lr_to; v.lr_to = tmp; }
      const double inv_dur = v.lr_inv; // the two divisions by the note's constant duration as multiplications (an ulp on the increment)
      inc = v.lr_from + (v.lr_to - v.lr_from) * (v.lr_cur * inv_dur);
      const double f = v.lr_from + (v.lr_to - v.lr_from) * ((v.lr_cur + (double).5f) * inv_dur);
      v.lr_cur += v.lr_C * f;
    }
    // Mix::setAngleIncrements -> LoudnessVolumeAdjusted::setAngleIncrements (:1291-1298) -> SineOscillatorAlgo::setAngleIncrements
    v.l_target = (double)loudness_volume(p, lf, le, (float)(inc * 0.5 * (double)p.sample_rate), v.loud_k);
    voice_osc_set_inc(v.c, inc);
    { // SineOscillatorAlgo::step
      const double re = v.c.cre * v.c.mre - v.c.cim * v.c.mim, im = v.c.cre * v.c.mim + v.c.cim * v.c.mre;
      const double g = 0.5 * (3.0 - (re * re + im * im));
      v.c.cre = re * g; v.c.cim = im * g;
    }
    one_pole_volume(v.l_has, v.l_vol, v.l_target, v.l_finit, v.l_k, v.l_y, fabs(v.c.inc));
    int has = (v.c.flags & FL_HAS_VOL) ? 1 : 0;
    one_pole_volume(has, v.c.vol, v.c.target, v.c.finit, v.c.k, v.c.y, fabs(ramp_angle_increments(p, v)));
    v.c.flags |= FL_HAS_VOL;
  }
  // ---- the outer BaseVolumeAdjusted::step: SoundEngine::angleIncrements (:407-421) is the sounding ramp's, else 0 ----
  const bool sounding = v.c.state == ENV_RELEASED || v.c.state == ENV_PRESSED;
  one_pole_volume(v.o_has, v.o_vol, v.o_target, v.o_finit, v.o_k, v.o_y, sounding ? fabs(ramp_angle_increments(p, v)) : 0.0);
  if (elem_rt_active(v) && sounding) {
    double sum = 0.f;                                             // Mix::imag :1867-1873
    sum += (double)p.sine_gain * (v.l_vol * (v.c.alias * v.c.cim));
    const double mono = v.c.vol * (sum * v.c.value);             // ramp: VolumeAdjusted over Enveloped
    vL = v.o_vol * (0.0 + v.c.gL * mono);                        // SoundEngine::imag(i): v{} += r.imag(i); BaseVolumeAdjusted::imag(i)
    vR = v.o_vol * (0.0 + v.c.gR * mono);
  }
      }

constexpr int VT = 16;

__global__ void __launch_bounds__(RA_MAX_CHANNELS) ra_voice_kernel(RaVoiceParams p, VoiceChan *__restrict__ state, const RaVoiceEv *__restrict__ ev,
                                                                   const int32_t *__restrict__ ev_begin, int32_t *__restrict__ dropped_out, int64_t t0,
                                                                   int64_t n, float *__restrict__ out) {
  const int s = blockIdx.x, l = threadIdx.x, Ppad = blockDim.x, P = p.P;
  double *tile = reinterpret_cast<double *>(ra_smem_voice);       // [Ppad][2 * VT + 1]
  int *xi = reinterpret_cast<int *>(tile + (size_t)Ppad * (2 * VT + 1));
  float *lf = reinterpret_cast<float *>(xi + 4), *le = lf + 29; // the loudness tables (a dynamically indexed kernel-argument array would live in scratch)
  if (l < 29) { lf[l] = p.loud_freqs[l]; le[l] = p.loud_elv[l]; }
  __syncthreads();
  VoiceChan v = state[(size_t)s * Ppad + l];
  const RaEnvParams ep = {p.envA, 0, 0, p.envR, 0.f}; // setAHDSR(AHDSR{xfade_len, LINEAR, 0, 0, LINEAR, xfade_len, LINEAR, 1.}), soundengine.h:790
  int e = ev_begin[s];
  const int e_end = ev_begin[s + 1];
  int dropped = 0;
  constexpr int64_t kNever = 0x7fffffffffffffffll;
  int64_t next_ev = (e < e_end) ? ev[e].sample : kNever;
  float *o = out + (size_t)s * n * 2;
  int64_t t = t0;
  const int64_t t_end = t0 + n;
  while (t < t_end) {
    if (next_ev <= t) { // Impl_::doProcessing (gen.voice.h:1159-1165): the events of this frame reach the synth before out.step
      const RaVoiceEv q = ev[e];
      ++e;
      next_ev = (e < e_end) ? ev[e].sample : kNever;
      if (l == 0) xi[0] = 0x7fffffff;
      __syncthreads();
      if (q.type == 0) { // onEvent(NoteOn), gen.crtp.h:389-449 + one-shot :455-527
        if (l < P && engine_finished(v)) atomicMin(&xi[0], l); // SoundEngine::tryAcquire :291-308
        __syncthreads();
        const int sel = xi[0];
        if (sel == 0x7fffffff) { if (l == 0) ++dropped; }
        else if (l == sel) {
          ++v.oddOn;
          v.noteid = q.noteid;
          v.remaining_silence = 0; v.start_angle = 0; v.o_has = 0;                 // forgetPastSignals: VolumeAdjusted + SoundEngine :314-317
          v.o_target = p.base_volume * (double)q.vel;                               // Element::baseVolume * velocity, gen.crtp.h:425
          // setupAudioElement (gen.voice.h:883-1003): parameters are uniform; the pan is per note
          {
            const float pan_f = p.random_pan ? hash_uniform_pm1(p.seed + (uint64_t)s, (uint64_t)q.noteid, 3) : p.pan;
            double pan = (double)pan_f;
            pan = fmin(pan, 1.); pan = fmax(pan, -1.);
            const double ang = (0.25 * M_PI) * (pan + 1.);
            v.c.gL = cos(ang); v.c.gR = sin(ang);                                   // stereo(), include/stereo.h:18-27
          }
          // engine.setAngleIncrements (soundengine.h:403-405) + initialize_sweep (:905-918) -> do_initialize (:1001-1044) -> node 0's LEAVE
          // lambda (:700-713); base_freq > 0 and n_frames > 0 were checked on the host
          v.spec_it = -1; v.spec_end = 1;
          v.sp_from = p.inc_low; v.sp_to = p.inc_high; v.sp_dur = p.n_frames; v.sp_cur = p.start_sample;
          v.sp_C = (v.sp_to == v.sp_from) ? (double)1.f : -log(v.sp_from / v.sp_to) / (v.sp_to - v.sp_from);
          // the one-shot: setPhase = setStartAngle(0) (DefaultStartPhase::Zero), onKeyPressed (soundengine.h:333-337)
          v.start_angle = 0;
          v.rt_active = 1;
        }
      } else { // NoteOff one-shot, gen.crtp.h:541-588
        if (l < P && v.noteid == q.noteid) atomicMin(&xi[0], l);
        __syncthreads();
        if (l == xi[0] && (v.oddOn & 1)) { // canHandleExplicitKeyReleaseNow = goOn(); onKeyReleased :345-353
          v.rt_active = 0;
          ++v.oddOn;                                         // stop(i)
          env_on_key_released(v.c, 0);                       // every ramp's onKeyReleased: only a KeyPressed one reacts
        }
      }
      __syncthreads();
      continue;
    }
    const int64_t seg_end = min(t_end, next_ev);
    if (!__syncthreads_or(l < P && elem_rt_active(v))) {
      for (int64_t r = (t - t0) * 2 + l; r < (seg_end - t0) * 2; r += blockDim.x) o[r] = 0.f;
      t = seg_end;
      continue;
    }
    const int m = (int)min((int64_t)VT, seg_end - t);
    double *mine = tile + (size_t)l * (2 * VT + 1);
    for (int i = 0; i < m; ++i) {
      double vL = 0.0, vR = 0.0;
      if (l < P && elem_rt_active(v)) voice_step(v, p, lf, le, ep, vL, vR);
      mine[i] = vL; mine[VT + i] = vR;
    }
    __syncthreads();
    if (l < 2 * m) { // buffer[2 i + j] += c.elem.imag(j), channels in index order; then the float cast of SimpleAudioOutContext::step
      const int fr = l >> 1, ch = l & 1;
      double acc = 0.0;
      for (int q = 0; q < P; ++q) acc += tile[(size_t)q * (2 * VT + 1) + ch * VT + fr];
      o[(t - t0 + fr) * 2 + ch] = (float)acc;
    }
    __syncthreads();
    t += m;
  }
  state[(size_t)s * Ppad + l] = v;
  if (l == 0 && dropped) dropped_out[s] += dropped;
}

__global__ void ra_voice_init_kernel(VoiceChan *state, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  VoiceChan v;
  memset(&v, 0, sizeof(v));
  chan_init(v.c);
  v.noteid = -1;
  state[i] = v;
}


// ---- one note per lane -------------------------------------------------------------------------------------------------
// A channel's state is made afresh by its note-on and the first step after it (forgetPastSignals, playNextSpec): a note's samples depend
// on nothing but the note.  Which channel a note-on gets — or that it is dropped — is integer bookkeeping (a channel is free again iff
// its note-off came after the ramp's envelope had finished, else it is lost for good), walked on the host (ra_capi.hip); here every
// note of the block is stepped by a lane of its own, from its stored state if it began in an earlier block, and added into a double
// accumulation buffer that a second kernel rounds to the float output.  The sum over a frame's notes is an atomic one: its association
// is not the channel order of the serial kernel (1e-16 relative, a float ulp now and then).
struct RaVoiceNote { int64_t begin; int64_t noteid; int32_t stream, slot, steps, fresh; float vel; int32_t store; }; // = ra_capi.hip: RaVoiceNoteHost

__global__ void __launch_bounds__(64) ra_voice_notes_kernel(RaVoiceParams p, const VoiceChan *__restrict__ slots_in, VoiceChan *__restrict__ slots_out, const RaVoiceNote *__restrict__ notes,
                                      int n_notes, int64_t t0, int64_t n, double *__restrict__ accum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = i < n_notes;
  RaVoiceNote q;
  if (have) q = notes[i]; else { q.steps = 0; q.slot = 0; q.fresh = 1; q.begin = t0; q.stream = 0; q.noteid = 0; q.vel = 0; q.store = 0; }
  const RaEnvParams ep = {p.envA, 0, 0, p.envR, 0.f};
  __shared__ float lf[29], le[29]; // (a dynamically indexed kernel-argument array would live in scratch)
  for (int j = threadIdx.x; j < 29; j += blockDim.x) { lf[j] = p.loud_freqs[j]; le[j] = p.loud_elv[j]; }
  __syncthreads();
  VoiceChan v;
  if (have && !q.fresh) v = slots_in[q.slot]; // (written by the previous block: the two sets alternate, so a successor on the same channel cannot overwrite it)
  else { // a free channel, then onEvent(NoteOn) and its one-shot exactly as in the serial kernel
    v = VoiceChan{};
    chan_init(v.c);
    ++v.oddOn;
    v.noteid = q.noteid;
    v.remaining_silence = 0; v.start_angle = 0; v.o_has = 0;
    v.o_target = p.base_volume * (double)q.vel;
    {
      const float pan_f = p.random_pan ? hash_uniform_pm1(p.seed + (uint64_t)q.stream, (uint64_t)q.noteid, 3) : p.pan;
      double pan = (double)pan_f;
      pan = fmin(pan, 1.); pan = fmax(pan, -1.);
      const double ang = (0.25 * M_PI) * (pan + 1.);
      v.c.gL = cos(ang); v.c.gR = sin(ang);
    }
    v.spec_it = -1; v.spec_end = 1;
    v.sp_from = p.inc_low; v.sp_to = p.inc_high; v.sp_dur = p.n_frames; v.sp_cur = p.start_sample;
    v.sp_C = (v.sp_to == v.sp_from) ? (double)1.f : -log(v.sp_from / v.sp_to) / (v.sp_to - v.sp_from);
    v.start_angle = 0;
    v.rt_active = 1;
  }
  // A lane's frames are consecutive in the accumulation buffer, but the lanes of a wave are far apart: one atomic per lane and frame
  // is 64 cache lines per instruction (29 of 75 ms per 256 synths x 10 s).  So a lane parks NS frames in LDS and the block then adds
  // every lane's run as one contiguous piece of 2 NS doubles.
  constexpr int NS = 16;
  __shared__ double stage[64][2 * NS + 1];
  __shared__ long long run_base[64];
  __shared__ int run_steps[64], max_steps;
  const int tl = threadIdx.x, nt = blockDim.x; // nt = 32 or 64
  if (tl == 0) max_steps = 0;
  __syncthreads();
  run_base[tl] = (long long)(((size_t)q.stream * n + (size_t)(q.begin - t0)) * 2);
  run_steps[tl] = q.steps;
  if (q.steps > 0) atomicMax(&max_steps, q.steps);
  __syncthreads();
  const int most = max_steps;
  for (int k0 = 0; k0 < most; k0 += NS) {
    for (int j = 0; j < NS; ++j) {
      double vL = 0.0, vR = 0.0;
      if (k0 + j < q.steps) voice_step(v, p, lf, le, ep, vL, vR);
      stage[tl][2 * j] = vL; stage[tl][2 * j + 1] = vR;
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      const int cnt = min(NS, run_steps[t] - k0); // frames of lane t in this piece (uniform over the block)
      if (cnt <= 0) continue;
      if (tl < 2 * cnt) {
        const double x = stage[t][tl];
        if (x != 0.0) atomicAdd(accum + run_base[t] + 2 * (long long)k0 + tl, x);
      }
    }
    __syncthreads();
  }
  if (have && q.store) slots_out[q.slot] = v; // (at most one note per channel goes on into the next block)
}

// the float cast of SimpleAudioOutContext::step, and the buffer is zero again for the next block
__global__ void ra_voice_cast_kernel(double *__restrict__ accum, float *__restrict__ out, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    out[i] = (float)accum[i];
    accum[i] = 0.0;
  }
}

// steps of the ramp's envelope from the note's first step until it is finished (EnvelopeDone2): what decides whether a channel is free again
__global__ void ra_voice_probe_kernel(RaVoiceParams p, int *out) {
  const RaEnvParams ep = {p.envA, 0, 0, p.envR, 0.f};
  Chan c;
  chan_init(c);
  chan_forget_past_signals(c);
  c.state = ENV_SOON;
  const int time_to_release = (int)(.5f + p.n_frames) - p.xfade_len;
  env_on_key_pressed(c, 0);
  env_on_key_released(c, max(time_to_release, 0));
  int k = 0;
  while (c.state != ENV_DONE2 && k < (1 << 30)) { env_step(c, ep); ++k; }
  *out = k;
}

} // namespace

size_t ra_voice_state_bytes() { return sizeof(VoiceChan); }
void ra_voice_launch_init(void *state, int count, hipStream_t st) {
  hipLaunchKernelGGL(ra_voice_init_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, static_cast<VoiceChan *>(state), count);
}
void ra_voice_launch(const RaVoiceParams &p, void *state, const RaVoiceEv *ev, const int32_t *ev_begin, int32_t *dropped, int n_streams, int64_t t0, int64_t n,
                     float *d_out, hipStream_t st) {
  if (n <= 0) return;
  const int Ppad = ((p.P + 63) / 64) * 64;
  const size_t lds = (size_t)Ppad * (2 * VT + 1) * sizeof(double) + 16 + 58 * sizeof(float);
  hipLaunchKernelGGL(ra_voice_kernel, dim3((unsigned)n_streams), dim3((unsigned)Ppad), lds, st, p, static_cast<VoiceChan *>(state), ev, ev_begin, dropped, t0, n, d_out);
}

size_t ra_voice_note_bytes() { return sizeof(RaVoiceNote); }
int ra_voice_probe_done2(const RaVoiceParams &p, int *d_scratch) {
  hipLaunchKernelGGL(ra_voice_probe_kernel, dim3(1), dim3(1), 0, nullptr, p, d_scratch);
  int k = -1;
  if (hipMemcpy(&k, d_scratch, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return k;
}
void ra_voice_launch_notes(const RaVoiceParams &p, const void *slots_in, void *slots_out, const void *notes, int n_notes, int n_streams, int64_t t0, int64_t n, double *accum, float *d_out,
                           hipStream_t st) {
  if (n <= 0) return;
  if (n_notes > 0) {
    // notes per wavefront (tuning aid): measured on 256 synths x 10 s, 64: 71.5 ms, 32: 69.3 ms, 16: 94.9 ms per step
    static const int per_wave = [] { const char *e = getenv("RA_VOICE_NOTES_PER_WAVE"); const int v = e ? atoi(e) : 32; return v == 64 ? 64 : 32; }();
    hipLaunchKernelGGL(ra_voice_notes_kernel, dim3((unsigned)((n_notes + per_wave - 1) / per_wave)), dim3(per_wave), 0, st, p, static_cast<const VoiceChan *>(slots_in),
                       static_cast<VoiceChan *>(slots_out), static_cast<const RaVoiceNote *>(notes), n_notes, t0, n, accum);
  }
  const size_t count = (size_t)n_streams * (size_t)n * 2;
  hipLaunchKernelGGL(ra_voice_cast_kernel, dim3((unsigned)std::min<size_t>((count + 255) / 256, 65536)), dim3(256), 0, st, accum, d_out, count);
}

