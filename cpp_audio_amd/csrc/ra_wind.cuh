// K10 — gen.voice WIND (SURVEY.md §8 f4's fourth mode; included by ra_voice.hip, inside its unnamed namespace: it shares the sine voice's helpers).
// SoundEngine<WIND> (reference include/soundengine.h:194-230, 721-735): a ramp is
//   FreqCtrl_< Mix< LowPassAlgo<pink>, AsymBandPassAlgo<pink>, AsymBandRejectAlgo<pink>, LoudnessVolumeAdjusted<sine> >,
//              ShortTermNoiseAdderCtrl< SoundEngineFreqCtrl<SlowIter<AbsIter<PinkNoiseIter>>>, SlowIter<PinkNoiseIter> > >
// and one spec plays for 2^30 steps: a note is ONE ramp that sounds until its note-off.  What it takes from the absent cpp.algorithms is decided
// in DESIGN.md §5 (parity unpinned): an order-n Filter = n one-poles in series, the band-pass' compensation = expt<VariableOrder = 0> = 1, the
// iterators' start positions = declared draws (the host hands them over in the job), the low-pass voice (never given a frequency) has gain 0.
//
// What depends on what, per sample of a ramp:
//   controls   five slow iterators over the pink buffer (audioelement.h:2846-3007) — the outer control's long-term one and, per band, a centre
//              and a width — are piecewise interpolations between consecutive |pink| samples whose pace (1 or 3 slow steps per sample) is fixed
//              per piece: the values of 64 consecutive samples are independent once the pieces that cover them are walked (a few per 64);
//              the exponentials, the widths' powers and the one-pole coefficients a / (1 + a) follow per sample, in parallel;
//              the outer control's short-term iterator changes its pace with every sample (soundengine.h:139-149): a serial walk, cheap, and
//              only there when its amplitude is not 0 (2^(0 x) = 1);
//   sine       phase = a running sum, loudness volume = a one-pole whose coefficient follows the increment: two scans over the 64 lanes;
//   filters    band-pass = n low-pass sections then n high-pass sections in series over the pink samples (HighPassAlgo<LowPassAlgo<pink>>,
//              audioelement.h:2163), band-reject = n low-pass sections + n high-pass sections, each over the pink samples (:2201-2203).
//              Section s at time t needs section s - 1 at time t and itself at time t - 1: a systolic array.  A wave holds the sections of one
//              chain, M consecutive ones per lane; in step tau lane l works on sample tau - l, its input handed over by lane l - 1 through one
//              DPP move, its two coefficients read from a ring in LDS the wave fills 64 samples at a time.
//   mix        Mix::imag's sum in voice order, the envelope, the ramp's and the note's gains, the ramps of a stream summed in job order.
// Three kernels per chunk of a block: controls (+ sine) -> chains -> mix; the per-sample coefficients and voice outputs go through scratch in HBM.

__device__ inline int wind_wrap(int i, int len) { return i >= len ? i - len : i; }
__device__ inline float wind_abs(double v) { return fabsf((float)v); } // AbsIter over BufferIter (the buffer's double narrowed to float)

// NormalizedInterpolation<>::get_unfiltered_value(slow_it, n_steps, prev, *it) as decided: itp::interpolate in double on the float operands
__device__ inline float wind_interp(int kind, int slow, int n, float prev, float next) {
  const double b = (double)prev, c = (double)next - (double)prev, t = (double)slow, d = (double)n;
  double v;
  if (kind == RA_ITP_LINEAR) v = c * t / d + b;
  else if (kind == RA_ITP_PROPORTIONAL_VALUE_DERIVATIVE) v = b * exp2((t / d) * log2((b + c) / b));
  else v = fma(c, ease01(kind, t / d), b);
  return (float)v;
}

// WindFreqIter<SlowIter<AbsIter<PinkNoiseIter>>, 3> (audioelement.h:2932-2966 over :2846-2927): `cnt` steps of the iterator; lane i < cnt returns
// the value read after step i + 1.  A step advances slow_it by 1 when the piece descends (prev > next) and by up to 3 when it ascends, and
// stops at the major step (slow_it reaches n_steps: prev <- next, the buffer moves on, slow_it <- 0).  `s` is the same in every lane.
__device__ inline float wind_iter_batch(RaWindIter &s, int kind, const double *__restrict__ pink, int len, int cnt, int lane) {
  float my_prev = 0.f, my_next = 0.f;
  int my_slow = 0;
  float next = wind_abs(pink[s.pos]);
  int done = 0;
  while (done < cnt) {
    const int rate = (s.prev > next) ? 1 : 3;
    const int to_major = max(1, (s.n - s.slow + rate - 1) / rate); // steps up to and including the major one
    const int take = min(to_major, cnt - done);
    const bool major = take == to_major;
    const int pos2 = wind_wrap(s.pos + 1, len);
    const float next2 = major ? wind_abs(pink[pos2]) : next;
    const int i = lane - done + 1;
    if (i >= 1 && i <= take) {
      if (i == to_major) { my_slow = 0; my_prev = next; my_next = next2; }
      else { my_slow = s.slow + rate * i; my_prev = s.prev; my_next = next; }
    }
    if (major) { s.prev = next; s.pos = pos2; next = next2; s.slow = 0; }
    else s.slow += rate * take;
    done += take;
  }
  return wind_interp(kind, my_slow, s.n, my_prev, my_next);
}
// a ramp's iterator as playNextSpec leaves it (soundengine.h:776-803): forgetPastSignals -> initializeForRun (position p0, prev taken there, the
// buffer moves on), then set_sample_rate draws the position again (p1); ABS: over AbsIter
__device__ inline RaWindIter wind_iter_fresh(const int32_t p[2], int n, const double *__restrict__ pink, bool abs_) {
  RaWindIter s;
  const float v = (float)pink[p[0]];
  s.prev = abs_ ? fabsf(v) : v;
  s.pos = p[1]; s.slow = 0; s.n = n;
  return s;
}
// SoundEngineFreqCtrl::step (soundengine.h:84-96), T = float; std::exp of a float = the double function rounded
__device__ inline float wind_center_inc(const RaWindParams &wp, float value) {
  const float v = value * wp.inv_approx_range;
  const float log_increment = wp.log_inc_min + (wp.log_inc_max - wp.log_inc_min) * v;
  return (float)exp((double)log_increment);
}
__device__ inline double wind_k(double w) { const double a = M_PI * w; return a / (1.0 + a); } // Filter::initWithAngleIncrement as decided

// ---- controls (+ the sine voice): one wave per (job, role); role 0 = the outer control and the sine, 1 = the band-pass' centre and width, 2 = the band-reject's
__global__ void __launch_bounds__(64) ra_wind_ctl_kernel(RaVoiceParams vp, RaWindParams wp, const RaWindJob *__restrict__ jobs, const double *__restrict__ pink,
                                                          RaWindState *__restrict__ states, RaWindScratch sc, int c0, int clen) {
  __shared__ float lf[29], le[29];
  load_tables(vp, lf, le);
  const int ji = blockIdx.x, role = blockIdx.y, lane = threadIdx.x;
  if ((role == 0 && wp.g_sine == 0.f) || (role == 1 && wp.g_bp == 0.f) || (role == 2 && wp.g_br == 0.f)) return; // Mix::imag: 0 * (a finite voice)
  const RaWindJob J = jobs[ji];
  const int lo = max(J.begin, c0), hi = min(J.begin + J.n, c0 + clen);
  if (hi <= lo) return;
  const int cnt = hi - lo, len = wp.pink_len;
  const int j_lo = J.j0 + (lo - J.begin); // the ramp's step index (1-based) of frame lo
  const bool fresh = j_lo == 1;
  RaWindState *S = states + J.slot;
  const size_t base = (size_t)ji * (size_t)sc.stride + (size_t)(lo - c0);
  if (role != 0) {
    // BandAlgo_::step (audioelement.h:2283-2300) under FreqCtrl_::step: the centre from the band's SoundEngineFreqCtrl, the width from its Ctrl
    const int ic = 2 * role, iw = 2 * role + 1;
    RaWindIter c = fresh ? wind_iter_fresh(J.pos[ic], wp.n_band, pink, true) : S->it[ic];
    RaWindIter w = fresh ? wind_iter_fresh(J.pos[iw], wp.n_band, pink, true) : S->it[iw];
    double *k_lp = (role == 1 ? sc.k_bp_lp : sc.k_br_lp) + base, *k_hp = (role == 1 ? sc.k_bp_hp : sc.k_br_hp) + base;
    for (int b0 = 0; b0 < cnt; b0 += 64) {
      const int c64 = min(64, cnt - b0);
      const float vc = wind_iter_batch(c, RA_ITP_LINEAR, pink, len, c64, lane);
      const float vw = wind_iter_batch(w, RA_ITP_LINEAR, pink, len, c64, lane);
      if (lane < c64) {
        const double increment = (double)wind_center_inc(wp, vc);
        const float at = wp.width_min + fabsf(vw) * (wp.width_max - wp.width_min); // range<float>::getAt
        const double width_factor = exp2((double)at);
        const double inv_width_factor = 1.0 / width_factor;
        const double low = increment * inv_width_factor, high = increment * width_factor;
        // band-pass: high-pass at `low`, low-pass at `high` (:2117-2118); band-reject: the other way round (:2220-2221)
        k_lp[b0 + lane] = wind_k(role == 1 ? high : low);
        k_hp[b0 + lane] = wind_k(role == 1 ? low : high);
      }
    }
    if (lane == 0) { S->it[ic] = c; S->it[iw] = w; }
    return;
  }
  // ShortTermNoiseAdderCtrl::step (soundengine.h:139-149), then LoudnessVolumeAdjusted<SineOscillatorAlgo> as the other modes run it
  RaWindIter lg = fresh ? wind_iter_fresh(J.pos[0], wp.n_long, pink, true) : S->it[0];
  RaWindIter sh = fresh ? wind_iter_fresh(J.pos[1], -1, pink, false) : S->it[1];
  double theta = fresh ? 0.0 : S->theta, y = fresh ? 0.0 : S->y;
  Ramp r;
  r.loud_k = fresh ? 14 : S->loud_k;
  r.lo_f = 1.f; r.hi_f = 0.f; r.inv_w = 0.f; r.lo_e = 0.f; r.hi_e = 0.f;
  loudness_seek(r, lf, le, lf[min(max(r.loud_k, 0), 27)]);
  double *p_sine = sc.p_sine + base;
  const bool short_term = wp.short_amplitude != 0.f; // 2^(0 x) = 1: the short-term iterator's walk decides nothing
  for (int b0 = 0; b0 < cnt; b0 += 64) {
    const int c64 = min(64, cnt - b0);
    const float vl = wind_iter_batch(lg, wp.interp_long, pink, len, c64, lane);
    const float long_inc = wind_center_inc(wp, vl);
    float inc_f = long_inc;
    if (short_term) {
      float mine = 0.f;
      for (int i = 0; i < c64; ++i) { // SlowIter<PinkNoiseIter>: set_n_slow_steps (:2862-2877), ++, * — the same in every lane
        const float li = __shfl(long_inc, i);
        const int n = (int)(1 + wp.short_rate / li);
        if (n != sh.n) {
          if (sh.slow) {
            const float ratio = (sh.slow + .5f) / (float)sh.n;
            sh.slow = (int)(ratio * (float)n);
            if (sh.slow == n) { sh.slow = 0; sh.prev = (float)pink[sh.pos]; sh.pos = wind_wrap(sh.pos + 1, len); }
          }
          sh.n = n;
        }
        ++sh.slow;
        if (sh.slow >= sh.n) { sh.slow = 0; sh.prev = (float)pink[sh.pos]; sh.pos = wind_wrap(sh.pos + 1, len); }
        const float nv = wind_interp(RA_ITP_LINEAR, sh.slow, sh.n, sh.prev, (float)pink[sh.pos]);
        if (i == lane) mine = nv;
      }
      inc_f = long_inc * (float)exp2((double)(mine * wp.short_amplitude));
    }
    const bool on = lane < c64;
    const double inc = on ? (double)inc_f : 0.0;
    double a = 1.0, b = 0.0;
    if (on) {
      const double tgt = (double)loudness_volume(r, vp, lf, le, (float)(inc * 0.5 * (double)vp.sample_rate));
      const double k = (j_lo + b0 + lane == 1) ? 1.0 : one_pole_k(fabs(inc)); // (the volume's filter starts at its first target)
      a = 1.0 - k; b = k * tgt;
    }
    double ph = inc;
#pragma unroll
    for (int s = 0; s < 6; ++s) { // inclusive scans over the lanes: the phase sum, and y -> a y + b composed
      const double pt = __shfl_up(ph, 1 << s), at = __shfl_up(a, 1 << s), bt = __shfl_up(b, 1 << s);
      if (lane >= (1 << s)) { ph += pt; b = fma(a, bt, b); a *= at; }
    }
    const double th = wrap2(theta + ph), yy = fma(a, y, b);
    if (on) p_sine[b0 + lane] = yy * (alias_gain(inc) * osc_imag(th));
    theta = wrap2(theta + __shfl(ph, 63));
    y = __shfl(yy, 63);
  }
  if (lane == 0) { S->it[0] = lg; S->it[1] = sh; S->theta = theta; S->y = y; }
  if (lane == min(63, (cnt - 1) & 63)) S->loud_k = r.loud_k; // (the table interval the last sample was in: a hint, any lane's would do)
}

// ---- filter chains: one wave per (job, chain); chain 0 = band-pass (n low-pass sections, then n high-pass), 1 = band-reject's low-pass, 2 = its high-pass
template <int M>
__global__ void __launch_bounds__(64) ra_wind_chain_kernel(RaWindParams wp, const RaWindJob *__restrict__ jobs, const double *__restrict__ pink, double *__restrict__ fstates,
                                                            RaWindScratch sc, int c0, int clen, int chain0) {
  __shared__ double ring_x[128], ring_a[128], ring_b[128];
  const int ji = blockIdx.x, chain = chain0 + blockIdx.y, l = threadIdx.x;
  const RaWindJob J = jobs[ji];
  const int lo = max(J.begin, c0), hi = min(J.begin + J.n, c0 + clen);
  if (hi <= lo) return;
  const int cnt = hi - lo, N = wp.order, len = wp.pink_len;
  const int L = chain == 0 ? 2 * N : N;            // sections of this chain
  const int ns = chain == 2 ? 0 : N;               // sections below ns are low-pass (coefficient a), the others high-pass (coefficient b)
  const size_t base = (size_t)ji * (size_t)sc.stride + (size_t)(lo - c0);
  const double *ka = (chain == 0 ? sc.k_bp_lp : sc.k_br_lp) + base, *kb = (chain == 0 ? sc.k_bp_hp : sc.k_br_hp) + base;
  double *out = (chain == 0 ? sc.p_bp : (chain == 1 ? sc.p_br_lp : sc.p_br_hp)) + base;
  double *fs = fstates + (size_t)J.slot * (size_t)(4 * N) + (chain == 0 ? 0 : (chain == 1 ? 2 * N : 3 * N));
  const int j_lo = J.j0 + (lo - J.begin);
  const bool fresh = j_lo == 1; // Filter::forgetPastSignals: the sections start at 0
  double y[M];
#pragma unroll
  for (int j = 0; j < M; ++j) { const int s = l * M + j; y[j] = (s < L && !fresh) ? fs[s] : 0.0; }
  const int out_lane = (L - 1) / M;
  const int total = cnt + out_lane;
  // the noise wrapper's index at step j: pink_start + j (soundBufferWrapperAlgo::step, audioelement.h:1568-1574)
  int px = (int)(((long long)wp.pink_start + j_lo + l) % len); // lane l stages sample tau0 + l
  double xout = 0.0; // what this lane handed on in the previous step
  for (int tau0 = 0; tau0 < total; tau0 += 64) {
    __syncthreads();
    {
      const int t = tau0 + l;
      if (t < cnt) { ring_x[t & 127] = pink[px]; ring_a[t & 127] = ka[t]; ring_b[t & 127] = kb[t]; }
      px = (int)(((long long)px + 64) % len);
    }
    __syncthreads();
    const int steps = min(64, total - tau0);
    for (int tt = 0; tt < steps; ++tt) {
      const int tau = tau0 + tt, t = tau - l;
      const bool act = t >= 0 && t < cnt;
      double x = wave_shr1_f64(0.0, xout);
      if (l == 0) x = ring_x[tau & 127];
      const double ca = ring_a[t & 127], cb = ring_b[t & 127];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int s = l * M + j;
        if (s < L) {
          const bool hp = s >= ns;
          const double k = hp ? cb : ca;
          const double yn = y[j] + k * (x - y[j]); // s = s + k (x - s), the operations of the restatement
          y[j] = act ? yn : y[j];
          x = hp ? x - yn : yn;
        }
      }
      xout = x;
      if (act && l == out_lane) out[t] = x;
    }
  }
#pragma unroll
  for (int j = 0; j < M; ++j) { const int s = l * M + j; if (s < L) fs[s] = y[j]; }
}

// ---- mix: one lane per (stream, frame) of the chunk
__global__ void __launch_bounds__(256) ra_wind_mix_kernel(RaWindParams wp, const RaWindJob *__restrict__ jobs, const int32_t *__restrict__ stream_begin, RaWindScratch sc, int c0,
                                                           int clen, int64_t n, float *__restrict__ out) {
  const int stream = blockIdx.y, f = c0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (f >= c0 + clen) return;
  double accL = 0.0, accR = 0.0;
  for (int ji = stream_begin[stream], je = stream_begin[stream + 1]; ji < je; ++ji) {
    const RaWindJob J = jobs[ji];
    if (f < J.begin || f >= J.begin + J.n) continue;
    const size_t at = (size_t)ji * (size_t)sc.stride + (size_t)(f - c0);
    double sum = 0.0; // Mix::imag (audioelement.h:1867-1873): gains[i] * voice_i in voice order; the band-pass' compensation is 1
    if (wp.g_bp != 0.f) sum += wp.g_bp * sc.p_bp[at];
    if (wp.g_br != 0.f) sum += wp.g_br * (sc.p_br_lp[at] + sc.p_br_hp[at]);
    if (wp.g_sine != 0.f) sum += wp.g_sine * sc.p_sine[at];
    const int j = J.j0 + (f - J.begin);
    const double inv_A = 1.0 / (double)J.A, inv_R = 1.0 / (double)J.R;
    const double top = (J.T < J.A) ? (double)J.T * inv_A : 1.0;
    const double env = (j <= J.T) ? ((j < J.A) ? (double)j * inv_A : 1.0) : fma(-top, (double)(j - J.T) * inv_R, top);
    const double mono = J.vol * (sum * env);
    accL += J.fL * mono; accR += J.fR * mono;
  }
  float2 *o = reinterpret_cast<float2 *>(out) + ((size_t)stream * (size_t)n + (size_t)f);
  *o = make_float2((float)accL, (float)accR);
}
