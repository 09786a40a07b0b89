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

// WindFreqIter<SlowIter<AbsIter<PinkNoiseIter>>, 3> (audioelement.h:2932-2966 over :2846-2927): `cnt` <= 64 steps of the iterator; lane i < cnt returns
// the value read after step i + 1.  A step advances slow_it by 1 when the piece descends (prev > next) and by up to 3 when it ascends, and
// stops at the major step (slow_it reaches n_steps: prev <- next, the buffer moves on, slow_it <- 0).  `s` is the same in every lane.
// No walk: a piece that starts at a major step lasts ceil(n / rate) steps, and its rate depends on the two buffer samples it runs between — so
// lane q holds the piece between window samples q - 1 and q (lane 0: the piece under way), an inclusive scan of the lengths says in which step each
// piece ends, the steps with a major become a 64-bit mask, and a sample's piece is the number of majors before it (one mbcnt).
// (A piece-by-piece walk cost ~25 dependent instructions per piece — with pieces of four samples most of the controls' time.)
__device__ inline float wind_iter_batch(RaWindIter &s, int kind, const double *__restrict__ pink, int len, int cnt, int lane, int *lds64) {
  float w0, w1; // the window: sample lane and sample 64 + lane after the iterator's position
  {
    int p0 = s.pos + lane;
    p0 = p0 >= len ? p0 - len : p0;
    int p1 = p0 + 64;
    p1 = p1 >= len ? p1 - len : p1;
    w0 = wind_abs(pink[p0]); w1 = wind_abs(pink[p1]);
  }
  const float w64 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), 0));
  // piece q = lane: from prev_q to next_q
  const float below = __shfl_up(w0, 1);
  const float prev_q = lane == 0 ? s.prev : below, next_q = w0;
  const int rate_q = (prev_q > next_q) ? 1 : 3;
  const int len_q = max(1, ((lane == 0 ? s.n - s.slow : s.n) + rate_q - 1) / rate_q);
  int end_q = len_q; // the step (1-based) of the piece's major
#pragma unroll
  for (int k = 0; k < 6; ++k) { const int t = __shfl_up(end_q, 1 << k); if (lane >= (1 << k)) end_q += t; }
  __syncthreads();
  lds64[lane] = 0;
  __syncthreads();
  if (end_q <= 64) lds64[end_q - 1] = 1;
  __syncthreads();
  const unsigned long long majors = __ballot(lds64[lane] != 0); // bit j - 1: step j ends a piece
  const int q = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(majors >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)majors, 0u)); // the piece of step lane + 1
  const bool major = (majors >> lane) & 1ull;
  // (every shuffle with all lanes in: a lane that sits out a branch reads as 0 to the others)
  const int end_sh = __shfl(end_q, max(q - 1, 0));
  const int end_before = q > 0 ? end_sh : 0;
  const float pq = __shfl(prev_q, q), nq = __shfl(next_q, q);
  const float after_sh = __shfl(w0, (q + 1) & 63);
  const float after = (q + 1 < 64) ? after_sh : w64; // window sample q + 1: the piece after q runs to it
  const int rq = (pq > nq) ? 1 : 3;
  const int slow = major ? 0 : (q == 0 ? s.slow : 0) + rq * (lane + 1 - end_before);
  const float my_prev = major ? nq : pq, my_next = major ? after : nq;
  // the state after step cnt: what lane cnt - 1 found
  const int lc = cnt - 1;
  const int q_c = __builtin_amdgcn_readlane(q, lc), major_c = __builtin_amdgcn_readlane((int)major, lc), slow_c = __builtin_amdgcn_readlane(slow, lc);
  const float prev_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_prev), lc));
  s.prev = prev_c; s.slow = slow_c;
  s.pos += q_c + (major_c ? 1 : 0);
  s.pos = s.pos >= len ? s.pos - len : s.pos;
  return wind_interp(kind, slow, s.n, my_prev, my_next);
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
  __shared__ int flags[64];
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
      const float vc = wind_iter_batch(c, RA_ITP_LINEAR, pink, len, c64, lane, flags);
      const float vw = wind_iter_batch(w, RA_ITP_LINEAR, pink, len, c64, lane, flags);
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
    const float vl = wind_iter_batch(lg, wp.interp_long, pink, len, c64, lane, flags);
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

// ---- filter chains: one workgroup per (job, chain); chain 0 = band-pass (n low-pass sections, then n high-pass), 1 = band-reject's low-pass, 2 = its high-pass.
// A lane holds ONE section; a chain of more than 64 sections takes several waves, wave w the sections 64 w .. 64 w + 63: its lane 63 writes what
// lane 0 of wave w + 1 reads into a ring in LDS, and wave w + 1 runs two batches of 64 steps behind wave w (lane 63 of a wave finishes a sample 63
// steps after lane 0 started it), one workgroup barrier per batch.  (r04's first version kept several sections per lane in one wave: a step's
// instructions then form one dependent chain on a SIMD with nothing else to issue — 460 cycles per step at three sections per lane.)
// A step of the steady state: two DPP moves for the input, subtract / multiply / add, the output (nothing for a wave of low-pass sections, a subtraction
// for a wave of high-pass sections), two LDS reads at immediate offsets, and two DPP moves of a second shift register that collects lane 63's outputs
// for one store per batch (a readlane / writelane pair per word cost twice that and SGPR hazards).
// LDS per wave, in doubles: two coefficient rings and the input ring (wave 0: the noise) of 128 samples, the output ring of 256 (a wave reads its input
// two batches after the wave before wrote it); every ring is followed by a copy of its first 64 entries, so that the 65 entries a lane reads during
// a batch are contiguous wherever they start: one address per batch and immediate offsets, instead of three address instructions per read.
constexpr int kWindRing = 128 + 64, kWindLink = 256 + 64, kWindWaveLds = 3 * kWindRing + kWindLink;
static_assert(kWindLink >= kWindRing && 9 * kWindWaveLds * 8 <= 65536, "nine waves (order 258) within the default 64 KB of dynamic LDS");
__device__ inline void wind_ring_put(double *ring, int size, int i, double v) { ring[i] = v; if (i < 64) ring[i + size] = v; }
__device__ inline double wind_fma3(double a, double b, double c) { // a b + c as the three-address instruction
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// 64 steps of a wave in the steady state (every lane has a sample in every step).  KIND: 0 = every lane a low-pass section (output y), 1 = every lane a
// high-pass section or one of the last wave's leading pass-through lanes (output x - y), 2 = both kinds in the wave (output fma(hm, x, -+y): exact either way).
// The tap is lane 63: its outputs are collected by a second shift register running the other way (wave_shl: lane l takes lane l + 1's, lane 63 its own
// new output), so that after the 64 steps lane tt holds step tt's.
template <int KIND>
__device__ inline double wind_chain_batch(const double *__restrict__ pc, const double *__restrict__ px0, double &y, double &xout, double hm, int sgn) {
  double collected = 0.0;
  double c = pc[0], x0 = px0[0];
#pragma unroll
  for (int tt = 0; tt < 64; ++tt) {
    const double c_n = pc[tt + 1], x0_n = px0[tt + 1]; // (one step ahead; entry 64 is read and never used)
    const double x = wave_shr1_f64(x0, xout);
    y = y + c * (x - y); // s = s + k (x - s), the operations of the restatement
    if (KIND == 0) xout = y;
    else if (KIND == 1) xout = x - y;
    else xout = wind_fma3(hm, x, __hiloint2double(__double2hiint(y) ^ sgn, __double2loint(y)));
    collected = wave_shl1_f64(xout, collected);
    c = c_n; x0 = x0_n;
  }
  return collected;
}

__global__ void __launch_bounds__(576) ra_wind_chain_kernel(RaWindParams wp, const RaWindJob *__restrict__ jobs, const double *__restrict__ pink, double *__restrict__ fstates,
                                                             RaWindScratch sc, int c0, int clen, int chain0) {
  double *wsm = reinterpret_cast<double *>(ra_smem_voice);
  const int ji = blockIdx.x, chain = chain0 + blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63, W = blockDim.x >> 6;
  const RaWindJob J = jobs[ji];
  const int lo = max(J.begin, c0), hi = min(J.begin + J.n, c0 + clen);
  if (hi <= lo) return; // (the whole workgroup)
  const int cnt = hi - lo, N = wp.order, len = wp.pink_len;
  const int L = chain == 0 ? 2 * N : N;            // sections of this chain
  const int ns = chain == 2 ? 0 : N;               // sections below ns are low-pass (coefficient a), the others high-pass (coefficient b)
  const size_t base = (size_t)ji * (size_t)sc.stride + (size_t)(lo - c0);
  const double *ka = (chain == 0 ? sc.k_bp_lp : sc.k_br_lp) + base, *kb = (chain == 0 ? sc.k_bp_hp : sc.k_br_hp) + base;
  double *out = (chain == 0 ? sc.p_bp : (chain == 1 ? sc.p_br_lp : sc.p_br_hp)) + base;
  double *fs = fstates + (size_t)J.slot * (size_t)(4 * N) + (chain == 0 ? 0 : (chain == 1 ? 2 * N : 3 * N));
  const int j_lo = J.j0 + (lo - J.begin);
  const bool fresh = j_lo == 1; // Filter::forgetPastSignals: the sections start at 0
  double *ring_a = wsm + (size_t)w * kWindWaveLds, *ring_b = ring_a + kWindRing, *ring_x = ring_b + kWindRing, *link_out = ring_x + kWindRing;
  double *zeros = wsm + (size_t)(W - 1) * kWindWaveLds + 3 * kWindRing; // kWindRing zeros, the "coefficients" of the pass-through lanes: in the last wave's output ring, which it does not use
  const double *xin = w == 0 ? ring_x : link_out - kWindWaveLds; // lane 0's input: the noise, or what the wave before hands on
  const int xmask = w == 0 ? 127 : 255;
  // The last wave holds its sections in its LAST lanes, so that every wave's output leaves at lane 63; the lanes before them pass their input on unchanged:
  // a high-pass section whose coefficient and state are 0 does exactly that (y = 0 + 0 (x - 0) = 0, output x - 0).
  const bool last = w == W - 1;
  const int lead = last ? 64 * W - L : 0;
  const int sidx = 64 * w + l - lead;
  const bool on = l >= lead, hp = !on || sidx >= ns;
  double y = (on && !fresh) ? fs[sidx] : 0.0;
  const double *rk = !on ? zeros : (hp ? ring_b : ring_a);
  const double hm = hp ? 1.0 : 0.0;
  const int sgn = hp ? (int)0x80000000 : 0;
  const int kind = __ballot(hp) == 0ull ? 0 : (__ballot(!hp) == 0ull ? 1 : 2);
  const int total = cnt + 63, nb = (total + 63) / 64;
  // the noise wrapper's index at step j: pink_start + j (soundBufferWrapperAlgo::step, audioelement.h:1568-1574); lane l of wave 0 stages sample tau0 + l
  int px = (int)(((long long)wp.pink_start + j_lo + l) % len);
  double xout = 0.0; // what this lane handed on in the previous step
  if (last) for (int i = l; i < kWindRing; i += 64) zeros[i] = 0.0;
  for (int b = 0; b < nb + 2 * (W - 1); ++b) {
    const int bl = b - 2 * w;
    const bool mine = bl >= 0 && bl < nb;
    const int tau0 = 64 * bl;
    if (mine) {
      const int t = tau0 + l;
      if (t < cnt) {
        wind_ring_put(ring_a, 128, t & 127, ka[t]);
        wind_ring_put(ring_b, 128, t & 127, kb[t]);
        if (w == 0) wind_ring_put(ring_x, 128, t & 127, pink[px]);
      }
      px = (int)(((long long)px + 64) % len);
    }
    __syncthreads(); // this batch's rings are filled; what the wave before wrote in the last batch is visible
    if (!mine) continue;
    const double *pc = rk + ((tau0 - l) & 127);   // the lane's coefficient of step tau0 + tt: pc[tt]
    const double *px0 = xin + (tau0 & xmask);     // lane 0's input of step tau0 + tt: px0[tt]
    if (tau0 >= 63 && tau0 + 64 <= cnt) { // every lane has a sample in every step of this batch
      double collected;
      if (kind == 0) collected = wind_chain_batch<0>(pc, px0, y, xout, hm, sgn);
      else if (kind == 1) collected = wind_chain_batch<1>(pc, px0, y, xout, hm, sgn);
      else collected = wind_chain_batch<2>(pc, px0, y, xout, hm, sgn);
      const int t = tau0 + l - 63; // the sample lane l's collected value belongs to
      if (last) out[t] = collected;
      else wind_ring_put(link_out, 256, t & 255, collected);
    } else {
      const int steps = min(64, total - tau0);
      for (int tt = 0; tt < steps; ++tt) {
        const int t = tau0 + tt - l;
        const bool act = t >= 0 && t < cnt;
        const double c = pc[tt], x0 = px0[tt];
        const double x = wave_shr1_f64(x0, xout);
        const double yn = y + c * (x - y);
        y = act ? yn : y;
        xout = fma(hm, x, __hiloint2double(__double2hiint(yn) ^ sgn, __double2loint(yn)));
        if (act && l == 63) { if (last) out[t] = xout; else wind_ring_put(link_out, 256, t & 255, xout); }
      }
    }
  }
  if (on) fs[sidx] = y;
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
