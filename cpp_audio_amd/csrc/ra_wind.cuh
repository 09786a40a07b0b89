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
//              Section s at time t needs section s - 1 at time t and itself at time t - 1.  A chain of at most 64 sections runs as scans over time
//              (ra_wind_scan_kernel: a one-pole with a coefficient per sample is an affine map of its state per sample); a longer one as a
//              systolic pipeline of waves (ra_wind_chain_kernel), a section per lane: in step tau lane l works on sample tau - l, its input handed
//              over by lane l - 1 through one DPP move, its coefficient read from a ring in LDS the wave fills 64 samples at a time.
//   mix        Mix::imag's sum in voice order, the envelope, the ramp's and the note's gains, the ramps of a stream summed in job order.
// Per chunk of a block: controls (+ sine) -> chains -> mix; the per-sample coefficients and voice outputs go through scratch in HBM; with short
// chains the controls of the next chunk run on a queue of their own beside this chunk's chains (ra_capi.hip: voice_wind_block).

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
// The window comes out of a stretch of |pink| the wave keeps in LDS (kWindWin samples from position `wbase` on) and refills when the iterator has moved past
// it: the iterator's next position depends on this batch, so a fetch from memory per batch would be a serial round trip (it was: ~1 of 1.6 us per batch).
constexpr int kWindWin = 512;
struct WindWin { float *v; int base; }; // base < 0: nothing loaded yet
__device__ inline float wind_iter_batch(RaWindIter &s, int kind, const double *__restrict__ pink, int len, int cnt, int lane, int *lds64, WindWin &win) {
  float w0, w1; // the window: sample lane and sample 64 + lane after the iterator's position
  {
    int rel = s.pos - win.base;
    rel = rel < 0 ? rel + len : rel;
    if (win.base < 0 || rel + 128 > kWindWin) { // (uniform) refill from the iterator's position on
      __syncthreads();
      for (int i = lane; i < kWindWin; i += 64) {
        int p = s.pos + i;
        p = p >= len ? p - len : p;
        win.v[i] = wind_abs(pink[p]);
      }
      win.base = s.pos;
      rel = 0;
      __syncthreads();
    }
    w0 = win.v[rel + lane]; w1 = win.v[rel + 64 + lane];
  }
  const float w64 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), 0));
  // piece q = lane: from prev_q to next_q
  const float below = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w0), 0x138, 0xf, 0xf, false)); // lane l - 1's (wave_shr:1)
  const float prev_q = lane == 0 ? s.prev : below, next_q = w0;
  const int rate_q = (prev_q > next_q) ? 1 : 3;
  const int len_q = max(1, ((lane == 0 ? s.n - s.slow : s.n) + rate_q - 1) / rate_q);
  // the step (1-based) of the piece's major: an inclusive scan of the lengths — rows of 16 lanes by DPP row shifts (lanes short of a source add 0), the rows'
  // totals through scalar registers (six rounds of ds_bpermute were ~700 cycles of dependent LDS round trips per batch on a lone wave)
  int end_q = len_q;
  end_q += __builtin_amdgcn_update_dpp(0, end_q, 0x111, 0xf, 0xf, true);
  end_q += __builtin_amdgcn_update_dpp(0, end_q, 0x112, 0xf, 0xf, true);
  end_q += __builtin_amdgcn_update_dpp(0, end_q, 0x114, 0xf, 0xf, true);
  end_q += __builtin_amdgcn_update_dpp(0, end_q, 0x118, 0xf, 0xf, true);
  {
    const int r0 = __builtin_amdgcn_readlane(end_q, 15), r1 = __builtin_amdgcn_readlane(end_q, 31), r2 = __builtin_amdgcn_readlane(end_q, 47);
    end_q += lane >= 48 ? r0 + r1 + r2 : (lane >= 32 ? r0 + r1 : (lane >= 16 ? r0 : 0));
  }
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
  __shared__ float windows[2][kWindWin];
  WindWin win0{windows[0], -1}, win1{windows[1], -1};
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
    RaWindIter c = fresh ? wind_iter_fresh(jobs[ji].pos[ic], wp.n_band, pink, true) : S->it[ic];
    RaWindIter w = fresh ? wind_iter_fresh(jobs[ji].pos[iw], wp.n_band, pink, true) : S->it[iw];
    double *k_lp = (role == 1 ? sc.k_bp_lp : sc.k_br_lp) + base, *k_hp = (role == 1 ? sc.k_bp_hp : sc.k_br_hp) + base;
    const bool fixed_width = wp.width_max == wp.width_min;
    for (int b0 = 0; b0 < cnt; b0 += 64) {
      const int c64 = min(64, cnt - b0);
      const float vc = wind_iter_batch(c, RA_ITP_LINEAR, pink, len, c64, lane, flags, win0);
      // (a width range of one value — five of the reference's programs —: min + |v| * 0 is the minimum whatever the width's iterator reads; it is not walked)
      const float vw = fixed_width ? 0.f : wind_iter_batch(w, RA_ITP_LINEAR, pink, len, c64, lane, flags, win1);
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
  RaWindIter lg = fresh ? wind_iter_fresh(jobs[ji].pos[0], wp.n_long, pink, true) : S->it[0];
  RaWindIter sh = fresh ? wind_iter_fresh(jobs[ji].pos[1], -1, pink, false) : S->it[1];
  double theta = fresh ? 0.0 : S->theta, y = fresh ? 0.0 : S->y;
  Ramp r;
  r.loud_k = fresh ? 14 : S->loud_k;
  r.lo_f = 1.f; r.hi_f = 0.f; r.inv_w = 0.f; r.lo_e = 0.f; r.hi_e = 0.f;
  loudness_seek(r, lf, le, lf[min(max(r.loud_k, 0), 27)]);
  double *p_sine = sc.p_sine + base;
  const bool short_term = wp.short_amplitude != 0.f; // 2^(0 x) = 1: the short-term iterator's walk decides nothing
  for (int b0 = 0; b0 < cnt; b0 += 64) {
    const int c64 = min(64, cnt - b0);
    const float vl = wind_iter_batch(lg, wp.interp_long, pink, len, c64, lane, flags, win0);
    const float long_inc = wind_center_inc(wp, vl);
    float inc_f = long_inc;
    if (short_term) {
      // SlowIter<PinkNoiseIter>: set_n_slow_steps (:2862-2877), ++, * — a serial walk, the same in every lane, over samples the lanes fetched beforehand (the
      // buffer moves on by at most two samples per step); lane i keeps step i's (slow_it, n_steps, prev, next) and the interpolations follow in parallel
      float b0v, b1v, b2v;
      {
        int p = sh.pos + lane;
        p = p >= len ? p - len : p;
        b0v = (float)pink[p];
        p += 64; p = p >= len ? p - len : p;
        b1v = (float)pink[p];
        p += 64; p = p >= len ? p - len : p;
        b2v = (float)pink[p];
      }
      auto at = [&](int rel) { // sample `rel` (uniform) after the batch's first position
        const int u = __builtin_amdgcn_readfirstlane(rel);
        const int a0 = __builtin_amdgcn_readlane(__float_as_int(b0v), u & 63), a1 = __builtin_amdgcn_readlane(__float_as_int(b1v), u & 63),
                  a2 = __builtin_amdgcn_readlane(__float_as_int(b2v), u & 63);
        return __int_as_float(u < 64 ? a0 : (u < 128 ? a1 : a2));
      };
      const int n_mine = (int)(1 + wp.short_rate / long_inc); // every sample's step count at once: the division is off the serial chain
      int rel = 0, m_slow = 0, m_n = 1, m_rel = 0;
      float m_prev = 0.f;
      for (int i = 0; i < c64; ++i) {
        const int n = __builtin_amdgcn_readlane(n_mine, i);
        if (n != sh.n) {
          if (sh.slow) {
            const float ratio = (sh.slow + .5f) / (float)sh.n;
            sh.slow = (int)(ratio * (float)n);
            if (sh.slow == n) { sh.slow = 0; sh.prev = at(rel); ++rel; }
          }
          sh.n = n;
        }
        ++sh.slow;
        if (sh.slow >= sh.n) { sh.slow = 0; sh.prev = at(rel); ++rel; }
        if (i == lane) { m_slow = sh.slow; m_n = sh.n; m_prev = sh.prev; m_rel = rel; }
      }
      sh.pos += rel;
      sh.pos = sh.pos >= len ? sh.pos - len : sh.pos;
      // the sample each step's iterator points at, gathered in one go (all lanes in every shuffle)
      const float g0 = __shfl(b0v, m_rel & 63), g1 = __shfl(b1v, m_rel & 63), g2 = __shfl(b2v, m_rel & 63);
      const float m_next = m_rel < 64 ? g0 : (m_rel < 128 ? g1 : g2);
      const float mine = wind_interp(RA_ITP_LINEAR, m_slow, m_n, m_prev, m_next);
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
static_assert(kWindLink >= kWindRing && 5 * kWindWaveLds * 8 <= 65536, "five waves (order 258: 516 sections, two per lane) within the default 64 KB of dynamic LDS");
__device__ inline void wind_ring_put(double *ring, int size, int i, double v) { ring[i] = v; if (i < 64) ring[i + size] = v; }
__device__ inline double wind_fma3(double a, double b, double c) { // a b + c as the three-address instruction
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// 64 steps of a wave in the steady state (every lane has a sample in every step); a lane holds M consecutive sections.  KIND: 0 = every section of the wave
// a low-pass (output y), 1 = every one a high-pass or a pass-through of the last wave (output x - y), 2 = both kinds in the wave (output fma(hm, x, -+y):
// exact either way).  The tap is lane 63: its outputs are collected by a second shift register running the other way (wave_shl: lane l takes lane l + 1's,
// lane 63 its own new output), so that after the 64 steps lane tt holds step tt's.
template <int M, int KIND>
__device__ inline double wind_chain_batch(const double *const (&pc)[M], const double *__restrict__ px0, double (&y)[M], double &xout, const double (&hm)[M], const int (&sgn)[M]) {
  double collected = 0.0;
  double c[M], x0 = px0[0];
#pragma unroll
  for (int j = 0; j < M; ++j) c[j] = pc[j][0];
#pragma unroll
  for (int tt = 0; tt < 64; ++tt) {
    double c_n[M];
#pragma unroll
    for (int j = 0; j < M; ++j) c_n[j] = pc[j][tt + 1]; // (one step ahead; entry 64 is read and never used)
    const double x0_n = px0[tt + 1];
    double x = wave_shr1_f64(x0, xout);
#pragma unroll
    for (int j = 0; j < M; ++j) {
      y[j] = y[j] + c[j] * (x - y[j]); // s = s + k (x - s), the operations of the restatement
      if (KIND == 0) x = y[j];
      else if (KIND == 1) x = x - y[j];
      else x = wind_fma3(hm[j], x, __hiloint2double(__double2hiint(y[j]) ^ sgn[j], __double2loint(y[j])));
    }
    xout = x;
    collected = wave_shl1_f64(xout, collected);
#pragma unroll
    for (int j = 0; j < M; ++j) c[j] = c_n[j];
    x0 = x0_n;
  }
  return collected;
}

// M sections per lane: 1 for a chain that fits one wave (the shortest dependent chain per step), 2 beyond (the two DPP moves in, the two out and the input
// read are shared by two sections: 11 -> 6.5 instructions per section and step)
template <int M>
__global__ void __launch_bounds__(320) ra_wind_chain_kernel(RaWindParams wp, const RaWindJob *__restrict__ jobs, const double *__restrict__ pink, double *__restrict__ fstates,
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
  double *zeros = wsm + (size_t)(W - 1) * kWindWaveLds + 3 * kWindRing; // kWindRing zeros, the "coefficients" of the pass-through sections: in the last wave's output ring, which it does not use
  const double *xin = w == 0 ? ring_x : link_out - kWindWaveLds; // lane 0's input: the noise, or what the wave before hands on
  const int xmask = w == 0 ? 127 : 255;
  // The chain ends at the last lane of the last wave, so that every wave's output leaves at lane 63; the places before its first section pass their input on
  // unchanged: a high-pass section whose coefficient and state are 0 does exactly that (y = 0 + 0 (x - 0) = 0, output x - 0).
  const bool last = w == W - 1;
  const int lead = 64 * M * W - L;
  int sidx[M], sgn[M];
  bool on[M];
  double y[M], hm[M];
  const double *rk[M];
  bool any_lp = false, any_hp = false;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    sidx[j] = (64 * w + l) * M + j - lead;
    on[j] = sidx[j] >= 0;
    const bool hp = !on[j] || sidx[j] >= ns;
    y[j] = (on[j] && !fresh) ? fs[sidx[j]] : 0.0;
    rk[j] = !on[j] ? zeros : (hp ? ring_b : ring_a);
    hm[j] = hp ? 1.0 : 0.0;
    sgn[j] = hp ? (int)0x80000000 : 0;
    any_lp |= !hp; any_hp |= hp;
  }
  const int kind = __ballot(any_hp) == 0ull ? 0 : (__ballot(any_lp) == 0ull ? 1 : 2);
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
    const double *pc[M]; // the lane's coefficients of step tau0 + tt: pc[j][tt]
#pragma unroll
    for (int j = 0; j < M; ++j) pc[j] = rk[j] + ((tau0 - l) & 127);
    const double *px0 = xin + (tau0 & xmask);     // lane 0's input of step tau0 + tt: px0[tt]
    if (tau0 >= 63 && tau0 + 64 <= cnt) { // every lane has a sample in every step of this batch
      double collected;
      if (kind == 0) collected = wind_chain_batch<M, 0>(pc, px0, y, xout, hm, sgn);
      else if (kind == 1) collected = wind_chain_batch<M, 1>(pc, px0, y, xout, hm, sgn);
      else collected = wind_chain_batch<M, 2>(pc, px0, y, xout, hm, sgn);
      const int t = tau0 + l - 63; // the sample lane l's collected value belongs to
      if (last) out[t] = collected;
      else wind_ring_put(link_out, 256, t & 255, collected);
    } else {
      const int steps = min(64, total - tau0);
      for (int tt = 0; tt < steps; ++tt) {
        const int t = tau0 + tt - l;
        const bool act = t >= 0 && t < cnt;
        double x = wave_shr1_f64(px0[tt], xout);
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const double yn = y[j] + pc[j][tt] * (x - y[j]);
          y[j] = act ? yn : y[j];
          x = fma(hm[j], x, __hiloint2double(__double2hiint(yn) ^ sgn[j], __double2loint(yn)));
        }
        xout = x;
        if (act && l == 63) { if (last) out[t] = xout; else wind_ring_put(link_out, 256, t & 255, xout); }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < M; ++j) if (on[j]) fs[sidx[j]] = y[j];
}

// ---- filter chains that fit one wave (<= 64 sections: band-pass orders <= 32, band-reject orders <= 64) — as scans over time instead.
// One section per lane would leave most of a lone wave's lanes idle (order 4: 8 of 64) and every step a dependent chain (measured: 200 cycles per sample).
// A one-pole with a coefficient per sample is still an affine map of its state per sample, y -> (1 - k) y + k x: a lane takes 16 consecutive samples,
// composes their maps (pass 1), an inclusive scan over the 64 lanes composes the lanes' (six rounds), and each lane replays its samples from the state it
// is handed (pass 2, the restatement's own y + k (x - y)) — section after section over a tile of 1024 samples held in registers.  The state a lane is
// handed went through another association than the sample-serial order: 1e-16 relative, where the wave pipeline above is bit-identical.
constexpr int kWindScanSamples = 16;
template <bool HP>
__device__ inline void wind_scan_section(double (&x)[kWindScanSamples], const double (&k)[kWindScanSamples], double *__restrict__ state, int lane) {
  double A = 1.0, B = 0.0;
#pragma unroll
  for (int j = 0; j < kWindScanSamples; ++j) { B = B + k[j] * (x[j] - B); A = fma(-k[j], A, A); }
#pragma unroll
  for (int r = 0; r < 6; ++r) { // (A, B) <- (A, B) o (A, B) of the lane 2^r below
    const double At = __shfl_up(A, 1 << r), Bt = __shfl_up(B, 1 << r);
    if (lane >= (1 << r)) { B = fma(A, Bt, B); A *= At; }
  }
  const double carry = *state; // the section's state before the tile
  double y = fma(wave_shr1_f64(1.0, A), carry, wave_shr1_f64(0.0, B)); // the state this lane's first sample sees
#pragma unroll
  for (int j = 0; j < kWindScanSamples; ++j) {
    y = y + k[j] * (x[j] - y);
    x[j] = HP ? x[j] - y : y;
  }
  __syncthreads();
  if (lane == 63) *state = y; // (lanes past the chunk's end hold identity maps: lane 63 ends at the state after the last sample)
  __syncthreads();
}
__global__ void __launch_bounds__(64) ra_wind_scan_kernel(RaWindParams wp, const RaWindJob *__restrict__ jobs, const double *__restrict__ pink, double *__restrict__ fstates,
                                                           RaWindScratch sc, int c0, int clen, int chain0) {
  __shared__ double st[64];
  constexpr int SC = kWindScanSamples;
  const int ji = blockIdx.x, chain = chain0 + blockIdx.y, l = threadIdx.x;
  const RaWindJob J = jobs[ji];
  const int lo = max(J.begin, c0), hi = min(J.begin + J.n, c0 + clen);
  if (hi <= lo) return;
  const int cnt = hi - lo, N = wp.order, len = wp.pink_len;
  const int L = chain == 0 ? 2 * N : N, ns = chain == 2 ? 0 : N;
  const size_t base = (size_t)ji * (size_t)sc.stride + (size_t)(lo - c0);
  const double *ka = (chain == 0 ? sc.k_bp_lp : sc.k_br_lp) + base, *kb = (chain == 0 ? sc.k_bp_hp : sc.k_br_hp) + base;
  double *out = (chain == 0 ? sc.p_bp : (chain == 1 ? sc.p_br_lp : sc.p_br_hp)) + base;
  double *fs = fstates + (size_t)J.slot * (size_t)(4 * N) + (chain == 0 ? 0 : (chain == 1 ? 2 * N : 3 * N));
  const int j_lo = J.j0 + (lo - J.begin);
  st[l] = (l < L && j_lo != 1) ? fs[l] : 0.0;
  __syncthreads();
  for (int tile0 = 0; tile0 < cnt; tile0 += 64 * SC) {
    const int t0 = tile0 + l * SC;
    double x[SC], a[SC], b[SC];
    int px = (int)(((long long)wp.pink_start + j_lo + t0) % len);
#pragma unroll
    for (int j = 0; j < SC; ++j) { // past the chunk's end: coefficient 0 (the identity map), input 0
      const bool in = t0 + j < cnt;
      x[j] = in ? pink[px] : 0.0;
      a[j] = (in && ns > 0) ? ka[t0 + j] : 0.0;
      b[j] = (in && ns < L) ? kb[t0 + j] : 0.0;
      px = px + 1 == len ? 0 : px + 1;
    }
    for (int s = 0; s < ns; ++s) wind_scan_section<false>(x, a, st + s, l);
    for (int s = ns; s < L; ++s) wind_scan_section<true>(x, b, st + s, l);
#pragma unroll
    for (int j = 0; j < SC; ++j) if (t0 + j < cnt) out[t0 + j] = x[j];
  }
  if (l < L) fs[l] = st[l];
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
