// K1 — analysis: one workgroup per (stream, analysis frame).
//
// Replaces, for every frame at once, PeriodicFFT::onFullBuffer -> SqMagFftOperation::operator()
// (reference source/rt.resynth.lib.periodicfft.cpp:136-180, 301-312: window * x, zero pad, real FFT, |X|^2),
// extractLocalMaxFreqsMags (call site source/rt.resynth.lib.cpp:1592-1595) and the stateless half of
// RtResynth::step (lib.cpp:1676-1708: frequencies_to_pitches, aggregate, reduce, shift, harmonize,
// autotune, shift, harmonize) plus the loudness sort key of lib.cpp:1717-1727.
//
// The window never leaves the chip: N contiguous floats are read once (coalesced), the real FFT runs
// as an M = L/2 point complex FFT in LDS (fp64, like the reference), |X|^2 overwrites the same LDS slots,
// peaks are compacted in bin order and only the short autotuned pitch list goes back to HBM.
#include "ra_fft.cuh"
#include <hip/hip_ext.h>
#include "ra_math.cuh"
#include "ra_lane.cuh"
#include <algorithm>
#include <cstdlib>

extern __shared__ __align__(16) unsigned char ra_smem[];

__host__ __device__ inline int ra_list_capacity(const RaParams &p) { return (p.harm_pre != 0.f || p.harm_post != 0.f) ? 2 * p.maxr : p.maxr; }
// The two reduced lists [cap] live INSIDE the FFT buffer when they fit its M slots (r03): by the time they are written (step 5c) the spectrum
// and the peak list in it have been consumed — a thread first takes its (at most four) reduced items into registers, a barrier retires
// every reader of the buffer, then the items are stored.  An analysis workgroup then holds the FFT buffer and ~1.5 KB (51 -> 34 KB at
// N = 4096 with the default capacity of 512): beside the renderer's waves a CU takes a second one (measured with the kernel compiled for
// 128 VGPRs: configs[1] 13.9 -> 13.2 ms per step; either change alone: nothing).  Otherwise (small windows, huge capacities) the lists follow
// the buffer as before.
__host__ __device__ inline bool ra_lists_in_fft_buffer(int M, int cap, int T) { return 2 * cap <= M && ((M >> 1) + T - 1) / T <= 4 && M >= 512; }
// ints of the small area behind the FFT buffer: the block scans (T + 32), then from step 6 on the published scalars (16), the merge's scan
// (T + 32) and — when the lists are in the buffer — the loudness tables (29 doubles + 29 floats)
__host__ __device__ inline int ra_aux_ints(int T) { return 16 + (T + 32) + 96; }

// tuning aid: -DRA_X_STOP=n ends the analysis kernel after phase n (instruction-count ablations with scripts/pmc_probe.sh); 0 = off
#ifndef RA_X_STOP
#define RA_X_STOP 0
#endif
#define RA_STOP_AFTER(n) do { if (RA_X_STOP == (n)) return; } while (0)

namespace {

__device__ inline double sqmag_to_db(double x) { return 10.0 * ra_log10(x > 1e-300 ? x : 1e-300); }

// canonical lower_bound on .x of a double2 list (find_closest_pitch, rt.resynth.lib.autotune.cpp:189-218)
__device__ inline int find_closest_x(const double2 *v, int n, double pitch) {
  if (n == 0) return -1;
  int first = 0, len = n;
  while (len > 0) {
    int half = len >> 1, mid = first + half;
    if (v[mid].x < pitch) { first = mid + 1; len = len - half - 1; } else len = half;
  }
  int lb = first;
  if (lb == n) return lb - 1;
  if (lb == 0) return 0;
  float dist_lb = fabsf((float)(pitch - v[lb].x));
  float dist_prev = fabsf((float)(pitch - v[lb - 1].x));
  return dist_lb < dist_prev ? lb : lb - 1;
}
__device__ inline int find_closest_d(const double *v, int n, double pitch) {
  if (n == 0) return -1;
  int first = 0, len = n;
  while (len > 0) {
    int half = len >> 1, mid = first + half;
    if (v[mid] < pitch) { first = mid + 1; len = len - half - 1; } else len = half;
  }
  int lb = first;
  if (lb == n) return lb - 1;
  if (lb == 0) return 0;
  float dist_lb = fabsf((float)(pitch - v[lb]));
  float dist_prev = fabsf((float)(pitch - v[lb - 1]));
  return dist_lb < dist_prev ? lb : lb - 1;
}

// shift_pitches, rt.resynth.lib.algo.cpp:307-316
__device__ inline void shift_list(double2 *l, int n, float shift) {
  if (!shift) return;
  for (int i = 0; i < n; ++i) l[i].x += shift;
}

// harmonize_pitches, rt.resynth.lib.algo.cpp:318-371.  list has room for 2n entries; work likewise.
__device__ inline int harmonize_list(double2 *&list, double2 *&work, int n, int cap, float amount, int *err) {
  if (!amount) return n;
  int sz = n, total = n;
  for (int i = 0; i < sz; ++i) {
    double2 p = list[i];
    double harmonized = p.x + amount;
    int c = find_closest_x(list, total, harmonized);
    float dist = fabsf((float)(list[c].x - harmonized));
    if (dist <= 0.0001f) list[c].y += p.y;
    else if (total < cap) list[total++] = make_double2(harmonized, p.y);
    else *err = 1;
  }
  int a = 0, b = sz, o = 0;
  while (a < sz && b < total) {
    if (list[b].x < list[a].x) work[o++] = list[b++]; else work[o++] = list[a++];
  }
  while (a < sz) work[o++] = list[a++];
  while (b < total) work[o++] = list[b++];
  double2 *t = list; list = work; work = t;
  return total;
}

// mkAutotuneFunction, lib.cpp:1761-1873.  returns false for "no pitch".
__device__ inline bool autotune_fn(const RaParams &p, const double *allowed, double pitch, double &out) {
  if (!p.use_autotune) { out = pitch; return true; }
  if (p.autotune_type == RA_AUTOTUNE_SCALE) { // MusicalScalePitches::closest_pitch, autotune.cpp:95-123
    double half_tones_dist = pitch - p.scale_root;
    double octave_dist = half_tones_dist / 12;
    int octaves_translation = (octave_dist >= 0.) ? (int)octave_dist : (int)octave_dist - 1;
    double translated = pitch - (octaves_translation * 12);
    double rel = translated - p.scale_root;
    int i = 0;
    for (; i < 7; ++i) if (rel < p.scale_equi[i]) break;
    out = pitch - (rel - p.scale_pitches[i]);
    return true;
  }
  int i = find_closest_d(allowed, p.n_allowed, pitch);
  if (i < 0) return false;
  out = allowed[i];
  return true;
}

// equal_loudness_volume_db(loudness::pitches, pitch, idx(60 phon)), include/loudness.h:159-186, 231-244
__device__ inline float equal_loudness_db(const double *arr, const float *elv, double value) {
  int min_ = 0, max_ = 28;
  do {
    int center = (min_ + max_) / 2;
    if (value < arr[center]) max_ = center; else min_ = center;
  } while (max_ - min_ >= 2);
  int i; float ratio;
  if (value <= arr[min_]) { i = min_; ratio = 1.f; }
  else if (value >= arr[max_]) { i = max_; ratio = 1.f; }
  else { i = max_; ratio = (float)((value - arr[min_]) / (arr[max_] - arr[min_])); }
  if (ratio == 1.f) return elv[i];
  return ratio * elv[i] + (1.f - ratio) * elv[i - 1];
}

// exclusive prefix sum of one int per thread over the T-thread workgroup: scan[tid] = sum of v over lower threads, scan[T] = total
// (wave-level shuffles + T / 64 wave totals; scan needs T + 1 + T / 64 ints)
template <int T>
__device__ inline void block_exclusive_scan(int *scan, int v, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o); if (lane >= o) x += y; }
  if (lane == 63) scan[T + 1 + w] = x;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += scan[T + 1 + i];
  scan[tid] = base + x - v;
  if (tid == T - 1) scan[T] = base + x;
  __syncthreads();
}

template <int T>
#ifndef RA_ANALYSIS_EU256
#define RA_ANALYSIS_EU256 4 // the minimum waves per SIMD the 128 / 256-thread kernel is compiled for: 4 = at most 128 VGPRs (143 without; see ra_lists_in_fft_buffer)
#endif
__global__ void __launch_bounds__(T, (T >= 512 ? 4 : RA_ANALYSIS_EU256)) // 512 / 1024 threads: 4 waves per SIMD (<= 128 VGPRs), so that two 512-thread groups share a CU
ra_analysis_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, int64_t t0, int64_t n, int64_t f_lo, int fb0, int nfb_total, int nfb_launch) {
#ifndef RA_ANALYSIS_PRIO
#define RA_ANALYSIS_PRIO 0
#endif
  // tuning knob: raising the wave priority here changes nothing measurable (22.7 - 22.9 ms per step for 0 .. 3): beside the renderer this
  // kernel is short of wave slots, not of issue cycles
  if (RA_ANALYSIS_PRIO) __builtin_amdgcn_s_setprio(RA_ANALYSIS_PRIO);
  const int tid = threadIdx.x;
  // XCD-aware frame order.  Workgroups go to the 8 XCDs round robin by linear id (= blockIdx.x + gridDim.x * blockIdx.y, gridDim.x a multiple
  // of 8), and each XCD has its own L2: with frame = blockIdx.x, the four frames that share a window's samples (75 % overlap) would land on
  // four different XCDs and every window would come from HBM (measured in round 1: each input sample fetched ~4 times).  Instead XCD g takes
  // the contiguous frames [g * per, (g + 1) * per) of the launch, in order: a window's samples are read from HBM once per XCD range.
  const int per = gridDim.x >> 3;
  const int fidx = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (fidx >= nfb_launch) return;
  const int fb = fb0 + fidx; // frame index inside the block [t0, t0 + n): this launch covers frames fb0 .. fb0 + nfb_launch - 1
  const int s = blockIdx.y;
  const int64_t f = f_lo + fb;
  const int M = p.M, N = p.N;

  double2 *z = reinterpret_cast<double2 *>(ra_smem);           // [ra_zsize(M)], element i lives in slot ra_zslot(i, zs)
  const int zs = ra_zshift(p.log2M);
  auto Z = [&](int i) -> double2 & { return z[ra_zslot(i, zs)]; };
  const int cap = ra_list_capacity(p);                         // harmonize may double a list; otherwise maxr entries suffice
  const bool lists_in_z = ra_lists_in_fft_buffer(M, cap, T);
  int *aux = reinterpret_cast<int *>(z + ra_zsize(M));         // [ra_aux_ints(T)] when the lists are in the buffer
  double2 *listA = lists_in_z ? z : z + ra_zsize(M);           // [cap]
  double2 *listB = listA + cap;                                // [cap]
  int *scan = lists_in_z ? aux + 16 : reinterpret_cast<int *>(listA); // [T+1] (outside the buffer it aliases listA: consumed (K read) before the lists are written)

  RaPhaseClock clk(b.dbg_clk, tid == 0);
  RaClockStamp stamp(b.dbg_stamp, tid == 0);
  // twiddles W^k of the real-spectrum split (step 3) for the pairs k0, k0 + T: fetched one trip ahead, the first trip's before the FFT
  // (W^(M-k) is the mirror image of W^k by the table's definition: ra_real_split_pair)
  double2 sw[2] = {make_double2(0, 0), make_double2(0, 0)};
  auto split_twiddles = [&](int k0) {
    const int Mh1 = M >> 1;
    if (k0 > Mh1) return;
    const int k1 = k0 + T, kb = (k1 <= Mh1) ? k1 : k0;
    sw[0] = b.tw_full[k0]; sw[1] = b.tw_full[kb];
  };
  split_twiddles(tid);
  // ---- 1 + 2. weighted window -> M point complex FFT of the packed sequence (x[2k], x[2k+1]) ------------------------
  // the first fused pass reads its (bit-reversed) inputs straight from HBM/L2: the window is 4 N bytes, touched once
  {
    const int64_t start = 2 + f * (int64_t)p.H; // absolute index of the window's first sample
    const float *hist = b.hist + (size_t)s * N;
    const float *vs = voice ? voice + (size_t)s * n : nullptr;
    const int h = N / 2;
    // branch-free so that the 16 sample / 16 window loads of a thread are all in flight together
    const float *vs0 = vs ? vs : hist; // never dereferenced past the block when voice is null (the select below picks 0)
    const bool fb_on = p.fb_gain != 0.f;
    const float *fbh = fb_on ? b.fb_hist + (size_t)s * b.fb_hist_len : nullptr, *fbb = fb_on ? b.fb_block + (size_t)s * n : nullptr;
    auto sample = [&](int i) -> double {
      const int ic = min(i, N - 1);
      const int64_t a = start + ic;
      const float *src = (a < t0) ? hist + (N - (t0 - a)) : vs0 + (vs ? a - t0 : 0);
      float v = *src;
      if (!(a < t0) && !vs) v = 0.f;
      float sum = 0.f;
      sum += p.input_gain * v; // analyze_until_input_starvation, lib.cpp:1626,1636 (float)
      if (fb_on) { // + feedback gain * the mono output fb_sz + 1 frames earlier (the aggregator pairs input a with output a - 1; lib.cpp:1637-1650)
        const int64_t j = a - 1 - p.fb_sz;
        float o = 0.f;
        if (a >= (int64_t)p.fb_sz + 2) o = (j >= t0) ? fbb[j - t0] : fbh[b.fb_hist_len - (t0 - j)];
        sum += p.fb_gain * o;
      }
      double x = (double)sum;  // PeriodicFFT::feed stores the float as FPT = double (periodicfft.cpp:61)
      const int j = (ic >= h) ? (ic - h) : (h - 1 - ic);
      const double w = b.half_window[j];
      return i < N ? x * w : 0.0;
    };
    // Usual case (the window lies inside the block, no zero padding, no feedback): one address computation per pair — samples 2k and
    // 2k + 1 sit on the same side of the window's centre, so their weights are one aligned 16-byte load of the half-window table.
    const bool plain = vs && !fb_on && p.L == N && start >= t0 && (N & 3) == 0; // workgroup-uniform
    if (plain) {
      const float *w0 = vs + (start - t0);
      const double2 *hw2 = reinterpret_cast<const double2 *>(b.half_window);
      ra_fft_fused(z, M, p.log2M, b.tw_half, tid, T, [&](int k) {
        const int i0 = 2 * k;
        const float v0 = w0[i0], v1 = w0[i0 + 1];
        const bool upper = i0 >= h;
        const double2 ww = hw2[upper ? (i0 - h) >> 1 : (h - 2 - i0) >> 1]; // upper: (w[j], w[j + 1]); lower: (w[j1 - 1], w[j1]) mirrored
        float s0 = 0.f, s1 = 0.f;
        s0 += p.input_gain * v0; s1 += p.input_gain * v1; // lib.cpp:1626,1636 (float)
        return make_double2((double)s0 * (upper ? ww.x : ww.y), (double)s1 * (upper ? ww.y : ww.x));
      });
    } else if (vs && !fb_on && start >= t0 && (N & 1) == 0) {
      // The same for any even window inside the block (the reference's window sizes are even, lib.cpp:484-487; its default 0.1814 s is
      // 8000 samples at 44.1 kHz, zero padded to 8192): the two weights of a pair are loaded separately (the pair may straddle the
      // window's centre or sit on an odd table index), and the zero padding is a multiplication on clamped loads — a select would let the
      // compiler sink the loads into branches and wait for each one in turn.
      const float *w0 = vs + (start - t0);
      ra_fft_fused(z, M, p.log2M, b.tw_half, tid, T, [&](int k) {
        const bool in = 2 * k < N;
        const int i0 = in ? 2 * k : N - 2, i1 = i0 + 1;
        const float v0 = w0[i0], v1 = w0[i1];
        const double wa = b.half_window[i0 >= h ? i0 - h : h - 1 - i0], wb = b.half_window[i1 >= h ? i1 - h : h - 1 - i1];
        float s0 = 0.f, s1 = 0.f;
        s0 += p.input_gain * v0; s1 += p.input_gain * v1; // lib.cpp:1626,1636 (float)
        const double keep = in ? 1.0 : 0.0;
        return make_double2((double)s0 * wa * keep, (double)s1 * wb * keep);
      });
    } else {
      ra_fft_fused(z, M, p.log2M, b.tw_half, tid, T, [&](int k) { return make_double2(sample(2 * k), sample(2 * k + 1)); });
    }
  }
  clk.mark(1);
  RA_STOP_AFTER(1);

  // ---- 3. split into the real spectrum, |X|^2 written over .x of the same slots -------------------
  // X[k] = E[k] + W^k O[k],  E = (Z[k] + conj Z[M-k]) / 2,  O = -i/2 (Z[k] - conj Z[M-k]);  bins 0..M-1 kept
  {
    auto sq = [](double2 x) -> double { return x.x * x.x + x.y * x.y; };
    const int Mh1 = M >> 1;
    for (int k0 = tid; k0 <= Mh1; k0 += 2 * T) { // two pairs per trip, their loads in flight together; the twiddles were fetched a trip ahead
      const int k1 = k0 + T;
      const bool has1 = k1 <= Mh1;
      const int ka = k0, kb = has1 ? k1 : k0;
      const int kma = (M - ka) % M, kmb = (M - kb) % M; // partner (k = 0 pairs with itself, k = M/2 too)
      const double2 za = Z(ka), zma = Z(kma), zb = Z(kb), zmb = Z(kmb);
      const double2 wa = sw[0], wb = sw[1];
      split_twiddles(k0 + 2 * T);
      // bins k and M - k from one twiddle product (ra_real_split_pair: bit-identical to two separate ra_real_split_bin evaluations);
      // the self-paired bins 0 and M / 2 take the first result only
      double2 xa, xma, xb, xmb;
      ra_real_split_pair(za, zma, wa, xa, xma);
      ra_real_split_pair(zb, zmb, wb, xb, xmb);
      const double sa = sq(xa), sma = sq(xma), sb = sq(xb), smb = sq(xmb);
      // the slots of a pair are private to this thread: safe to overwrite now
      Z(ka).x = sa;
      if (kma != ka) Z(kma).x = sma;
      if (has1) { Z(kb).x = sb; if (kmb != kb) Z(kmb).x = smb; }
    }
  }
  __syncthreads();
  clk.mark(2);
  RA_STOP_AFTER(2);
  if (b.dbg_sqmag) {
    double *dst = b.dbg_sqmag + ((size_t)s * nfb_total + fb) * M;
    for (int k = tid; k < M; k += T) dst[k] = Z(k).x;
  }

  // ---- 4. local maxima -> (pitch, volume) list in bin order ---------------------------------------
  // peak entries go to the .y halves: pitch of entry j at Z(j).y, volume at Z(M/2 + j).y  (count <= M/2 - 1).
  // Bins are scanned in contiguous chunks (count, block scan, compact the peak bins); the logarithms are then spread evenly:
  // thread t refines peaks t, t + T, ...
  const int chunk = (M + T - 1) / T;
  const int c0 = tid * chunk, c1 = min(M, c0 + chunk);
  const int Mh = M >> 1;
  int cnt = 0;
  const int i_first = max(c0, 1), i_stop = min(c1, M - 1); // bins i_first .. i_stop - 1; a sliding (a, m, c) window: one LDS read per bin
  if (i_first < i_stop) {
    double a = Z(i_first - 1).x, m = Z(i_first).x;
    for (int i = i_first; i < i_stop; ++i) {
      const double c = Z(i + 1).x;
      cnt += (a < m && m >= c) ? 1 : 0;
      a = m; m = c;
    }
  }
  block_exclusive_scan<T>(scan, cnt, tid);
  if (i_first < i_stop) {
    int pos = scan[tid];
    double a = Z(i_first - 1).x, m = Z(i_first).x;
    for (int i = i_first; i < i_stop; ++i) {
      const double c = Z(i + 1).x;
      if (a < m && m >= c) Z(pos++).y = (double)i;
      a = m; m = c;
    }
  }
  const int K = scan[T];
  __syncthreads();
  clk.mark(3);
  {
    const double bin_hz = (double)p.sample_rate / (double)p.L;
    for (int pk = tid; pk < K; pk += T) {
      const int i = (int)Z(pk).y;
      double a = Z(i - 1).x, m = Z(i).x, c = Z(i + 1).x;
      double da = sqmag_to_db(a), db = sqmag_to_db(m), dc = sqmag_to_db(c);
      double den = da - 2.0 * db + dc;
      double pp = den != 0.0 ? 0.5 * (da - dc) / den : 0.0;
      double freq = (i + pp) * bin_hz;
      double mag_db = db - 0.25 * (da - dc) * pp;
      // frequencies_to_pitches, algo.cpp:10-29 with midi.h:223-232 (f > 0 always holds here: i >= 1, pp > -1/2)
      double pitch = 69.0 + (12 / 1.0) * ra_log2(freq / (double)440.f);
      double vol = ra_exp10(mag_db / 20.0); // DbToMag: 10^(dB / 20)
      Z(pk).y = pitch;        // slot pk held this peak's bin: read above by this thread only
      Z(Mh + pk).y = vol;     // Mh + pk > K - 1: never a bin slot
    }
  }
  __syncthreads();
  clk.mark(4);
  RA_STOP_AFTER(4);

  // ---- 5. aggregate_pitches + reduce_pitches (algo.cpp:124-184, PonderateByVolume / SumVolumes) -------------------
  // Pitches ascend, so an interval opened at peak i0 runs up to the first peak j with pitch[j] - pitch[i0] > nearby:
  //  5a every peak finds that j by binary search (nxt, kept in the now unused .x half of its slot)
  //  5b the nxt chain from peak 0 lists the interval starts (tens to a few hundred hops): pointer jumping, ~log2 rounds
  //  5c each interval is summed by one thread in index order (the reference's summation order) and kept if its volume
  //     reaches min_volume; kept intervals are compacted in order with a block scan
  // (r04, measured and not kept: the interval starts by ONE wavefront walking the chain with 64 pitches in registers — readlane of the start's
  // pitch, one compare over all lanes, the ballot's first set bit: 15 % fewer instructions for the kernel, but ~300 dependent hops per frame on
  // one wave while the workgroup's other waves hold their registers and LDS at the barrier: configs[3]'s analysis 30.8 -> 48.3 ms.)
  // pointer jumping needs two ints per peak: beside the scan area inside the list region, which is idle until 5c — or, when the lists live in
  // the FFT buffer, the unused .x half of the peak's own slot (8 bytes: the pointer and the mark)
  const int area_ints = max(8 * cap, T + 32);
  int *jump = scan + T + 32, *mark = jump + K;
  const bool par = K > 0 && K <= 4 * T && (lists_in_z || T + 32 + 3 * K <= area_ints);
  auto JP = [&](int i) -> int & { return lists_in_z ? reinterpret_cast<int *>(&Z(i).x)[0] : jump[i]; };
  auto MK = [&](int i) -> int & { return lists_in_z ? reinterpret_cast<int *>(&Z(i).x)[1] : mark[i]; };
  for (int i = tid; i < K; i += T) {
    const double pi0 = Z(i).y;
    // the successor is a few peaks away (an interval holds ~3 peaks): gallop 1, 2, 4, ... peaks ahead, then bisect the last stretch —
    // 3 - 5 probes where the plain bisection of [i + 1, K) took log2 K = 10 (r04)
    int lo = i + 1, hi = K, st = 1;
    while (lo < hi) {
      const int probe = min(lo + st - 1, hi - 1);
      if (Z(probe).y - pi0 > p.nearby) { hi = probe; break; }
      lo = probe + 1; st <<= 1;
    }
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (Z(mid).y - pi0 > p.nearby) hi = mid; else lo = mid + 1;
    }
    if (par) { JP(i) = lo; MK(i) = (i == 0) ? 1 : 0; } else Z(i).x = (double)lo;
  }
  __syncthreads();
  clk.mark(5);
  if (par) {
    // 5b by pointer jumping.  Before round r: jump = nxt^(2^r), and the first 2^r interval starts are marked.  A marked peak marks
    // where its pointer leads (marks only ever land on interval starts, so marks seen within the same round are harmless), then every
    // pointer doubles.  Done when 2^(r+1) hops from peak 0 leave the list.  The pointers exist twice — a round reads one copy and writes the
    // other — so a round costs ONE barrier (r04; the second copy is the unused .x half of the peak's volume slot Mh + i, or a third int array
    // beside the scan area): these ~log2(intervals) rounds of barriers were 20 % of a configs[3] frame.
    auto JQ = [&](int i) -> int & { return lists_in_z ? reinterpret_cast<int *>(&Z(Mh + i).x)[0] : mark[K + i]; };
    // the rounds start from nxt^4 instead of nxt: every peak follows its pointer three more times on its own (dependent LDS reads, but no
    // barrier between them), peak 0 marks the three starts it passes: two rounds fewer
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + T * q;
      if (i < K) {
        int j = JP(i);
#pragma unroll
        for (int hop = 0; hop < 3; ++hop) j = (j < K) ? JP(j) : K;
        JQ(i) = j;
      }
    }
    if (tid == 0) {
      int j = JP(0);
      for (int hop = 0; hop < 3 && j < K; ++hop) { MK(j) = 1; j = JP(j); }
    }
    __syncthreads();
    bool flip = true;
    while (true) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = tid + T * q;
        if (i < K) {
          const int j = flip ? JQ(i) : JP(i);
          int jn = K;
          if (j < K) { if (MK(i)) MK(j) = 1; jn = flip ? JQ(j) : JP(j); }
          if (flip) JP(i) = jn; else JQ(i) = jn;
        }
      }
      __syncthreads();
      flip = !flip;
      if ((flip ? JQ(0) : JP(0)) >= K) break; // the same LDS word for every thread: uniform
    }
    // the marked peaks, in index order, are the interval starts
    const int c = (K + T - 1) / T;
    const int a0 = min(K, tid * c), a1 = min(K, a0 + c);
    int cntm = 0;
    for (int i = a0; i < a1; ++i) cntm += MK(i);
    block_exclusive_scan<T>(scan, cntm, tid);
    int posm = scan[tid];
    for (int i = a0; i < a1; ++i) if (MK(i)) Z(Mh + posm++).x = (double)i;
    if (tid == 0) Z(Mh + scan[T]).x = (double)K; // R <= K <= Mh - 1, so slot Mh + R exists; scan[T] = R stays for 5c
  } else if (tid == 0) {
    int R = 0;
    for (int i = 0; i < K; i = (int)Z(i).x) Z(Mh + R++).x = (double)i;
    Z(Mh + R).x = (double)K;
    scan[T] = R;
  }
  __syncthreads();
  clk.mark(6);
  RA_STOP_AFTER(6);
  const int R = scan[T];
  const int chunkR = (R + T - 1) / T;
  const int r0 = min(R, tid * chunkR), r1 = min(R, r0 + chunkR);
  auto interval = [&](int r, double &sumV, double &sumPV) {
    const int i0 = (int)Z(Mh + r).x, i1 = (int)Z(Mh + r + 1).x;
    sumV = 0; sumPV = 0;
    for (int i = i0; i < i1; ++i) { double pt = Z(i).y, v = Z(Mh + i).y; sumV += v; sumPV += pt * v; }
  };
  // (with the lists in the FFT buffer a thread owns at most four intervals: they are summed ONCE, into registers, before the scan that gives
  // them their places — r03 summed every interval twice, once to count and once to store)
  double2 item[4];
  unsigned have = 0; // (static indices only: a dynamically indexed private array would live in scratch memory)
  {
    int kept = 0;
    if (lists_in_z) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + j;
        item[j] = make_double2(0.0, 0.0);
        if (r < r1) {
          double sv, spv;
          interval(r, sv, spv);
          if (!(sv < p.min_volume)) { item[j] = make_double2(spv / sv, sv); have |= 1u << j; ++kept; }
        }
      }
    } else {
      for (int r = r0; r < r1; ++r) { double sv, spv; interval(r, sv, spv); kept += !(sv < p.min_volume) ? 1 : 0; }
    }
    block_exclusive_scan<T>(scan, kept, tid);
  }
  int pos = scan[tid];
  const int Rk = scan[T];
  if (lists_in_z) { // the list overwrites what `interval` read: everybody's last read is behind the scan's barriers
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (have & (1u << j)) { if (pos < cap) listA[pos] = item[j]; ++pos; }
  } else {
    __syncthreads(); // scan aliases listA: everyone has its offset before the list is written
    for (int r = r0; r < r1; ++r) {
      double sv, spv;
      interval(r, sv, spv);
      if (!(sv < p.min_volume)) { if (pos < cap) listA[pos] = make_double2(spv / sv, sv); ++pos; }
    }
  }
  __syncthreads();
  clk.mark(7);
  RA_STOP_AFTER(7);

  // ---- 6. shift / harmonize / autotune on the reduced list ---------------------------------------------------------
  // The autotune function (divisions, scale / chord searches) is evaluated for all items at once; one lane then runs the
  // order-dependent merge of algo.cpp:191-229 over the precomputed values, and the harmonize steps when they are enabled.
  // The ISO-226 pitches [29 doubles] and, behind them as floats, the 60 phon row [29]: the loudness keys' binary search (step 8) stays
  // out of global memory.  They sit in the idle FFT buffer behind the scan area (from M = 512 up; smaller frames have LDS to spare and
  // keep them behind the lists) — 360 bytes that would otherwise push an 8192-point frame over half a CU's LDS.
  double *lds_loud_pitches = lists_in_z ? reinterpret_cast<double *>(aux + 16 + (T + 32)) : (M >= 512 ? reinterpret_cast<double *>(z + 384) : reinterpret_cast<double *>(listB + cap));
  float *lds_loud_elv = reinterpret_cast<float *>(lds_loud_pitches + 29);
  if (tid < 29) { lds_loud_pitches[tid] = b.loud_pitches[tid]; lds_loud_elv[tid] = b.loud_elv[tid]; } // read after step 7's barriers
  // [0] = A, [1] = which list holds the result, [2] = Rn, [3] = input list, [4] = err, [5] = merge fallback, [6] = kept   (z is free now; with the lists in it: the small area)
  int *pub = lists_in_z ? aux : reinterpret_cast<int *>(z);
  if (tid == 0) {
    int err = Rk > cap ? 1 : 0;
    int Rn = min(Rk, cap);
    double2 *cur = listA, *oth = listB;
    if (p.harm_pre != 0.f) {
      shift_list(cur, Rn, p.shift_pre);
      Rn = harmonize_list(cur, oth, Rn, cap, p.harm_pre, &err);
    }
    pub[2] = Rn; pub[3] = (cur == listA) ? 0 : 1; pub[4] = err;
  }
  __syncthreads();
  {
    const int Rn = pub[2];
    double2 *cur = pub[3] ? listB : listA, *oth = pub[3] ? listA : listB;
    const bool shift_here = p.harm_pre == 0.f && p.shift_pre != 0.f; // shift_pitches, algo.cpp:307-316
    for (int i = tid; i < Rn; i += T) { // autotune_pitches, algo.cpp:191-229: the per-item half
      double2 pv = cur[i];
      if (shift_here) pv.x += p.shift_pre;
      double t;
      bool has;
      if (pv.x <= p.autotune_max_pitch) {
        has = autotune_fn(p, b.allowed, pv.x, t);
        if (has && t > p.autotune_max_pitch) t = pv.x;
      } else { t = pv.x; has = true; }
      const bool keep = has && fabs(t - pv.x) <= (double)p.tolerance;
      oth[i] = make_double2(t, keep ? pv.y : -1.0); // volumes are sums of positive magnitudes: -1 marks a dropped item
    }
  }
  __syncthreads();
  // The merge half: consecutive kept items whose autotuned pitch lies within 1e-4 of the run's first one pool their volumes
  // (algo.cpp:211-226).  In parallel: compact the kept items, flag run starts by comparing neighbours, sum every run from its start in
  // index order.  That equals the sequential rule whenever pitches do not descend and neighbours closer than 1e-4 are equal (autotuned
  // pitches are table values; untuned neighbours are > nearby apart); anything else falls back to the one-lane loop.
  {
    int *mscan = lists_in_z ? aux + 16 : reinterpret_cast<int *>(z) + 16; // block scan area: the idle FFT buffer (the lists are live here), or the small area
    const int Rn = pub[2];
    double2 *cur = pub[3] ? listB : listA, *oth = pub[3] ? listA : listB;
    const bool parm = Rn <= 4 * T, wave_local = Rn <= 64;
    int A_par = 0;
    if (wave_local) {
      // the usual case (tens of items): wave 0 alone, one item per lane, ballots instead of block scans, no barriers
      // (a wave's LDS operations execute in order)
      if (tid < 64) {
        const unsigned long long lower = (tid == 0) ? 0ull : (~0ull >> (64 - tid));
        const double2 e0 = (tid < Rn) ? oth[tid] : make_double2(0.0, -1.0);
        const bool keep = e0.y >= 0.0;
        const unsigned long long km = __ballot(keep);
        const int nk = __popcll(km);
        if (keep) cur[__popcll(km & lower)] = e0;
        const bool in = tid < nk;
        const double tj = in ? cur[tid].x : 0.0, tp = (in && tid > 0) ? cur[tid - 1].x : 0.0;
        const bool joins = in && tid > 0 && fabs(tp - tj) < (double)0.0001f;
        const bool bad = in && tid > 0 && (tj < tp || (joins && tj != tp));
        const unsigned long long sm = __ballot(in && !joins);
        const bool any_bad = __ballot(bad) != 0ull;
        if (tid == 0) { pub[5] = any_bad ? 1 : 0; pub[6] = nk; }
        if (!any_bad) {
          A_par = __popcll(sm);
          if (in && !joins) {
            double2 e = cur[tid];
            for (int q = tid + 1; q < nk && !((sm >> q) & 1ull); ++q) e.y += cur[q].y;
            oth[__popcll(sm & lower)] = e; // the per-item results in oth are dead on this path (all read above)
          }
        }
      }
    } else if (parm) {
      const int c = (Rn + T - 1) / T;
      const int a0 = min(Rn, tid * c), a1 = min(Rn, a0 + c);
      int cntk = 0;
      for (int i = a0; i < a1; ++i) cntk += (oth[i].y >= 0.0) ? 1 : 0;
      block_exclusive_scan<T>(mscan, cntk, tid);
      int pk = mscan[tid];
      const int nk = mscan[T];
      for (int i = a0; i < a1; ++i) { const double2 e = oth[i]; if (e.y >= 0.0) cur[pk++] = e; } // cur was consumed before the barrier above
      if (tid == 0) { pub[5] = 0; pub[6] = nk; }
      __syncthreads();
      const int c2 = (nk + T - 1) / T;
      const int b0 = min(nk, tid * c2), b1 = min(nk, b0 + c2);
      int cntb = 0;
      bool bad = false;
      for (int j = b0; j < b1; ++j) {
        if (j == 0) { ++cntb; continue; }
        const double tp = cur[j - 1].x, tj = cur[j].x;
        const bool joins = fabs(tp - tj) < (double)0.0001f;
        if (!joins) ++cntb;
        if (tj < tp || (joins && tj != tp)) bad = true;
      }
      if (bad) atomicOr(&pub[5], 1);
      block_exclusive_scan<T>(mscan, cntb, tid); // its barriers also publish pub[5]
      if (!pub[5]) {
        int pb = mscan[tid];
        A_par = mscan[T];
        for (int j = b0; j < b1; ++j) {
          if (j > 0 && fabs(cur[j - 1].x - cur[j].x) < (double)0.0001f) continue; // not a run start
          double2 e = cur[j];
          for (int q = j + 1; q < nk && fabs(cur[q - 1].x - cur[q].x) < (double)0.0001f; ++q) e.y += cur[q].y;
          oth[pb++] = e; // the per-item results in oth are dead on this path
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      int err = pub[4];
      int A = 0;
      double2 *res = cur, *spare = oth;
      if (parm && !pub[5]) { A = A_par; res = oth; spare = cur; }
      else { // one lane, in order; the output overwrites cur (A <= i)
        double2 last = make_double2(0.0, 0.0);
        bool have = false;
        const double2 *src = parm ? cur : oth;   // parm: already compacted into cur (all kept); else the per-item results
        const int ns = parm ? pub[6] : Rn;
        for (int i = 0; i < ns; ++i) {
          const double2 e = src[i];
          if (!(e.y >= 0.0)) continue;
          if (have && fabs(last.x - e.x) < (double)0.0001f) last.y += e.y;
          else { if (have) cur[A++] = last; last = e; have = true; }
        }
        if (have) cur[A++] = last;
      }
      shift_list(res, A, p.shift_post);
      A = harmonize_list(res, spare, A, cap, p.harm_post, &err);
      if (A > p.maxr) { err = 1; A = p.maxr; }
      if (err) atomicOr(b.error_flags, RA_ERR_OVERFLOW);
      pub[0] = A;
      pub[1] = (res == listA) ? 0 : 1;
      b.pl_count[(size_t)s * b.frames_cap + fb] = A;
    }
  }
  __syncthreads();
  clk.mark(8);
  {
    const int A = pub[0];
    const double2 *cur = pub[1] ? listB : listA;
    const size_t base = ((size_t)s * b.frames_cap + fb) * p.maxr;
    for (int i = tid; i < A; i += T) {
      double2 pv = cur[i];
      b.pl_pitch[base + i] = pv.x;
      b.pl_vol[base + i] = pv.y;
      // order_pitches_by_perceived_loudness key, lib.cpp:1717-1727 (a float vector, algo.cpp:380-395)
      b.pl_key[base + i] = (float)(pv.y / (double)equal_loudness_db(lds_loud_pitches, lds_loud_elv, pv.x));
    }
  }
  clk.mark(9);
  stamp.finish(0);
}

// keeps the last N input samples of every stream for the next block's overlapping windows
// (the memmove of periodicfft.cpp:167-175, done once per block instead of once per frame)
__global__ void ra_roll_history_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, int64_t n) {
  const int s = blockIdx.y;
  const int N = p.N;
  float *hist = b.hist + (size_t)s * N;
  const float *vs = voice ? voice + (size_t)s * n : nullptr;
  // new_hist[i] = concat(hist, block)[n + i]  for i in [0, N)
  if (n >= N) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) hist[i] = vs ? vs[n - N + i] : 0.f;
  } else {
    // shift by n inside one workgroup (grid.x == 1 for this branch)
    extern __shared__ float tmp[];
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      int64_t src = i + n;
      tmp[i] = (src < N) ? hist[src] : (vs ? vs[src - N] : 0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) hist[i] = tmp[i];
  }
}

// keeps the mono output of the last fb_hist_len frames for the next block's feedback term: new[i] = concat(hist, block)[n + i]
__global__ void ra_roll_feedback_kernel(RaBuffers b, int64_t n) {
  const int s = blockIdx.y;
  const int64_t Hn = b.fb_hist_len;
  float *hist = b.fb_hist + (size_t)s * Hn;
  const float *blk = b.fb_block + (size_t)s * n;
  if (n >= Hn) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Hn; i += (int64_t)gridDim.x * blockDim.x) hist[i] = blk[n - Hn + i];
  } else { // one workgroup per stream (grid.x == 1): shift by n in passes that never overwrite unread data (ascending, src index > dst index)
    for (int64_t base = 0; base < Hn; base += blockDim.x) {
      const int64_t i = base + threadIdx.x;
      float v = 0.f;
      if (i < Hn) { const int64_t src = i + n; v = (src < Hn) ? hist[src] : blk[src - Hn]; }
      __syncthreads();
      if (i < Hn) hist[i] = v;
      __syncthreads();
    }
  }
}

} // namespace

void ra_launch_roll_feedback(const RaParams &p, const RaBuffers &b, int64_t n, hipStream_t st) {
  if (n <= 0 || !b.fb_hist) return;
  const bool whole = n >= b.fb_hist_len;
  dim3 grid(whole ? (unsigned)std::min<int64_t>((b.fb_hist_len + 255) / 256, 1024) : 1u, (unsigned)p.n_streams);
  hipLaunchKernelGGL(ra_roll_feedback_kernel, grid, dim3(256), 0, st, b, n);
}

// Threads per analysis workgroup.  The FFT buffer of a frame decides how many workgroups share a CU (160 KB of LDS): up to N = 4096 three
// or more 256-thread groups fit; an 8192-point frame (70 KB + lists) is alone on its CU, so it brings 512 threads (two waves per SIMD —
// one wave per SIMD issues fp64 at half the rate two reach, scripts/ubench/fp64_rate.hip).  RA_ANALYSIS_T overrides (tuning aid).
int ra_analysis_threads(const RaParams &p) {
  static const int forced = [] { const char *e = getenv("RA_ANALYSIS_T"); return e ? atoi(e) : 0; }();
  if (forced == 128 || forced == 256 || forced == 512 || forced == 1024) return forced;
  return p.M >= 4096 ? 512 : 256;
}

size_t ra_analysis_lds_bytes(const RaParams &p) {
  size_t m = (size_t)ra_zsize(p.M);
  if (ra_lists_in_fft_buffer(p.M, ra_list_capacity(p), ra_analysis_threads(p))) return m * sizeof(double2) + (size_t)ra_aux_ints(ra_analysis_threads(p)) * sizeof(int);
  size_t lists = 2 * (size_t)ra_list_capacity(p) * sizeof(double2);
  size_t scan = ((size_t)ra_analysis_threads(p) + 32) * sizeof(int);
  size_t tables = p.M >= 512 ? 0 : 29 * (sizeof(double) + sizeof(float)) + 12; // the loudness tables move into the FFT buffer from M = 512 up
  return m * sizeof(double2) + (lists > scan ? lists : scan) + tables;
}

void ra_launch_analysis(const RaParams &p, const RaBuffers &b, const float *d_voice, int64_t t0, int64_t n, int64_t f_lo, int64_t fb0, int64_t nfb_chunk,
                        int64_t nfb_total, hipStream_t st) {
  if (nfb_chunk <= 0) return;
  dim3 grid((unsigned)(8 * ((nfb_chunk + 7) / 8)), (unsigned)p.n_streams); // a multiple of 8 per row: see the kernel's frame order
  size_t lds = ra_analysis_lds_bytes(p);
  static unsigned long long attr_done = 0;
  ra_once_per_device(attr_done, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_analysis_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_analysis_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_analysis_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_analysis_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipEvent_t ea = nullptr, eb = nullptr; // profiling: the launch stamps its own begin and end (RaExtTiming)
  if (ra_ext_timing) { ea = ra_ext_timing->a; eb = ra_ext_timing->b; ra_ext_timing->used = true; }
  switch (ra_analysis_threads(p)) {
  case 1024: hipExtLaunchKernelGGL(ra_analysis_kernel<1024>, grid, dim3(1024), (uint32_t)lds, st, ea, eb, 0, p, b, d_voice, t0, n, f_lo, (int)fb0, (int)nfb_total, (int)nfb_chunk); break;
  case 128: hipExtLaunchKernelGGL(ra_analysis_kernel<128>, grid, dim3(128), (uint32_t)lds, st, ea, eb, 0, p, b, d_voice, t0, n, f_lo, (int)fb0, (int)nfb_total, (int)nfb_chunk); break;
  case 512: hipExtLaunchKernelGGL(ra_analysis_kernel<512>, grid, dim3(512), (uint32_t)lds, st, ea, eb, 0, p, b, d_voice, t0, n, f_lo, (int)fb0, (int)nfb_total, (int)nfb_chunk); break;
  default: hipExtLaunchKernelGGL(ra_analysis_kernel<256>, grid, dim3(256), (uint32_t)lds, st, ea, eb, 0, p, b, d_voice, t0, n, f_lo, (int)fb0, (int)nfb_total, (int)nfb_chunk); break;
  }
}

void ra_launch_roll_history(const RaParams &p, const RaBuffers &b, const float *d_voice, int64_t n, hipStream_t st) {
  if (n <= 0) return;
  if (n >= p.N) {
    dim3 grid((unsigned)((p.N + 255) / 256), (unsigned)p.n_streams);
    hipLaunchKernelGGL(ra_roll_history_kernel, grid, dim3(256), 0, st, p, b, d_voice, n);
  } else {
    dim3 grid(1, (unsigned)p.n_streams);
    hipLaunchKernelGGL(ra_roll_history_kernel, grid, dim3(256), (size_t)p.N * sizeof(float), st, p, b, d_voice, n);
  }
}
