// LDS-resident fp64 FFT building blocks shared by the analysis and vocoder kernels.
// A real transform of length L runs as one complex transform of M = L/2 points held in LDS (16 B per point);
// butterflies are radix-2 decimation-in-time over bit-reversed input, twiddles come from a small table that
// stays L2/L1 resident (every workgroup reads the same M/2 entries).
#pragma once
#include "ra_internal.h"

// ---- the radix-2 DIT butterfly network, three stages at a time in registers --------------------------------------------------
// Stages with half = h, 2h, 4h of the radix-2 DIT network close over groups of 8 points {base + k h}: a thread loads one group,
// runs the 3 x 4 butterflies exactly as the stage-by-stage network would (same operands, same twiddles, same operation order:
// results are bit-identical) and stores it back, so an M-point transform takes ceil(log2 M / 3) LDS round trips and barriers
// instead of log2 M.  (The half = 8 and half = 64 passes both stay inside the 512 points a wave owns, so the barrier between them can be a
// wave-local ordering point; measured r03: slower — modulator transform 22.0 -> 24.7 ms, analysis 3.73 -> 3.93 ms per step — the waves of a
// workgroup drift apart and their LDS accesses collide more; every pass ends in the workgroup barrier.)
//
// LDS layout: element i of an M-point buffer lives in slot ra_zslot(i, ra_zshift(log2 M)) — the low four index bits XORed with bits
// 3..6 and, from M = 128 up, the low three also with the top three bits (a bijection inside every aligned run of 16 slots, no padding).
// ds_read_b128 is served in four 16-lane groups over 16 slots, ds_write_b128 in groups of 8 contiguous lanes over 8 slots
// (MI355X_MICROARCH.md, LDS): with this swizzle the first pass (lanes own bit-reversed groups: without it all 8 lanes of a write
// group hit one slot) and the stride-8 / stride-64 passes come out at 1.03 - 1.16 x the conflict-free cycle count, where one spare
// slot per 16 gave 1.4 - 2.5 x (scripts/lds_layout_sim.py; SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE with the padding).
__host__ __device__ inline int ra_zshift(int log2M) { return log2M >= 7 ? log2M - 3 : 31; }
__host__ __device__ inline int ra_zslot(int i, int zs) { return i ^ ((i >> zs) & 7) ^ ((i >> 3) & 15); }
__host__ __device__ inline int ra_zsize(int M) { return M < 16 ? 16 : M; }

// S fused stages starting at half = 1 << log2h.  FIRST: the group's inputs come from load(packed index) (bit-reversed order is
// resolved here: slot i of the DIT input holds packed element bitrev(i)) instead of from z.
// FMA (the vocoder's transforms only: nothing is decided on them, their bar is 1e-5 RMS): the twiddle product's multiply-adds are fused — two
// roundings fewer per butterfly, 8 instead of 10 instructions; the analysis keeps the unfused network whose bits the CPU restatement shares.
template <bool FMA>
__device__ inline void ra_butterfly(double2 &u_io, double2 &y_io, const double2 ww) {
  const double2 u = u_io, y = y_io;
  double vr, vi;
  if (FMA) { vr = fma(y.x, ww.x, -(y.y * ww.y)); vi = fma(y.x, ww.y, y.y * ww.x); }
  else { vr = y.x * ww.x - y.y * ww.y; vi = y.x * ww.y + y.y * ww.x; }
  u_io = make_double2(u.x + vr, u.y + vi);
  y_io = make_double2(u.x - vr, u.y - vi);
}

template <int S, bool FIRST, typename LoadFn, bool FMA = false>
__device__ inline void ra_fft_pass(double2 *z, int M, int log2M, int log2h, const double2 *__restrict__ tw_half, int tid, int T, LoadFn load) {
  constexpr int R = 1 << S;
  const int groups = M >> S, h = 1 << log2h, zs = ra_zshift(log2M);
  for (int gi = tid; gi < groups; gi += T) {
    // FIRST: slot 8 g + k of the DIT input holds packed element bitrev_m(8 g + k) = bitrev_S(k) << (m - S) | bitrev_{m-S}(g);
    // thread gi takes group g = bitrev_{m-S}(gi), so that consecutive lanes load consecutive packed elements (coalesced)
    const int g = (FIRST && log2M > S) ? (int)(__brev((unsigned)gi) >> (32 - (log2M - S))) : gi;
    const int r = g & (h - 1), base = ((g - r) << S) + r;
    // base and k h occupy disjoint bits and ra_zslot is XOR-linear: slot(base + k h) = slot(base) ^ slot(k h), the second term wave-uniform
    const int sb = ra_zslot(base, zs);
    double2 x[R], w[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (FIRST) {
        const unsigned rk = __brev((unsigned)k) >> (32 - S);
        x[k] = load((int)((rk << (log2M - S)) | (unsigned)gi));
      } else x[k] = z[sb ^ ra_zslot(k << log2h, zs)];
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int q = 0; q < (1 << s); ++q) w[(1 << s) - 1 + q] = tw_half[(r + q * h) << (log2M - 1 - s - log2h)]; // j * (M / len), len = 2 h 2^s
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (k & (1 << s)) continue;
        ra_butterfly<FMA>(x[k], x[k | (1 << s)], w[(1 << s) - 1 + (k & ((1 << s) - 1))]);
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) z[sb ^ ra_zslot(k << log2h, zs)] = x[k];
  }
  __syncthreads();
}

// ---- the same passes with the twiddle loads of pass n + 1 issued before pass n computes (one group per thread) -----------------------
// The twiddles of a pass depend only on the thread's group index, so their (L2) latency can hide behind the previous pass instead of
// sitting between two barriers.
template <int S>
__device__ inline void ra_fft_load_tw(double2 (&w)[8], int M, int log2M, int log2h, const double2 *__restrict__ tw_half, int g) {
  const int h = 1 << log2h, r = g & (h - 1);
  if (g >= (M >> S)) return;
#pragma unroll
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int q = 0; q < (1 << s); ++q) w[(1 << s) - 1 + q] = tw_half[(r + q * h) << (log2M - 1 - s - log2h)];
  }
}
template <int S, bool FMA = false>
__device__ inline void ra_fft_pass_w(double2 *z, int M, int log2M, int log2h, int g, const double2 (&w)[8]) {
  constexpr int R = 1 << S;
  const int h = 1 << log2h, zs = ra_zshift(log2M);
  if (g < (M >> S)) {
    const int r = g & (h - 1), base = ((g - r) << S) + r;
    const int sb = ra_zslot(base, zs);
    double2 x[R];
#pragma unroll
    for (int k = 0; k < R; ++k) x[k] = z[sb ^ ra_zslot(k << log2h, zs)];
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (k & (1 << s)) continue;
        ra_butterfly<FMA>(x[k], x[k | (1 << s)], w[(1 << s) - 1 + (k & ((1 << s) - 1))]);
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) z[sb ^ ra_zslot(k << log2h, zs)] = x[k];
  }
  __syncthreads();
}

// forward transform of the packed sequence load(0..M-1) into z (ra_zslot indexing), M = 1 << log2M
// clk (tuning aid, RA_PHASE_CLOCKS=1): laps 60 + pass index
template <typename LoadFn>
__device__ inline void ra_fft_fused(double2 *z, int M, int log2M, const double2 *__restrict__ tw_half, int tid, int T, LoadFn load, RaPhaseClock *clk = nullptr) {
  auto none = [](int) { return make_double2(0.0, 0.0); };
  int done;
  if (log2M == 0) {
    if (tid == 0) z[ra_zslot(0, 31)] = load(0);
    __syncthreads();
    return;
  }
  if (log2M >= 3 && (M >> 3) <= T) { // one group per thread in every pass: software-pipelined twiddles
    double2 wn[8], w[8];
    done = 3;
    int Sn = min(3, log2M - done);
    if (Sn == 3) ra_fft_load_tw<3>(wn, M, log2M, done, tw_half, tid);
    else if (Sn == 2) ra_fft_load_tw<2>(wn, M, log2M, done, tw_half, tid);
    else if (Sn == 1) ra_fft_load_tw<1>(wn, M, log2M, done, tw_half, tid);
    ra_fft_pass<3, true>(z, M, log2M, 0, tw_half, tid, T, load);
    int pass_no = 0;
    if (clk) clk->mark(60);
    while (done < log2M) {
      const int S = Sn, next = done + S;
#pragma unroll
      for (int q = 0; q < 8; ++q) w[q] = wn[q];
      Sn = min(3, log2M - next);
      if (Sn == 3) ra_fft_load_tw<3>(wn, M, log2M, next, tw_half, tid);
      else if (Sn == 2) ra_fft_load_tw<2>(wn, M, log2M, next, tw_half, tid);
      else if (Sn == 1) ra_fft_load_tw<1>(wn, M, log2M, next, tw_half, tid);
      // a final pass of fewer than 3 stages has more groups than threads: it takes the generic pass (twiddles loaded inside)
      if (S == 3) ra_fft_pass_w<3>(z, M, log2M, done, tid, w);
      else if (S == 2) { if ((M >> 2) <= T) ra_fft_pass_w<2>(z, M, log2M, done, tid, w); else ra_fft_pass<2, false>(z, M, log2M, done, tw_half, tid, T, none); }
      else { if ((M >> 1) <= T) ra_fft_pass_w<1>(z, M, log2M, done, tid, w); else ra_fft_pass<1, false>(z, M, log2M, done, tw_half, tid, T, none); }
      if (clk) clk->mark(61 + pass_no++);
      done = next;
    }
    return;
  }
  if (log2M >= 3) { ra_fft_pass<3, true>(z, M, log2M, 0, tw_half, tid, T, load); done = 3; }
  else if (log2M == 2) { ra_fft_pass<2, true>(z, M, log2M, 0, tw_half, tid, T, load); done = 2; }
  else { ra_fft_pass<1, true>(z, M, log2M, 0, tw_half, tid, T, load); done = 1; }
  while (log2M - done >= 3) { ra_fft_pass<3, false>(z, M, log2M, done, tw_half, tid, T, none); done += 3; }
  if (log2M - done == 2) ra_fft_pass<2, false>(z, M, log2M, done, tw_half, tid, T, none);
  else if (log2M - done == 1) ra_fft_pass<1, false>(z, M, log2M, done, tw_half, tid, T, none);
}

// ---- the vocoder modulator's transform (r04): same network, same LDS layout, but -------------------------------------------------------
//  * the first pass is what it is when its twiddles are exact — an 8-point DFT with the constants 1, -i, (+-1 - i) / sqrt 2 — and its inputs
//    that lie in the zero padding are never loaded or added: a window of W samples in a transform of L = 2 M leaves the packed elements from
//    W / 2 on zero, and input rk of group gi is element rk M / 8 + gi, so only the first NZ = ceil(W / 2 / (M / 8)) of the 8 inputs can be
//    non-zero (NZ >= 5 because L < 2 W; 4800 samples in 8192: NZ = 5, the pass's first stage shrinks from four butterflies to one);
//  * the later passes fuse their multiply-adds (ra_butterfly<true>).
// Requires log2M >= 3 and M / 8 <= T (one group per thread in every three-stage pass).  load(i): packed element i < NZ M / 8.
template <int NZ, typename LoadFn>
__device__ inline void ra_fft_first_pass_pruned(double2 *z, int M, int log2M, int tid, LoadFn load) {
  const int G = M >> 3, zs = ra_zshift(log2M);
  if (tid < G) {
    const int gi = tid;
    const int g = log2M > 3 ? (int)(__brev((unsigned)gi) >> (32 - (log2M - 3))) : 0; // consecutive lanes load consecutive packed elements
    double2 a[8];
#pragma unroll
    for (int rk = 0; rk < 8; ++rk) a[rk] = rk < NZ ? load(rk * G + gi) : make_double2(0.0, 0.0);
    auto add = [](double2 p, double2 q) { return make_double2(p.x + q.x, p.y + q.y); };
    auto sub = [](double2 p, double2 q) { return make_double2(p.x - q.x, p.y - q.y); };
    auto add_mi = [](double2 p, double2 q) { return make_double2(p.x + q.y, p.y - q.x); }; // p - i q
    auto sub_mi = [](double2 p, double2 q) { return make_double2(p.x - q.y, p.y + q.x); }; // p + i q
    // stage 1 (pairs 4 apart): a missing partner leaves sum and difference equal to the input
    const double2 b0 = NZ > 4 ? add(a[0], a[4]) : a[0], b1 = NZ > 4 ? sub(a[0], a[4]) : a[0];
    const double2 b2 = NZ > 6 ? add(a[2], a[6]) : a[2], b3 = NZ > 6 ? sub(a[2], a[6]) : a[2];
    const double2 b4 = NZ > 5 ? add(a[1], a[5]) : a[1], b5 = NZ > 5 ? sub(a[1], a[5]) : a[1];
    const double2 b6 = NZ > 7 ? add(a[3], a[7]) : a[3], b7 = NZ > 7 ? sub(a[3], a[7]) : a[3];
    // stage 2 (twiddles 1, -i)
    const double2 c0 = add(b0, b2), c2 = sub(b0, b2), c1 = add_mi(b1, b3), c3 = sub_mi(b1, b3);
    const double2 c4 = add(b4, b6), c6 = sub(b4, b6), c5 = add_mi(b5, b7), c7 = sub_mi(b5, b7);
    // stage 3 (twiddles 1, (1 - i) / sqrt 2, -i, (-1 - i) / sqrt 2)
    const double r = 0.70710678118654752440;
    const double2 t5 = make_double2((c5.x + c5.y) * r, (c5.y - c5.x) * r);    // c5 (1 - i) / sqrt 2
    const double2 t7 = make_double2((c7.y - c7.x) * r, -(c7.x + c7.y) * r);   // c7 (-1 - i) / sqrt 2
    double2 y[8];
    y[0] = add(c0, c4); y[4] = sub(c0, c4);
    y[1] = add(c1, t5); y[5] = sub(c1, t5);
    y[2] = add_mi(c2, c6); y[6] = sub_mi(c2, c6);
    y[3] = add(c3, t7); y[7] = sub(c3, t7);
    const int sb = ra_zslot(g << 3, zs);
#pragma unroll
    for (int k = 0; k < 8; ++k) z[sb ^ ra_zslot(k, zs)] = y[k];
  }
  __syncthreads();
}
template <int NZ, typename LoadFn>
__device__ inline void ra_fft_fused_fast(double2 *z, int M, int log2M, const double2 *__restrict__ tw_half, int tid, int T, LoadFn load) {
  auto none = [](int) { return make_double2(0.0, 0.0); };
  double2 wn[8], w[8];
  int done = 3;
  int Sn = min(3, log2M - done);
  if (Sn == 3) ra_fft_load_tw<3>(wn, M, log2M, done, tw_half, tid);
  else if (Sn == 2) ra_fft_load_tw<2>(wn, M, log2M, done, tw_half, tid);
  else if (Sn == 1) ra_fft_load_tw<1>(wn, M, log2M, done, tw_half, tid);
  ra_fft_first_pass_pruned<NZ>(z, M, log2M, tid, load);
  while (done < log2M) {
    const int S = Sn, next = done + S;
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = wn[q];
    Sn = min(3, log2M - next);
    if (Sn == 3) ra_fft_load_tw<3>(wn, M, log2M, next, tw_half, tid);
    else if (Sn == 2) ra_fft_load_tw<2>(wn, M, log2M, next, tw_half, tid);
    else if (Sn == 1) ra_fft_load_tw<1>(wn, M, log2M, next, tw_half, tid);
    if (S == 3) ra_fft_pass_w<3, true>(z, M, log2M, done, tid, w);
    else if (S == 2) { if ((M >> 2) <= T) ra_fft_pass_w<2, true>(z, M, log2M, done, tid, w); else ra_fft_pass<2, false, decltype(none), true>(z, M, log2M, done, tw_half, tid, T, none); }
    else { if ((M >> 1) <= T) ra_fft_pass_w<1, true>(z, M, log2M, done, tid, w); else ra_fft_pass<1, false, decltype(none), true>(z, M, log2M, done, tw_half, tid, T, none); }
    done = next;
  }
}

// One bin of the real spectrum from the packed transform: X[k] = E[k] + W^k O[k],
// E = (Z[k] + conj Z[M-k]) / 2, O = -i/2 (Z[k] - conj Z[M-k]).  a = Z[k mod M], c = Z[(M-k) mod M].
__device__ inline double2 ra_real_split_bin(double2 a, double2 c, double2 w) {
  double er = 0.5 * (a.x + c.x), ei = 0.5 * (a.y - c.y);
  double dr = a.x - c.x, di = a.y + c.y;
  double o_r = 0.5 * di, o_i = -0.5 * dr;
  return make_double2(er + (w.x * o_r - w.y * o_i), ei + (w.x * o_i + w.y * o_r));
}

// Bins k and M - k of the real spectrum at once, for 0 < k < M / 2: a = Z[k], c = Z[M - k], w = W^k.  The split's table is mirrored by
// definition, W^(M - k) = (-w.x, w.y), and E[M - k] = conj E[k], O[M - k] = conj O[k], so bin M - k evaluated by ra_real_split_bin(c, a, W^(M-k))
// is (er - pr, pi - ei) where bin k is (er + pr, ei + pi): every product and sum of the two separate evaluations appears here once, with the
// same operands in the same order (a sign change of an operand changes the sign of a correctly rounded result and nothing else) — the two
// bins are bit-identical to the separate evaluations, at 24 instead of 38 operations.
__device__ inline void ra_real_split_pair(double2 a, double2 c, double2 w, double2 &xk, double2 &xm) {
  const double er = 0.5 * (a.x + c.x), ei = 0.5 * (a.y - c.y);
  const double dr = a.x - c.x, di = a.y + c.y;
  const double o_r = 0.5 * di, o_i = -0.5 * dr;
  const double pr = w.x * o_r - w.y * o_i, pi = w.x * o_i + w.y * o_r;
  xk = make_double2(er + pr, ei + pi);
  xm = make_double2(er - pr, pi - ei);
}
