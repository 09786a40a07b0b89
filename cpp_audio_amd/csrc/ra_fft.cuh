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
template <int S, bool FIRST, typename LoadFn>
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
        const double2 ww = w[(1 << s) - 1 + (k & ((1 << s) - 1))];
        const double2 u = x[k], y = x[k | (1 << s)];
        const double vr = y.x * ww.x - y.y * ww.y;
        const double vi = y.x * ww.y + y.y * ww.x;
        x[k] = make_double2(u.x + vr, u.y + vi);
        x[k | (1 << s)] = make_double2(u.x - vr, u.y - vi);
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
template <int S>
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
        const double2 ww = w[(1 << s) - 1 + (k & ((1 << s) - 1))];
        const double2 u = x[k], y = x[k | (1 << s)];
        const double vr = y.x * ww.x - y.y * ww.y;
        const double vi = y.x * ww.y + y.y * ww.x;
        x[k] = make_double2(u.x + vr, u.y + vi);
        x[k | (1 << s)] = make_double2(u.x - vr, u.y - vi);
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

// One bin of the real spectrum from the packed transform: X[k] = E[k] + W^k O[k],
// E = (Z[k] + conj Z[M-k]) / 2, O = -i/2 (Z[k] - conj Z[M-k]).  a = Z[k mod M], c = Z[(M-k) mod M].
__device__ inline double2 ra_real_split_bin(double2 a, double2 c, double2 w) {
  double er = 0.5 * (a.x + c.x), ei = 0.5 * (a.y - c.y);
  double dr = a.x - c.x, di = a.y + c.y;
  double o_r = 0.5 * di, o_i = -0.5 * dr;
  return make_double2(er + (w.x * o_r - w.y * o_i), ei + (w.x * o_i + w.y * o_r));
}
