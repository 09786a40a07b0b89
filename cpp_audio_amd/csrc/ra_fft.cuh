// LDS-resident fp64 FFT building blocks shared by the analysis and vocoder kernels.
// A real transform of length L runs as one complex transform of M = L/2 points held in LDS (16 B per point);
// butterflies are radix-2 decimation-in-time over bit-reversed input, twiddles come from a small table that
// stays L2/L1 resident (every workgroup reads the same M/2 entries).
#pragma once
#include "ra_internal.h"

// in-place DIT passes over z[0..M) (already bit-reversed).  inverse: conjugated twiddles, unnormalised.
__device__ inline void ra_fft_dit(double2 *z, int M, const double2 *__restrict__ tw_half, int tid, int T, bool inverse) {
  for (int len = 2; len <= M; len <<= 1) {
    const int half = len >> 1, step = M / len;
    for (int bf = tid; bf < (M >> 1); bf += T) {
      int grp = bf / half, j = bf - grp * half;
      int i0 = grp * len + j;
      double2 w = tw_half[j * step];
      if (inverse) w.y = -w.y;
      double2 u = z[i0], x = z[i0 + half];
      double vr = x.x * w.x - x.y * w.y;
      double vi = x.x * w.y + x.y * w.x;
      z[i0] = make_double2(u.x + vr, u.y + vi);
      z[i0 + half] = make_double2(u.x - vr, u.y - vi);
    }
    __syncthreads();
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
