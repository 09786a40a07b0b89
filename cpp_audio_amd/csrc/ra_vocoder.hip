// K4 — FFT band vocoder (reference source/rt.resynth.lib.vocoder.cpp, the live FFTModulator / FFTCarrier pair).
//
//  K4a modulator : one workgroup per (stream, modulator frame): Gaussian-windowed |X|^2 of the voice (PeriodicFFT<SqMag>,
//                  vocoder.cpp:106-163) -> band amplitudes sqrt(sum |X|^2 over the band's bins).
//  K4b carrier   : one workgroup per (stream, carrier frame): rectangular 2S window of the carrier, real FFT, every bin scaled
//                  by its band's amplitude (RealFBins_::modulate_bands, vocoder.cpp:453-456), inverse FFT, 1/L (vocoder.cpp:420-475).
//  K4c mix       : per output frame the 50 % overlap cross-fade of the two latest carrier frames (FFTCarrier::feed, :501-530)
//                  and the mix vol.mod*m + vol.car*c + vol.voc*y (Vocoder compute lambda, :799-808) into a double buffer that
//                  K3 adds after the synth channels (compute registration order lib.cpp:1353,1377).
//  K4d roll      : carries the last frames / samples over to the next block.
//
// Frames are independent given "the latest modulator frame fired at or before this carrier frame", so all of a block's
// frames run concurrently; the sequential per-sample loop of the reference is only an ordering, not a data dependence.
#include "ra_fft.cuh"
#include "ra_lane.cuh"
#include <algorithm>
#include <cstdlib>

extern __shared__ __align__(16) unsigned char ra_smem_voc[];

namespace {


__device__ inline int ilog2(int m) { return 31 - __clz(m); }

// ---- K4a ------------------------------------------------------------------------------------------
// T threads per frame: 256, or 512 when the FFT buffer (M >= 4096: 70 KB) leaves room for only two workgroups per CU.
// BIG: the transform buffer does not fit the 160 KB of LDS (windows above 16384 samples, up to the reference's 65536): the same code runs on a
// per-workgroup buffer in HBM (the barriers between passes order a workgroup's global writes and reads); slow, but the size is supported.
template <int T, bool BIG>
__global__ void __launch_bounds__(T, 4) ra_voc_modulator_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, int64_t t0, int64_t n, int64_t j_lo,
                                                            int n_mod, int fpw) {
  // a workgroup takes fpw consecutive frames of its stream (workgroup launches amortised; the window weights and twiddles stay warm)
  const int s = blockIdx.y;
  for (int jb = blockIdx.x * fpw, jb_end = min(jb + fpw, n_mod); jb < jb_end; ++jb) {
  // The body is the single-frame kernel; tid and M pass through an empty asm so that nothing computed from them is hoisted out of the
  // loop (hoisted addresses and twiddles cost 100 VGPRs and with them the second workgroup per CU).
  int tid = threadIdx.x, M = p.voc_Mm;
  asm volatile("" : "+v"(tid), "+s"(M));
  const int W = p.voc_Wm, S = p.voc_S, log2M = ilog2(M);
  double2 *z = BIG ? b.voc_big + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * b.voc_big_slot
                   : reinterpret_cast<double2 *>(ra_smem_voc); // element i lives in slot ra_zslot(i, zs)
  const int zs = ra_zshift(log2M);
  auto Z = [&](int i) -> double2 & { return z[ra_zslot(i, zs)]; };
  // twiddles of the first trip of the real-spectrum split, fetched before the FFT
  const int Mh1 = M >> 1;
  RaPhaseClock clk(b.dbg_clk, tid == 0);
  double2 sw0 = make_double2(0, 0), sw1 = sw0;
  if (tid <= Mh1) { sw0 = b.vm_tw_full[tid]; sw1 = b.vm_tw_full[(M - tid) % M]; }
  {
    const int64_t start = (j_lo + jb) * (int64_t)S;
    const float *hist = b.voc_mod_hist + (size_t)s * W;
    const float *vs = voice ? voice + (size_t)s * n : nullptr;
    const int h = W / 2;
    // usual case: the window lies inside the block and its two halves hold whole pairs — one address computation and one 16-byte
    // weight load per packed element, consecutive lanes on consecutive samples (the fused first pass resolves the bit reversal)
    const bool plain = vs && start >= t0 && (W & 3) == 0; // workgroup-uniform
    if (plain) {
      const float *w0 = vs + (start - t0);
      const double2 *hw2 = reinterpret_cast<const double2 *>(b.voc_half_window);
      ra_fft_fused(z, M, log2M, b.vm_tw_half, tid, T, [&](int k) {
        // Zero padding without control flow: out-of-window elements load the window's last pair (one address for the whole wave) and are
        // multiplied by 0 — a select would let the compiler sink the loads into a branch and wait for each one in turn; this way a
        // thread's 8 sample pairs and 8 weight pairs are in flight together.
        const bool in = 2 * k < W;
        const int i0 = in ? 2 * k : W - 2;
        const float v0 = w0[i0], v1 = w0[i0 + 1];
        const bool upper = i0 >= h;
        const double2 ww = hw2[upper ? (i0 - h) >> 1 : (h - 2 - i0) >> 1];
        const double keep = in ? 1.0 : 0.0;
        return make_double2((double)v0 * (upper ? ww.x : ww.y) * keep, (double)v1 * (upper ? ww.y : ww.x) * keep);
      }, &clk);
    } else {
      auto sample = [&](int i) -> double {
        if (i >= W) return 0.0;
        int64_t a = start + i;
        float v = (a < t0) ? hist[W - (t0 - a)] : (vs ? vs[a - t0] : 0.f);
        int j = (i >= h) ? (i - h) : (h - 1 - i);
        return (double)v * b.voc_half_window[j];
      };
      ra_fft_fused(z, M, log2M, b.vm_tw_half, tid, T, [&](int k) { return make_double2(sample(2 * k), sample(2 * k + 1)); });
    }
  }
  for (int k = tid; k <= Mh1; k += T) {
    const int km = (M - k) % M;
    const double2 zk = Z(k), zm = Z(km);
    const double2 wk = sw0, wm = sw1;
    if (k + T <= Mh1) { sw0 = b.vm_tw_full[k + T]; sw1 = b.vm_tw_full[M - k - T]; }
    const double2 xk = ra_real_split_bin(zk, zm, wk);
    const double sk = xk.x * xk.x + xk.y * xk.y;
    Z(k).x = sk; // the pair's slots are private to this thread
    if (km != k) { const double2 xm = ra_real_split_bin(zm, zk, wm); Z(km).x = xm.x * xm.x + xm.y * xm.y; }
  }
  __syncthreads();
  clk.mark(58);
  // band amplitudes sqrt(sum of |X|^2 over the band's bins) (vocoder.cpp:137-160).  One wave per band, four bands at a time (their LDS
  // reads and lane sums overlap); lanes stride over the band's bins and the 64 partial sums are folded pairwise on the VALU: the
  // reference adds in bin order, so a sum differs from its value by rounding only (~1e-16 relative on a gain; a band of one bin is exact).
  // Bands hold up to several hundred bins: one lane per band cost more than the whole FFT.
  const int B = p.voc_bands, lane = tid & 63, nw = T >> 6, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int band0 = wv; band0 < B; band0 += 4 * nw) {
    int lo[4], len[4], maxlen = 0;
    double acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int band = band0 + u * nw;
      const bool ok = band < B;
      const int l = ok ? b.voc_mod_band_start[2 * band] : 0, hh = ok ? b.voc_mod_band_start[2 * band + 1] : 0;
      lo[u] = l; len[u] = hh - l; acc[u] = 0.0;
      maxlen = max(maxlen, len[u]);
    }
    for (int i = lane; i < maxlen; i += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i < len[u]) acc[u] += Z(lo[u] + i).x;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int band = band0 + u * nw;
      const double tot = wave_sum_row0(acc[u]);
      if (lane == 0 && band < B) b.voc_amp[((size_t)s * b.voc_mod_cap + jb) * B + band] = sqrt(tot);
    }
  }
  clk.mark(59);
  __syncthreads(); // the buffer is reused by the next frame
  }
}

// ---- K4a, the product path (r04) --------------------------------------------------------------------
// The modulator's spectrum decides nothing (its bar is the vocoded signal's 1e-5 RMS), so its transform need not share the analysis' unfused
// network: ra_fft_fused_fast — a first pass with exact constants that never touches the zero padding, fused multiply-adds after it — and
//  * the real-spectrum split works on the pair (k, M - k) at once: with E = (Z[k] + conj Z[M-k]) / 2, O = -i (Z[k] - conj Z[M-k]) / 2 and
//    P = W^k O the two bins are E + P and conj(E - P) (W^(M-k) = -conj W^k): one twiddle, one complex product, |E + P|^2 and |E - P|^2;
//  * |X|^2 never goes back to LDS: a wave's 64 bins of a trip are consecutive, so they belong to one band or to a few; per band present
//    in the wave the lanes' values are folded on the VALU (wave_sum_row0) and lane 0 adds the total to the wave's own row of partial sums;
//    the rows are added in wave order at the end.  (r03 wrote 4096 doubles back, gave every band to one wave and waited for the wave with
//    the widest bands: 34 % of the frame for split + sums.)
// The order of the additions inside a band differs from r03's (and from the reference's bin order) by rounding: ~1e-16 of an amplitude.
// bin_band: [M + 1] band of every bin, -1 below the first band's lower edge, `bands` above the last band's upper edge (and for the Nyquist
// bin M, which the reference drops): non-decreasing in the bin index.
template <int T, int NZ>
__global__ void __launch_bounds__(T, 4) ra_voc_modulator_fast_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, int64_t t0, int64_t n, int64_t j_lo,
                                                                 int n_mod, int fpw) {
  const int s = blockIdx.y;
  constexpr int NW = T / 64;
  for (int jb = blockIdx.x * fpw, jb_end = min(jb + fpw, n_mod); jb < jb_end; ++jb) {
  int tid = threadIdx.x, M = p.voc_Mm;
  asm volatile("" : "+v"(tid), "+s"(M)); // (as in the kernel above: nothing computed from them is hoisted out of the frame loop)
  const int W = p.voc_Wm, S = p.voc_S, log2M = ilog2(M), B = p.voc_bands;
  double2 *z = reinterpret_cast<double2 *>(ra_smem_voc);
  double *part = reinterpret_cast<double *>(z + ra_zsize(M)); // [NW][64] the waves' band sums of this frame
  const int zs = ra_zshift(log2M);
  auto Z = [&](int i) -> double2 & { return z[ra_zslot(i, zs)]; };
  const int Mh1 = M >> 1, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < NW * 64) part[tid] = 0.0; // (read after the transform's barriers)
  double2 sw0 = make_double2(0, 0);
  if (tid <= Mh1) sw0 = b.vm_tw_full[tid];
  {
    const int64_t start = (j_lo + jb) * (int64_t)S;
    const float *hist = b.voc_mod_hist + (size_t)s * W;
    const float *vs = voice ? voice + (size_t)s * n : nullptr;
    const int h = W / 2;
    const bool plain = vs && start >= t0; // workgroup-uniform (W is a multiple of 4 on this path)
    if (plain) {
      const float *w0 = vs + (start - t0);
      const double2 *hw2 = reinterpret_cast<const double2 *>(b.voc_half_window);
      ra_fft_fused_fast<NZ>(z, M, log2M, b.vm_tw_half, tid, T, [&](int k) {
        // (only the last of the NZ inputs of a group can lie in the padding: a clamped load times 0, so that the loads stay unconditional)
        const bool in = 2 * k < W;
        const int i0 = in ? 2 * k : W - 2;
        const float v0 = w0[i0], v1 = w0[i0 + 1];
        const bool upper = i0 >= h;
        const double2 ww = hw2[upper ? (i0 - h) >> 1 : (h - 2 - i0) >> 1];
        const double keep = in ? 1.0 : 0.0;
        return make_double2((double)v0 * (upper ? ww.x : ww.y) * keep, (double)v1 * (upper ? ww.y : ww.x) * keep);
      });
    } else { // the block's first frames look back into the history: the same values from two places
      auto sample = [&](int i) -> double {
        if (i >= W) return 0.0;
        int64_t a = start + i;
        float v = (a < t0) ? hist[W - (t0 - a)] : (vs ? vs[a - t0] : 0.f);
        int j = (i >= h) ? (i - h) : (h - 1 - i);
        return (double)v * b.voc_half_window[j] * 1.0;
      };
      ra_fft_fused_fast<NZ>(z, M, log2M, b.vm_tw_half, tid, T, [&](int k) { return make_double2(sample(2 * k), sample(2 * k + 1)); });
    }
  }
  // split + |X|^2 + band sums, one pair of bins per lane and trip
  for (int k0 = wv * 64; k0 <= Mh1; k0 += T) { // wave-uniform trip: bins k0 .. k0 + 63 and M - k0 - 63 .. M - k0
    const int k = k0 + lane;
    const bool on = k <= Mh1;
    const int kc = on ? k : Mh1, km = M - kc; // km in M/2 .. M
    const double2 zk = Z(kc), zm = Z(km & (M - 1));
    const double2 w = sw0;
    if (k + T <= Mh1) sw0 = b.vm_tw_full[k + T];
    const double ex = zk.x + zm.x, ey = zk.y - zm.y;      // 2 E
    const double ox = zk.y + zm.y, oy = zm.x - zk.x;      // 2 O = -i (Z[k] - conj Z[M-k])
    const double px = fma(w.x, ox, -(w.y * oy)), py = fma(w.x, oy, w.y * ox); // 2 W^k O
    const double ax = ex + px, ay = ey + py, bx = ex - px, by = ey - py;
    double sk = 0.25 * fma(ax, ax, ay * ay), sm = 0.25 * fma(bx, bx, by * by);
    int bk = on ? b.voc_mod_bin_band[kc] : B, bm = (on && km != kc) ? b.voc_mod_bin_band[km] : B; // (k = M/2 pairs with itself: counted once)
    if (bk < 0 || bk >= B) sk = 0.0;
    if (bm < 0 || bm >= B) sm = 0.0;
    auto fold = [&](double v, int band, int b_first, int b_last) { // bands b_first <= ... <= b_last are the ones present in the wave (clamped to 0 .. B - 1)
      for (int q = b_first; q <= b_last; ++q) {
        const double tot = wave_sum_row0(band == q ? v : 0.0);
        if (lane == 0) part[wv * 64 + q] += tot;
      }
    };
    // ascending bins k0 .. k0 + 63: lane 0 holds the lowest band, lane 63 the highest; the mirrored bins descend
    {
      const int lo = max(__builtin_amdgcn_readlane(bk, 0), 0), hi = min(__builtin_amdgcn_readlane(bk, 63), B - 1);
      fold(sk, bk, lo, hi);
    }
    {
      const int lo = max(__builtin_amdgcn_readlane(bm, 63), 0), hi = min(__builtin_amdgcn_readlane(bm, 0), B - 1);
      fold(sm, bm, lo, hi);
    }
  }
  __syncthreads();
  if (tid < B) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < NW; ++q) tot += part[q * 64 + tid];
    b.voc_amp[((size_t)s * b.voc_mod_cap + jb) * B + tid] = sqrt(tot);
  }
  __syncthreads(); // the buffer and the partial sums are reused by the next frame
  }
}

// ---- K4b ------------------------------------------------------------------------------------------
// T threads per frame: one wavefront for the usual short carrier frames (L = 512: 32 radix-8 groups per pass), 256 for long ones.
template <int T, bool BIG>
__global__ void __launch_bounds__(T) ra_voc_carrier_kernel(RaParams p, RaBuffers b, const float *__restrict__ carrier, int64_t t0, int64_t n, int64_t i_lo,
                                                          int64_t j_lo, int64_t n_mod, int n_car, int fpw) {
  // a workgroup takes fpw consecutive frames of its stream, like the modulator's (tid and M laundered against hoisting)
  const int s = blockIdx.y;
  for (int ib = blockIdx.x * fpw, ib_end = min(ib + fpw, n_car); ib < ib_end; ++ib) {
  int tid = threadIdx.x, M = p.voc_Mc;
  asm volatile("" : "+v"(tid), "+s"(M));
  const int L = p.voc_Lc, S = p.voc_S, W = 2 * p.voc_S, B = p.voc_bands, log2M = ilog2(M);
  double2 *z = BIG ? b.voc_big + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * b.voc_big_slot
                   : reinterpret_cast<double2 *>(ra_smem_voc);  // [ra_zsize(M)] transform buffer (BIG: in HBM, see the modulator)
  double2 *zin = z + ra_zsize(M);                      // [M] input of the inverse transform
  const int zs = ra_zshift(log2M);
  auto Z = [&](int i) -> double2 & { return z[ra_zslot(i, zs)]; };
  const int64_t i = i_lo + ib;
  const int64_t fire = (int64_t)W - 1 + i * S;
  {
    const int64_t start = i * (int64_t)S;
    const float *hist = b.voc_car_hist + (size_t)s * W;
    const float *cs = carrier ? carrier + (size_t)s * n : nullptr;
    auto sample = [&](int q) -> double {
      if (q >= W) return 0.0;
      int64_t a = start + q;
      float v = (a < t0) ? hist[W - (t0 - a)] : (cs ? cs[a - t0] : 0.f);
      return (double)v; // PeriodicFFT::feed(float), apply_rectangular_window (periodicfft.cpp:228)
    };
    if (cs && start >= t0) { // the frame lies inside the block (W = 2 S is even)
      const float *w0 = cs + (start - t0);
      ra_fft_fused(z, M, log2M, b.vc_tw_half, tid, T, [&](int k) {
        const bool in = 2 * k < W; // zero padding as a multiplication, like the modulator's loader
        const int i0 = in ? 2 * k : W - 2;
        const float v0 = w0[i0], v1 = w0[i0 + 1];
        const double keep = in ? 1.0 : 0.0;
        return make_double2((double)v0 * keep, (double)v1 * keep);
      });
    } else {
      ra_fft_fused(z, M, log2M, b.vc_tw_half, tid, T, [&](int k) { return make_double2(sample(2 * k), sample(2 * k + 1)); });
    }
  }

  // band amplitudes in force when this frame fires: the latest modulator frame with fire_j <= fire_i (it is fed first, :782-790)
  const double *amp = nullptr;
  {
    int64_t a = fire - ((int64_t)p.voc_Wm - 1);
    if (a >= 0) {
      int64_t j = a / S;
      if (j >= j_lo && j - j_lo < n_mod) amp = b.voc_amp + ((size_t)s * b.voc_mod_cap + (j - j_lo)) * B;
      else amp = b.voc_amp_prev + (size_t)s * B;
    }
  }
  auto gain = [&](int k) -> double {
    int band = b.voc_car_band[k];
    return (band >= 0 && amp) ? amp[band] : 0.0;
  };
  // forward split, modulate, and the inverse transform's pre-pass on the pair (k, M-k).  The inverse runs as conj(FFT(conj(.))) through
  // the same fused forward passes: conjugation is exact, so the bits equal those of a conjugated-twiddle inverse network.
  for (int k = tid; k <= (M >> 1); k += T) {
    const int km = M - k; // 1..M ; km == M stands for the Nyquist bin (pairs with k = 0)
    double2 zk = Z(k), zm = Z(km % M);
    double2 Xk = ra_real_split_bin(zk, zm, b.vc_tw_full[k]);
    double2 Xm = ra_real_split_bin(zm, zk, b.vc_tw_full[km]); // for k = 0 this is X[M] (Nyquist)
    double gk = gain(k), gm = gain(km);
    Xk.x *= gk; Xk.y *= gk;
    Xm.x *= gm; Xm.y *= gm;
    // z'[k] = (X[k] + conj X[M-k]) + i (X[k] - conj X[M-k]) conj(W^k)
    auto pre = [&](double2 xa, double2 xb, double2 w) -> double2 { // xa = X[q], xb = X[M-q], w = tw_full[q]
      double er = xa.x + xb.x, ei = xa.y - xb.y;
      double dr = xa.x - xb.x, di = xa.y + xb.y;
      double o_r = dr * w.x + di * w.y;   // (dr + i di) * (w.x - i w.y)
      double o_i = di * w.x - dr * w.y;
      return make_double2(er - o_i, -(ei + o_r)); // conjugated for the forward passes below
    };
    zin[k] = pre(Xk, Xm, b.vc_tw_full[k]);
    if (k != 0 && km != k) zin[km] = pre(Xm, Xk, b.vc_tw_full[km]);
  }
  __syncthreads();
  ra_fft_fused(z, M, log2M, b.vc_tw_half, tid, T, [&](int k) { return zin[k]; });
  const double scale = 1.0 / (double)L;
  double *dst = b.voc_sig + ((size_t)s * b.voc_car_cap + ib) * W;
  for (int q = tid; q < W; q += T) {
    double2 v = Z(q >> 1);
    dst[q] = ((q & 1) ? -v.y : v.x) * scale;
  }
  __syncthreads(); // both buffers are reused by the next frame
  }
}

// ---- K4c ------------------------------------------------------------------------------------------
__global__ void ra_voc_mix_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, const float *__restrict__ carrier,
                                  const double *__restrict__ carrier_d, int64_t t0, int64_t n, int64_t i_lo, int64_t n_car) {
  const int s = blockIdx.y;
  const int S = p.voc_S, W = 2 * p.voc_S;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = t0 + r;
    double y = 0.0;
    if (a >= W - 1) {
      const int64_t i = (a - (W - 1)) / S;
      const int idx = (int)(a - ((int64_t)W - 1 + i * S));
      auto sig = [&](int64_t f) -> const double * { // frame f's 2S samples, or null (= zeros) before the first frame
        if (f < 0) return nullptr;
        if (f >= i_lo) return b.voc_sig + ((size_t)s * b.voc_car_cap + (f - i_lo)) * W;
        return b.voc_sig_prev + ((size_t)s * 2 + (f - (i_lo - 2))) * W; // i_lo-2 -> slot 0, i_lo-1 -> slot 1
      };
      const double *nw = sig(i), *od = sig(i - 1);
      const double nm = b.voc_xf_new[idx], om = 1.0 - nm;
      y = (nw ? nw[idx] : 0.0) * nm + (od ? od[idx + S] : 0.0) * om;
    }
    const double m = voice ? (double)voice[(size_t)s * n + r] : 0.0;
    // the carrier as the compute lambda holds it (vocoder.cpp:777-803): a double — the carrier synth's sum, or the job's float sample widened
    const double c = carrier_d ? carrier_d[(size_t)s * n + r] : carrier ? (double)carrier[(size_t)s * n + r] : 0.0;
    b.voc_mix[(size_t)s * n + r] = (double)p.vol_mod * m + (double)p.vol_car * c + (double)p.vol_voc * y;
  }
}

// ---- K4d ------------------------------------------------------------------------------------------
// one workgroup per stream; LDS staging so that in-place shifts are safe
__global__ void ra_voc_roll_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, const float *__restrict__ carrier, int64_t n, int64_t i_lo,
                                   int64_t n_car, int64_t n_mod) {
  const int s = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int W = 2 * p.voc_S, Wm = p.voc_Wm, B = p.voc_bands;
  // staging buffer of max(2 W doubles, Wm floats): LDS, or (long strides / windows) this stream's part of the HBM scratch
  double *tmp = b.voc_roll_tmp ? b.voc_roll_tmp + (size_t)s * b.voc_roll_slot : reinterpret_cast<double *>(ra_smem_voc);
  // carrier frames: new prev = frames (i_hi - 1, i_hi) where i_hi = i_lo + n_car - 1
  if (n_car > 0) {
    double *prev = b.voc_sig_prev + (size_t)s * 2 * W;
    const double *blk = b.voc_sig + (size_t)s * b.voc_car_cap * W;
    for (int q = tid; q < 2 * W; q += T) {
      int slot = q / W, off = q - slot * W;
      int64_t f = i_lo + n_car - 2 + slot;     // absolute frame wanted in this slot
      double v;
      if (f >= i_lo) v = blk[(size_t)(f - i_lo) * W + off];
      else v = prev[(size_t)(f - (i_lo - 2)) * W + off]; // f == i_lo - 1 -> old slot 1
      tmp[q] = v;
    }
    __syncthreads();
    for (int q = tid; q < 2 * W; q += T) prev[q] = tmp[q];
    __syncthreads();
  }
  if (n_mod > 0) {
    const double *src = b.voc_amp + ((size_t)s * b.voc_mod_cap + (n_mod - 1)) * B;
    for (int q = tid; q < B; q += T) b.voc_amp_prev[(size_t)s * B + q] = src[q];
  }
  // sample histories
  auto roll = [&](float *hist, const float *blk, int len) {
    float *ft = reinterpret_cast<float *>(tmp);
    for (int q = tid; q < len; q += T) {
      int64_t src = (int64_t)q + n;            // index into concat(hist, block)
      ft[q] = (src < len) ? hist[src] : (blk ? blk[src - len] : 0.f);
    }
    __syncthreads();
    for (int q = tid; q < len; q += T) hist[q] = ft[q];
    __syncthreads();
  };
  roll(b.voc_mod_hist + (size_t)s * Wm, voice ? voice + (size_t)s * n : nullptr, Wm);
  roll(b.voc_car_hist + (size_t)s * W, carrier ? carrier + (size_t)s * n : nullptr, W);
}

} // namespace

void ra_launch_vocoder(const RaParams &p, const RaBuffers &b, const float *d_voice, const float *d_carrier, const double *carrier_d, int64_t t0, int64_t n,
                       hipStream_t st) {
  if (n <= 0) return;
  if (p.voc_kind == RA_VOCODER_BANDPASS) { ra_launch_voc_bank(p, b, d_voice, d_carrier, carrier_d, t0, n, st); return; }
  const int S = p.voc_S, W = 2 * S;
  RaFrameRange mod = ra_frames_firing(p.voc_Wm, S, t0, n);
  RaFrameRange car = ra_frames_firing(W, S, t0, n);
  static unsigned long long attr_done = 0;
  ra_once_per_device(attr_done, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voc_modulator_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voc_modulator_kernel<512, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define RA_FAST_ATTR(TT, NZ) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voc_modulator_fast_kernel<TT, NZ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    RA_FAST_ATTR(512, 5); RA_FAST_ATTR(512, 6); RA_FAST_ATTR(512, 7); RA_FAST_ATTR(512, 8); RA_FAST_ATTR(256, 5); RA_FAST_ATTR(256, 6); RA_FAST_ATTR(256, 7); RA_FAST_ATTR(256, 8);
#undef RA_FAST_ATTR
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voc_carrier_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voc_carrier_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voc_roll_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (mod.count > 0) {
    static const int fpw_env = [] { const char *e = getenv("RA_VOC_FPW"); return e ? atoi(e) : 0; }();
    // frames per workgroup: 8 when that still leaves every CU several workgroups (28.0 vs 33.1 ms per configs[2] step: a 512-thread
    // workgroup lives ~25 us and its successor's launch is not free), fewer for short blocks / few streams
    int fpw = 8;
    while (fpw > 1 && ((mod.count + fpw - 1) / fpw) * (int64_t)p.n_streams < 1024) fpw >>= 1;
    if (fpw_env > 0) fpw = fpw_env;
    const size_t lds = (size_t)ra_zsize(p.voc_Mm) * sizeof(double2);
    const bool big = lds > (size_t)RA_LDS_LIMIT_BYTES;
    if (big) fpw = (int)std::max<int64_t>(1, (mod.count * (int64_t)p.n_streams + RA_VOC_BIG_WORKGROUPS - 1) / RA_VOC_BIG_WORKGROUPS); // <= RA_VOC_BIG_WORKGROUPS + n_streams buffers
    const dim3 grid((unsigned)((mod.count + fpw - 1) / fpw), (unsigned)p.n_streams);
    // the product path: the pruned / fused transform (needs one radix-8 group per thread: M / 8 <= T, whole pairs in both window halves)
    static const bool slow_env = getenv("RA_VOC_UNFUSED") != nullptr; // tuning aid: r03's kernel (the analysis' unfused network)
    const int Tm = p.voc_Mm >= 4096 ? 512 : 256;
    if (!big && !slow_env && (p.voc_Wm & 3) == 0 && p.voc_Mm >= 128 && (p.voc_Mm >> 3) <= Tm && p.voc_bands <= 64) {
      const int G = p.voc_Mm >> 3, nz = (p.voc_Wm / 2 + G - 1) / G; // non-zero inputs of a first-pass group (5 .. 8: L < 2 W)
      const size_t lds_fast = lds + (size_t)(Tm / 64) * 64 * sizeof(double);
#define RA_FAST_MOD(TT, NZ) hipLaunchKernelGGL((ra_voc_modulator_fast_kernel<TT, NZ>), grid, dim3(TT), lds_fast, st, p, b, d_voice, t0, n, mod.lo, (int)mod.count, fpw)
      if (Tm == 512) { if (nz <= 5) RA_FAST_MOD(512, 5); else if (nz == 6) RA_FAST_MOD(512, 6); else if (nz == 7) RA_FAST_MOD(512, 7); else RA_FAST_MOD(512, 8); }
      else { if (nz <= 5) RA_FAST_MOD(256, 5); else if (nz == 6) RA_FAST_MOD(256, 6); else if (nz == 7) RA_FAST_MOD(256, 7); else RA_FAST_MOD(256, 8); }
#undef RA_FAST_MOD
    } else
    if (big) hipLaunchKernelGGL((ra_voc_modulator_kernel<512, true>), grid, dim3(512), 0, st, p, b, d_voice, t0, n, mod.lo, (int)mod.count, fpw);
    else if (p.voc_Mm >= 4096) hipLaunchKernelGGL((ra_voc_modulator_kernel<512, false>), grid, dim3(512), lds, st, p, b, d_voice, t0, n, mod.lo, (int)mod.count, fpw);
    else hipLaunchKernelGGL((ra_voc_modulator_kernel<256, false>), grid, dim3(256), lds, st, p, b, d_voice, t0, n, mod.lo, (int)mod.count, fpw);
  }
  if (car.count > 0) {
    static const int cfpw_env = [] { const char *e = getenv("RA_VOC_CARRIER_FPW"); return e ? atoi(e) : 0; }();
    int cfpw = 8; // frames per workgroup, as for the modulator
    while (cfpw > 1 && ((car.count + cfpw - 1) / cfpw) * (int64_t)p.n_streams < 4096) cfpw >>= 1;
    if (cfpw_env > 0) cfpw = cfpw_env;
    const int Mc = std::max(p.voc_Mc, 1);
    const size_t lds = ((size_t)ra_zsize(Mc) + Mc) * sizeof(double2);
    const bool big = lds > (size_t)RA_LDS_LIMIT_BYTES;
    if (big) cfpw = (int)std::max<int64_t>(1, (car.count * (int64_t)p.n_streams + RA_VOC_BIG_WORKGROUPS - 1) / RA_VOC_BIG_WORKGROUPS);
    const dim3 grid((unsigned)((car.count + cfpw - 1) / cfpw), (unsigned)p.n_streams);
    if (big) hipLaunchKernelGGL((ra_voc_carrier_kernel<256, true>), grid, dim3(256), 0, st, p, b, d_carrier, t0, n, car.lo, mod.lo, mod.count, (int)car.count, cfpw);
    else if (Mc <= 512) hipLaunchKernelGGL((ra_voc_carrier_kernel<64, false>), grid, dim3(64), lds, st, p, b, d_carrier, t0, n, car.lo, mod.lo, mod.count, (int)car.count, cfpw);
    else hipLaunchKernelGGL((ra_voc_carrier_kernel<256, false>), grid, dim3(256), lds, st, p, b, d_carrier, t0, n, car.lo, mod.lo, mod.count, (int)car.count, cfpw);
  }
  {
    unsigned gx = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(ra_voc_mix_kernel, dim3(gx, (unsigned)p.n_streams), dim3(256), 0, st, p, b, d_voice, d_carrier, carrier_d, t0, n, car.lo, car.count);
  }
  size_t lds = b.voc_roll_tmp ? 0 : std::max((size_t)2 * W * sizeof(double), (size_t)p.voc_Wm * sizeof(float));
  hipLaunchKernelGGL(ra_voc_roll_kernel, dim3((unsigned)p.n_streams), dim3(256), lds, st, p, b, d_voice, d_carrier, n, car.lo, car.count, mod.count);
}
