// K5 — the main.resynth driver's renderer (reference include/resynth.hpp:4-100): a pool of
// Enveloped<VolumeAdjusted<SineOscillatorAlgo<double>>, AHDSREnvelope> elements driven note by note.
//
// The host replays the reference's pool policy (first element that is not RT-active, else a new one, :19-27);
// an element's life is deterministic (key press, release after `delay` steps, release time, 17 steps of
// EnvelopeDone1), so each pool slot gets a fixed, time-ordered note list.  One lane = one pool slot, which keeps
// the oscillator phase carried from note to note exactly like the reused C++ object does.  A wavefront skips
// tiles in which none of its 64 slots sounds.
#include "ra_chan.cuh"

namespace {

// phase_io (optional, [n_slots]): oscillator phasor each slot starts from / ends with (default: a fresh element, 1 + 0i)
__global__ void __launch_bounds__(64) ra_notes_kernel(const RaNoteEv *__restrict__ ev, const int *__restrict__ slot_begin, int n_slots, RaEnvParams ep,
                                                      int64_t T, double *__restrict__ partial, double2 *__restrict__ phase_io) {
  __shared__ double tile[16][65];
  const int l = threadIdx.x, g = blockIdx.x;
  const int slot = g * 64 + l;
  int next = 0, end = 0;
  if (slot < n_slots) { next = slot_begin[slot]; end = slot_begin[slot + 1]; }
  Chan c;
  chan_init(c);
  if (phase_io && slot < n_slots) { c.cre = phase_io[slot].x; c.cim = phase_io[slot].y; }
  const double max_filter_inc = 1.7976931348623157e308; // BaseVolumeAdjusted::max_filter_increment default, audioelement.h:1225
  double *dst = partial + (size_t)g * T;
  for (int64_t t = 0; t < T; t += 16) {
    const int m = (int)min((int64_t)16, T - t);
    const bool busy = rt_active(c) || (next < end && ev[next].start < t + m);
    if (!__any(busy)) { // nothing sounds in this tile for any of the 64 slots
      if (l < m) dst[t + l] = 0.0;
      continue;
    }
    for (int i = 0; i < m; ++i) {
      if (next < end && ev[next].start == t + i) { // resynth.hpp:83-96
        const RaNoteEv e = ev[next++];
        chan_forget_past_signals(c);
        chan_set_angle_increments(c, e.inc, 1);
        c.target = e.vol; c.flags |= FL_HAS_TARGET;
        env_on_key_pressed(c, 0);
        env_on_key_released(c, e.delay);
      }
      double v = 0.0;
      if (rt_active(c)) { // recordFrame, resynth.hpp:38-52 (steps, then adds imag() unconditionally)
        chan_step(c, ep, 1.0, max_filter_inc, 1);
        v = (c.vol * (c.alias * c.cim)) * c.value; // Enveloped::imag :231 over VolumeAdjusted::imag :1128
      }
      tile[i][l] = v;
    }
    __syncthreads();
    {
      const int fr = l & 15, q = l >> 4;
      double acc = 0.0;
      if (fr < m) for (int k = q * 16; k < q * 16 + 16; ++k) acc += tile[fr][k];
      acc += __shfl_down(acc, 32);
      acc += __shfl_down(acc, 16);
      if (l < 16 && fr < m) dst[t + fr] = acc;
    }
    __syncthreads();
  }
  if (phase_io && slot < n_slots) phase_io[slot] = make_double2(c.cre, c.cim);
}

__global__ void ra_notes_sum_kernel(const double *__restrict__ partial, int groups, int64_t T, double *__restrict__ out) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int g = 0; g < groups; ++g) acc += partial[(size_t)g * T + t];
    out[t] = acc;
  }
}

} // namespace

void ra_launch_notes(const RaNoteEv *d_ev, const int *d_slot_begin, int n_slots, RaEnvParams ep, int64_t T, double *d_partial, double *d_out,
                     double2 *d_phase_io, hipStream_t st) {
  if (T <= 0) return;
  int groups = (n_slots + 63) / 64;
  if (groups < 1) groups = 1;
  hipLaunchKernelGGL(ra_notes_kernel, dim3((unsigned)groups), dim3(64), 0, st, d_ev, d_slot_begin, n_slots, ep, T, d_partial, d_phase_io);
  unsigned gx = (unsigned)std::min<int64_t>((T + 255) / 256, 8192);
  hipLaunchKernelGGL(ra_notes_sum_kernel, dim3(gx), dim3(256), 0, st, d_partial, groups, T, d_out);
}
