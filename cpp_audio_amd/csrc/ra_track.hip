// K2 — tracking: one wavefront per stream, frames in order.
//
// Replaces the stateful half of RtResynth::step (reference source/rt.resynth.lib.cpp:1710-1758):
// track_pitches (rt.resynth.lib.algo.cpp:256-305), order_pitches_by_perceived_loudness (:375-411),
// synthesize_sounds (lib.cpp:264-382) with the channel acquisition of ImplCRTP::onEvent
// (include/gen.crtp.h:393-412), remove_dead_notes + sort_by_current_pitch (algo.cpp:415-440).
//
// The reference decides "is a channel free" by CAS on the envelope state (audioelement.h:363-369).
// An envelope's life after its note-off is deterministic (release time, then 17 steps through
// EnvelopeDone1, audioelement.h:732-750), so this kernel keeps a per-channel `free_at` sample index
// instead of reading synth state, which lets the whole block be tracked before one sample is rendered.
// K3 cross-checks every acquisition against the real envelope (RA_ERR_BOOKKEEPING).
#include "ra_internal.h"

extern __shared__ __align__(16) unsigned char ra_smem_trk[];

namespace {

// Enveloped::setAngleIncrements, audioelement.h:218-225 with sound.functions.h:69-76
__device__ inline int min_change_for(float freq, int sample_rate, float mult) {
  float inc_f = 2 * (freq * mult) / sample_rate; // freq_to_angle_increment<float>, sound.functions.h:58-62
  double a = (double)inc_f;
  double period = (a == 0.) ? 1. : 2. / fabs(a);
  return (int)(0.5f + 2.5 * period);
}

__global__ void __launch_bounds__(64) ra_track_kernel(RaParams p, RaBuffers b, int64_t f_lo, int64_t n_frames_blk) {
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  const int P = p.P, maxr = p.maxr;

  double *pitch = reinterpret_cast<double *>(ra_smem_trk);       // [maxr]
  double *vol = pitch + maxr;                                    // [maxr]
  double *pl_pitch = vol + maxr;                                 // [P] played: current pitch
  int64_t *pl_noteid = reinterpret_cast<int64_t *>(pl_pitch + P); // [P]
  int64_t *free_at = pl_noteid + P;                              // [P]
  int64_t *cnote = free_at + P;                                  // [P] staged commands: note id
  float *key = reinterpret_cast<float *>(cnote + P);             // [maxr]
  int *ord = reinterpret_cast<int *>(key + maxr);                // [maxr]
  int *pc = ord + maxr;                                          // [maxr]  pitch_changes (-1 = none)
  int *pl_chan = pc + maxr;                                      // [P]
  int *cont = pl_chan + P;                                       // [P]
  int *held = cont + P;                                          // [P]
  int *minch = held + P;                                         // [P]
  int *ctype = minch + P;                                        // [P] staged commands
  int *cord = ctype + P;
  float *cfreq = reinterpret_cast<float *>(cord + P);
  float *cvel = cfreq + P;

  RaTrackState st = b.track[s];
  int n_played = st.n_played;
  for (int i = lane; i < P; i += 64) {
    free_at[i] = b.chan_free_at[(size_t)s * P + i];
    held[i] = b.chan_held[(size_t)s * P + i];
    minch[i] = b.chan_minch[(size_t)s * P + i];
    if (i < n_played) {
      RaPlayed pn = b.played[(size_t)s * P + i];
      pl_pitch[i] = pn.pitch; pl_noteid[i] = pn.noteid; pl_chan[i] = pn.channel;
    }
  }
  __syncthreads();

  const int R_env = p.envR;

  for (int64_t fb = 0; fb < n_frames_blk; ++fb) {
    const int64_t k_f = ra_fire_sample(p.N, p.H, f_lo + fb);
    const size_t lbase = ((size_t)s * b.frames_cap + fb) * maxr;
    const int A = b.pl_count[(size_t)s * b.frames_cap + fb];
    for (int i = lane; i < A; i += 64) { pitch[i] = b.pl_pitch[lbase + i]; vol[i] = b.pl_vol[lbase + i]; key[i] = b.pl_key[lbase + i]; }
    for (int i = lane; i < P; i += 64) ctype[i] = RA_CMD_NONE;
    __syncthreads();

    if (lane == 0) {
      // track_pitches, algo.cpp:256-305
      for (int i = 0; i < A; ++i) pc[i] = -1;
      for (int j = 0; j < n_played; ++j) cont[j] = 0;
      {
        int it = 0;
        for (int idx = 0; idx < A; ++idx) {
          double np = pitch[idx];
          for (; it != n_played; ++it) {
            if (pl_pitch[it] < np - p.max_track) continue;
            if (pl_pitch[it] <= np + p.max_track) { pc[idx] = it; cont[it] = 1; ++it; }
            break;
          }
        }
      }
      // order by key, descending, equal keys by ascending index (insertion sort = stable)
      for (int i = 0; i < A; ++i) {
        float k = key[i];
        int j = i - 1;
        while (j >= 0 && key[ord[j]] < k) { ord[j + 1] = ord[j]; --j; }
        ord[j + 1] = i;
      }
      int n_ev = 0;
      // note-offs, lib.cpp:296-313
      for (int j = 0; j < n_played; ++j) {
        if (!cont[j]) {
          int c = pl_chan[j];
          ctype[c] = RA_CMD_OFF; cord[c] = n_ev++; cfreq[c] = 0.f; cvel[c] = 0.f; cnote[c] = pl_noteid[j];
          held[c] = 0;
          int r_eff = max(minch[c], R_env);     // getReleaseTime, audioelement.h:1002-1004 (latched at startReleased :836-841)
          free_at[c] = k_f + r_eff + 17;        // Done1 after r_eff steps, Done2 17 steps later (:738-750)
        }
      }
      // note-changes and note-ons in loudness order, lib.cpp:316-376
      int n_new = n_played;
      for (int oi = 0; oi < A; ++oi) {
        int idx = ord[oi];
        double new_pitch = pitch[idx];
        float new_freq = (float)((double)440.f * pow(p.half_tone_ratio, new_pitch - 69.0)); // midi.h:254-264
        float volume = (float)((double)p.analysis_volume * vol[idx] / (double)0.1f);       // lib.cpp:324
        int pci = pc[idx];
        if (pci >= 0) {
          int c = pl_chan[pci];
          ctype[c] = RA_CMD_CHANGE; cord[c] = n_ev++; cfreq[c] = new_freq; cvel[c] = volume; cnote[c] = pl_noteid[pci];
          minch[c] = min_change_for(new_freq, p.sample_rate, 1.f);
          pl_pitch[pci] = new_pitch;
        } else {
          if (volume <= 0) continue;
          ++st.next_noteid;
          int c = -1;
          for (int q = 0; q < P; ++q) if (!held[q] && free_at[q] <= k_f) { c = q; break; } // tryAcquire in channel order, gen.crtp.h:397-405
          if (c < 0) { ++st.dropped_note_on; continue; }
          ctype[c] = RA_CMD_ON; cord[c] = n_ev++; cfreq[c] = new_freq; cvel[c] = volume; cnote[c] = st.next_noteid;
          held[c] = 1;
          minch[c] = min_change_for(new_freq, p.sample_rate, 1.f);
          pl_pitch[n_new] = new_pitch; pl_noteid[n_new] = st.next_noteid; pl_chan[n_new] = c; cont[n_new] = 1;
          ++n_new;
        }
      }
      // remove_dead_notes (stable) + sort_by_current_pitch (stable insertion), algo.cpp:415-440
      int w = 0;
      for (int j = 0; j < n_new; ++j) {
        if (cont[j]) {
          if (w != j) { pl_pitch[w] = pl_pitch[j]; pl_noteid[w] = pl_noteid[j]; pl_chan[w] = pl_chan[j]; }
          ++w;
        }
      }
      n_played = w;
      for (int i = 1; i < n_played; ++i) {
        double kp = pl_pitch[i]; int64_t kn = pl_noteid[i]; int kc = pl_chan[i];
        int j = i - 1;
        while (j >= 0 && kp < pl_pitch[j]) { pl_pitch[j + 1] = pl_pitch[j]; pl_noteid[j + 1] = pl_noteid[j]; pl_chan[j + 1] = pl_chan[j]; --j; }
        pl_pitch[j + 1] = kp; pl_noteid[j + 1] = kn; pl_chan[j + 1] = kc;
      }
    }
    __syncthreads();
    n_played = __shfl(n_played, 0);

    RaCmd *out = b.cmd + ((size_t)s * b.frames_cap + fb) * p.Ppad;
    for (int i = lane; i < p.Ppad; i += 64) {
      RaCmd c;
      if (i < P && ctype[i] != RA_CMD_NONE) { c.type = ctype[i]; c.ord = cord[i]; c.freq = cfreq[i]; c.vel = cvel[i]; c.noteid = cnote[i]; }
      else { c.type = RA_CMD_NONE; c.ord = 0; c.freq = 0.f; c.vel = 0.f; c.noteid = 0; }
      out[i] = c;
    }
    __syncthreads();
  }

  for (int i = lane; i < P; i += 64) {
    b.chan_free_at[(size_t)s * P + i] = free_at[i];
    b.chan_held[(size_t)s * P + i] = held[i];
    b.chan_minch[(size_t)s * P + i] = minch[i];
    if (i < n_played) {
      RaPlayed pn; pn.pitch = pl_pitch[i]; pn.noteid = pl_noteid[i]; pn.channel = pl_chan[i]; pn.pad = 0;
      b.played[(size_t)s * P + i] = pn;
    }
  }
  if (lane == 0) { st.n_played = n_played; b.track[s] = st; }
}

} // namespace

void ra_launch_track(const RaParams &p, const RaBuffers &b, int64_t f_lo, int64_t n_frames_blk, hipStream_t st) {
  if (n_frames_blk <= 0) return;
  size_t lds = (size_t)p.maxr * (8 + 8 + 4 + 4 + 4) + (size_t)p.P * (8 + 8 + 8 + 4 * 8 + 8) + 64;
  hipLaunchKernelGGL(ra_track_kernel, dim3((unsigned)p.n_streams), dim3(64), lds, st, p, b, f_lo, n_frames_blk);
}
