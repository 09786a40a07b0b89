// K2 — tracking: one wavefront per stream, frames in order, the 64 lanes cooperating inside a frame.
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
//
// Only the two-pointer matching of track_pitches is inherently sequential (one lane, tens of steps).  Everything
// else maps to lanes: the loudness order and the pitch order are rank sorts (stable: ties keep index order, the
// declared resolution of the reference's std::sort), event numbering and channel hand-out are ballot prefix counts
// (the k-th note-on candidate in loudness order gets the k-th free channel in channel order — what the reference's
// sequential first-free search yields, since channels only get taken inside one frame).
#include "ra_internal.h"
#include "ra_math.cuh"

extern __shared__ __align__(16) unsigned char ra_smem_trk[];

namespace {

__device__ inline double ra_readlane_f64(double v, int lane) { // lane is wave-uniform
  const long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffll), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Enveloped::setAngleIncrements, audioelement.h:218-225 with sound.functions.h:69-76
__device__ inline int min_change_for(float freq, int sample_rate) {
  float inc_f = 2 * (freq * 1.f) / sample_rate; // freq_to_angle_increment<float>, sound.functions.h:58-62
  double a = (double)inc_f;
  double period = (a == 0.) ? 1. : 2. / fabs(a);
  return (int)(0.5f + 2.5 * period);
}

__global__ void __launch_bounds__(64) ra_track_kernel(RaParams p, RaBuffers b, int64_t f_lo, int64_t fb0, int64_t n_frames_blk) {
  __builtin_amdgcn_s_setprio(3); // latency-bound chain: win issue arbitration against the throughput kernels running beside it
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  const int P = p.P, maxr = p.maxr;
  const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  // per pitch-list item
  double *pitch = reinterpret_cast<double *>(ra_smem_trk);        // [maxr]
  // per played note / channel (8-byte)
  double *pl_pitch = pitch + maxr;                                // [P]
  double *tmp_pitch = pl_pitch + P;                               // [P]
  int64_t *pl_noteid = reinterpret_cast<int64_t *>(tmp_pitch + P); // [P]
  int64_t *tmp_noteid = pl_noteid + P;                            // [P]
  int64_t *free_at = tmp_noteid + P;                              // [P]
  int64_t *cnote = free_at + P;                                   // [P] staged commands: note id
  // per item (4-byte)
  float *nf = reinterpret_cast<float *>(cnote + P);               // [maxr] new frequency (float, events.h:19-29)
  float *nv = nf + maxr;                                          // [maxr] new velocity
  int *nmc = reinterpret_cast<int *>(nv + maxr);                  // [maxr] min change duration of the new frequency
  int *ord = nmc + maxr;                                          // [maxr]
  int *pc = ord + maxr;                                           // [maxr] pitch_changes (-1 = none)
  // per played / channel (4-byte)
  int *pl_chan = pc + maxr;                                       // [P]
  int *tmp_chan = pl_chan + P;
  int *cont = tmp_chan + P;
  int *held = cont + P;
  int *minch = held + P;
  int *free_list = minch + P;
  int *ctype = free_list + P;
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
  RaPhaseClock clk(b.dbg_clk, lane == 0);
  unsigned long long sumA = 0, sumPlayed = 0;

  // The head of the next frame's list (one item per lane: pitch, event payload, loudness order) and its length are requested a frame ahead
  // into registers nothing else touches, and moved into the frame's LDS arrays at the END of the frame before (when they have long
  // arrived): a frame's chain of ~10 k cycles never waits for a global round trip.  (r02 / early r03 handed the registers over at the top of
  // the frame: the wait-count bookkeeping at the loop header then made every reuse of the first frame's load registers wait for the loads
  // just issued, and the length — wave-uniform, so the compiler wanted it in an SGPR at once — cost a round trip of its own.)
  struct Head { double pitch; float nf, nv; int nmc, ord, count; };
  int lane_zero = 0;
  asm volatile("" : "+v"(lane_zero)); // keeps the length's load a per-lane one: its value is needed a frame later
  auto fetch_head = [&](int64_t fbn) {
    Head h; h.pitch = 0.0; h.nf = 0.f; h.nv = 0.f; h.nmc = 0; h.ord = 0; h.count = 0;
    if (fbn < fb0 + n_frames_blk) {
      const size_t lb = ((size_t)s * b.frames_cap + fbn) * maxr;
      h.count = (b.pl_count + (size_t)s * b.frames_cap + fbn)[lane_zero];
      if (lane < maxr) { h.pitch = b.pl_pitch[lb + lane]; h.nf = b.pl_nf[lb + lane]; h.nv = b.pl_nv[lb + lane]; h.nmc = b.pl_nmc[lb + lane]; h.ord = b.pl_ord[lb + lane]; }
    }
    return h;
  };
  auto stage_head = [&](const Head &h) -> int { // into the arrays phases 1 and 5 read; returns the list's length
    if (lane < maxr) { pitch[lane] = h.pitch; nf[lane] = h.nf; nv[lane] = h.nv; nmc[lane] = h.nmc; ord[lane] = h.ord; }
    return __builtin_amdgcn_readfirstlane(h.count);
  };
  int A_next = stage_head(fetch_head(fb0));
  __syncthreads();

  for (int64_t fb = fb0; fb < fb0 + n_frames_blk; ++fb) { // frames fb0 .. of the block; scratch is indexed by fb
    const int64_t k_f = ra_fire_sample(p.N, p.H, f_lo + fb);
    const size_t lbase = ((size_t)s * b.frames_cap + fb) * maxr;
    const int A = A_next;
    const Head nx = fetch_head(fb + 1);

    // ---- phase 0: the frame's list with the event payload and the loudness order (lib.cpp:319-324, midi.h:254-264, algo.cpp:375-411): all of
    // it depends on the list alone and was computed by ra_track_prep_kernel, frame-parallel, off this chain; the first 64 items are staged ----
    for (int i = 64 + lane; i < A; i += 64) {
      pitch[i] = b.pl_pitch[lbase + i]; nf[i] = b.pl_nf[lbase + i]; nv[i] = b.pl_nv[lbase + i]; nmc[i] = b.pl_nmc[lbase + i]; ord[i] = b.pl_ord[lbase + i];
    }
    for (int i = lane; i < A; i += 64) pc[i] = -1;
    for (int i = lane; i < P; i += 64) { ctype[i] = RA_CMD_NONE; if (i < n_played) cont[i] = 0; }
    __syncthreads();

    clk.lap<0>(); sumA += A; sumPlayed += n_played;
    // ---- phase 1: track_pitches, algo.cpp:256-305 (sequential two-pointer scan) ----
    if (A <= 64 && n_played <= 64) {
      // The scan is a chain of comparisons; run by one lane out of LDS it costs two dependent LDS round trips per step (~200 cycles per
      // step of either pointer), and as one ballot step per item (r02) still ~290 cycles per item: broadcast, subtract, compare, mask, find
      // first, all waiting for each other and for the pointer.  Played pitches ascend (sort_by_current_pitch, algo.cpp:434-439), so for an
      // item with window [lo, hi] the notes below the window are the first L of them and the notes not above it the first U: both counts
      // depend on the item alone and every item lane takes them in a loop over the played notes whose steps do not wait for each other.
      // The reference's inner loop started at `it` stops at j = max(it, L); the note lies inside the window iff j < U.  What is left
      // of the chain is integer: two lane reads, a maximum, a compare and an add per item.
      const double my_played = lane < n_played ? pl_pitch[lane] : 0.0;
      const double my_pitch = lane < A ? pitch[lane] : 0.0;
      const double lo = my_pitch - p.max_track, hi = my_pitch + p.max_track;
      int L = 0, U = 0;
      for (int j = 0; j < n_played; ++j) {
        const double pj = ra_readlane_f64(my_played, j);
        L += (pj < lo) ? 1 : 0;
        U += (pj <= hi) ? 1 : 0;
      }
      // (no branches: a taken branch costs a lone wave more than the three instructions it would skip.  U <= n_played, so "inside the window"
      // implies "a note is left"; once `it` has reached n_played nothing matches any more, which is the reference's early exit.)
      const int Uz = lane < A ? U : 0;                 // lanes past the list never match: the loop may run over them
      int my_pc = -1, my_cont = 0;                     // lane = item: the note it continues; lane = played note: continued
      int it = 0;
#pragma unroll 4
      for (int idx = 0; idx < ((A + 3) & ~3); ++idx) {
        const int j = max(it, __builtin_amdgcn_readlane(L, idx));
        const bool m = j < __builtin_amdgcn_readlane(Uz, idx);
        it = j + (m ? 1 : 0);
        if (lane == (m ? idx : -1)) my_pc = j;
        if (lane == (m ? j : -1)) my_cont = 1;
      }
      if (lane < A) pc[lane] = my_pc;
      if (lane < n_played) cont[lane] = my_cont;
    } else if (lane == 0) {
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
    clk.lap<1>();
    // (phase 2, the loudness order, is ra_track_prep_kernel's)
    __syncthreads();

    clk.lap<2>();
    // ---- phase 3: note-offs for played notes that were not continued, lib.cpp:296-313 ----
    int n_off = 0;
    for (int j0 = 0; j0 < n_played; j0 += 64) {
      int j = j0 + lane;
      bool is = j < n_played && !cont[j];
      unsigned long long m = __ballot(is);
      if (is) {
        int c = pl_chan[j];
        ctype[c] = RA_CMD_OFF; cord[c] = n_off + __popcll(m & lower); cfreq[c] = 0.f; cvel[c] = 0.f; cnote[c] = pl_noteid[j];
        held[c] = 0;
        int r_eff = max(minch[c], R_env);     // getReleaseTime, audioelement.h:1002-1004 (latched at startReleased :836-841)
        free_at[c] = k_f + r_eff + 17;        // Done1 after r_eff steps, Done2 17 steps later (:738-750)
      }
      n_off += __popcll(m);
    }
    __syncthreads();

    // ---- phase 4: channels a note-on may take now, in channel order (tryAcquire loop, gen.crtp.h:397-405) ----
    int n_free = 0;
    for (int q0 = 0; q0 < P; q0 += 64) {
      int q = q0 + lane;
      bool is = q < P && !held[q] && free_at[q] <= k_f;
      unsigned long long m = __ballot(is);
      if (is) free_list[n_free + __popcll(m & lower)] = q;
      n_free += __popcll(m);
    }
    __syncthreads();

    // ---- phase 5: note-changes and note-ons in loudness order, lib.cpp:316-376 ----
    int ev_base = n_off, cand_base = 0, succ_base = 0;
    for (int o0 = 0; o0 < A; o0 += 64) {
      int oi = o0 + lane;
      bool valid = oi < A;
      int idx = valid ? ord[oi] : 0;
      int pci = valid ? pc[idx] : -1;
      bool tracked = valid && pci >= 0;
      bool cand = valid && !tracked && !(nv[idx] <= 0);     // `if (volume <= 0) continue;`
      unsigned long long cm = __ballot(cand);
      int crank = cand_base + __popcll(cm & lower);
      bool succ = cand && crank < n_free;                   // else onDroppedNote (gen.crtp.h:407-409)
      unsigned long long em = __ballot(tracked || succ);
      int evi = ev_base + __popcll(em & lower);
      unsigned long long sm = __ballot(succ);
      int srank = succ_base + __popcll(sm & lower);
      if (tracked) {
        int c = pl_chan[pci];
        ctype[c] = RA_CMD_CHANGE; cord[c] = evi; cfreq[c] = nf[idx]; cvel[c] = nv[idx]; cnote[c] = pl_noteid[pci];
        minch[c] = nmc[idx];
        pl_pitch[pci] = pitch[idx];
      } else if (succ) {
        int c = free_list[crank];
        int64_t noteid = st.next_noteid + crank + 1;        // ++next_noteid for every attempted note-on, lib.cpp:350
        ctype[c] = RA_CMD_ON | (srank << 8); cord[c] = evi; cfreq[c] = nf[idx]; cvel[c] = nv[idx]; cnote[c] = noteid;
        held[c] = 1;
        minch[c] = nmc[idx];
        int slot = n_played + srank;
        pl_pitch[slot] = pitch[idx]; pl_noteid[slot] = noteid; pl_chan[slot] = c; cont[slot] = 1;
      }
      cand_base += __popcll(cm); ev_base += __popcll(em); succ_base += __popcll(sm);
    }
    st.next_noteid += cand_base;
    st.dropped_note_on += cand_base - succ_base;
    const int n_new = n_played + succ_base;
    __syncthreads();

    // ---- phase 6: remove_dead_notes + sort_by_current_pitch (both stable), algo.cpp:415-440 ----
    int kept = 0;
    if (n_new <= 64) { // one note per lane: ranks among the kept notes from v_readlane broadcasts, walking the set bits of the kept mask
      const bool is = lane < n_new && cont[lane];
      const double pj = lane < n_new ? pl_pitch[lane] : 0.0;
      const unsigned long long km = __ballot(is);
      int r = 0;
      for (unsigned long long m = km; m; m &= m - 1) { // (r03: a straight loop over all lanes with the ties counted apart measured slower)
        const int i = __ffsll((long long)m) - 1;
        const double pi = ra_readlane_f64(pj, i);
        r += (pi < pj || (pi == pj && i < lane)) ? 1 : 0;
      }
      if (is) { tmp_pitch[r] = pj; tmp_noteid[r] = pl_noteid[lane]; tmp_chan[r] = pl_chan[lane]; }
      kept = __popcll(km);
    } else
    for (int j0 = 0; j0 < n_new; j0 += 64) {
      int j = j0 + lane;
      bool is = j < n_new && cont[j];
      if (is) {
        double pj = pl_pitch[j];
        int r = 0;
        for (int i = 0; i < n_new; ++i) { double pi = pl_pitch[i]; r += (cont[i] && (pi < pj || (pi == pj && i < j))) ? 1 : 0; }
        tmp_pitch[r] = pj; tmp_noteid[r] = pl_noteid[j]; tmp_chan[r] = pl_chan[j];
      }
      kept += __popcll(__ballot(is));
    }
    __syncthreads();
    n_played = kept;
    for (int i = lane; i < n_played; i += 64) { pl_pitch[i] = tmp_pitch[i]; pl_noteid[i] = tmp_noteid[i]; pl_chan[i] = tmp_chan[i]; }

    // the next frame's list head (requested at the top of this frame) into the list arrays, which phase 5 was the last to read — before
    // this frame's command stores are issued: a wait for the loads would wait for those stores too
    A_next = stage_head(nx);
    if (lane == 0) b.cmd_non[(size_t)s * b.frames_cap + fb] = succ_base;
    RaCmd *out = b.cmd + ((size_t)s * b.frames_cap + fb) * p.Ppad;
    for (int i = lane; i < p.Ppad; i += 64) {
      RaCmd c;
      if (i < P && ctype[i] != RA_CMD_NONE) { c.type = ctype[i]; c.ord = cord[i]; c.freq = cfreq[i]; c.vel = cvel[i]; c.noteid = cnote[i]; }
      else { c.type = RA_CMD_NONE; c.ord = 0; c.freq = 0.f; c.vel = 0.f; c.noteid = 0; }
      out[i] = c;
    }
    __syncthreads();
    clk.lap<3>();
  }
  clk.flush(48);
  if (clk.dst) { atomicAdd(clk.dst + 56, sumA); atomicAdd(clk.dst + 57, sumPlayed); }

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

// ---- K2a: what tracking needs of a frame's list and the list alone determines — the event payload of every item (lib.cpp:319-324,
// midi.h:254-264, the power as in oracle_math.hpp: events are compared bit for bit) and the frame's loudness order (stable rank sort by key,
// descending: order_pitches_by_perceived_loudness, algo.cpp:375-411, ties in index order) — one wave per frame, RA_PREP_FRAMES frames per wave.
// (r03: these were phases 0 and 2 of the tracking chain, 8.5 k of its 24 k cycles per frame.)
#define RA_PREP_WAVES 4
#define RA_PREP_FRAMES 4
__global__ void __launch_bounds__(64 * RA_PREP_WAVES) ra_track_prep_kernel(RaParams p, RaBuffers b, int64_t fb0, int nfb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = blockIdx.y, maxr = p.maxr;
  float *keys = reinterpret_cast<float *>(ra_smem_trk) + (size_t)wave * maxr; // [maxr] per wave
  for (int q = 0; q < RA_PREP_FRAMES; ++q) {
    const int f = (blockIdx.x * RA_PREP_WAVES + wave) * RA_PREP_FRAMES + q;
    if (f >= nfb) break; // wave-uniform
    const size_t fi = (size_t)s * b.frames_cap + fb0 + f, base = fi * maxr;
    const int A = min(b.pl_count[fi], maxr);
    for (int i = lane; i < A; i += 64) {
      const double pt = b.pl_pitch[base + i], v = b.pl_vol[base + i];
      const float fr = (float)((double)440.f * ra_pow(p.half_tone_ratio, pt - 69.0));
      b.pl_nf[base + i] = fr;
      b.pl_nv[base + i] = (float)((double)p.analysis_volume * v / (double)0.1f);
      b.pl_nmc[base + i] = min_change_for(fr, p.sample_rate);
      keys[i] = b.pl_key[base + i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < A; i += 64) {
      const float k = keys[i];
      int r = 0;
      for (int j = 0; j < A; ++j) { const float kj = keys[j]; r += (kj > k || (kj == k && j < i)) ? 1 : 0; }
      b.pl_ord[base + r] = i;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // keys are rewritten by the next frame
  }
}

} // namespace

size_t ra_track_lds_bytes(const RaParams &p) { return (size_t)p.maxr * (8 + 5 * 4) + (size_t)p.P * (6 * 8 + 10 * 4) + 64; }

void ra_launch_track_prep(const RaParams &p, const RaBuffers &b, int64_t fb0, int64_t n_frames_blk, hipStream_t st) {
  if (n_frames_blk <= 0) return;
  const size_t lds = (size_t)RA_PREP_WAVES * p.maxr * sizeof(float);
  static unsigned long long attr_done = 0;
  ra_once_per_device(attr_done, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_track_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RA_LDS_LIMIT_BYTES);
  });
  const int per_wg = RA_PREP_WAVES * RA_PREP_FRAMES;
  hipLaunchKernelGGL(ra_track_prep_kernel, dim3((unsigned)((n_frames_blk + per_wg - 1) / per_wg), (unsigned)p.n_streams), dim3(64 * RA_PREP_WAVES), lds, st, p, b, fb0,
                     (int)n_frames_blk);
}

void ra_launch_track(const RaParams &p, const RaBuffers &b, int64_t f_lo, int64_t fb0, int64_t n_frames_blk, hipStream_t st) {
  if (n_frames_blk <= 0) return;
  const size_t lds = ra_track_lds_bytes(p);
  static unsigned long long attr_done = 0; // above the 64 KB default from max_pitches_per_frame ~ 1500 on (ra_create checks the 160 KB limit)
  ra_once_per_device(attr_done, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_track_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RA_LDS_LIMIT_BYTES);
  });
  hipLaunchKernelGGL(ra_track_kernel, dim3((unsigned)p.n_streams), dim3(64), lds, st, p, b, f_lo, fb0, n_frames_blk);
}
