// K6 / K7 — SURVEY.md §8 f2: the vocoder carrier synth and the band-pass bank vocoder.
//
// K6 ra_carrier_synth_kernel: SynthVocoderCarier (reference source/rt.resynth.lib.cpp:212-221) = ImplCRTP (include/gen.crtp.h:350-622)
//    over VocoderCarrierElement<double> (lib.cpp:21-52; initializer :137-196).  One workgroup per stream, lane = channel, samples in
//    order: the phase-controlled oscillators' discontinuous waveforms (saw / square / pulse, include/sound.functions.h:86-138) make
//    every bit of the accumulated angle audible — `angle += inc; if (angle > 2) angle -= 2` (Phased::step, audioelement.h:1452-1470)
//    has no closed form that reproduces its roundings, and a flipped edge sample is an error of twice the voice's volume.
//    What is constant in the reference's element and therefore not stepped here (the CPU restatement steps all of it):
//      * the six inner VolumeAdjusted filters: their targets are set once per note by the initializer and the first step copies the
//        target into the filter (audioelement.h:1201-1215), so volume == target for the whole note;
//      * the four Phased oscillators share one angle: same start angle (UnityGainMix::setStartAngle / synchronizeAngles,
//        :1940-1964), same increments, same steps;
//      * the pulse's "used" width / levels: reset by forgetPastSignals and re-latched by the initializer at every note-on.
//    The noise voice (soundBufferWrapperAlgo<NOISE>, audioelement.h:1507-1583) is an index into the white-noise sound buffer
//    (source/sound.cpp:3-21; uploaded once per context, ra_capi.hip: build_white_noise): setStartAngle / synchronizeAngles set it, every
//    step increments it modulo the buffer's length, imag() reads the buffer; its constant angleIncrements() = 0.5f (include/sound.h:256-258)
//    enters UnityGainMix::angleIncrements() (:1948-1958).  With volume 0 (the reference default) no buffer exists and the voice is skipped.
//
// K7 ra_voc_bank_kernel: Modulator<1, double> + Carrier<1, double> (source/rt.resynth.lib.vocoder.cpp:310-394, 667-732) inside the
//    Vocoder compute lambda (:763-812).  One wavefront per stream, lane = band, samples in order; per sample and band five one-pole
//    recurrences (modulator high-pass -> low-pass -> |.| -> envelope follower; carrier high-pass -> low-pass).  The band sum
//    `res += ponderation[i] * filtered` runs in band order, like the reference's loop, on a transposed 64-sample tile.
//    Same operations in the same order as the CPU restatement (no contraction): the vocoded signal is equal bit for bit.
#include "ra_chan.cuh"
#include "ra_lane.cuh"
#include <algorithm>
#include <cstdlib>

extern __shared__ __align__(16) unsigned char ra_smem_car[];

namespace {

// ---- include/sound.functions.h:86-138 ------------------------------------------------------------------------------------
__device__ inline double wave_square(double ang) { return (0.5 < ang && ang < 1.5) ? -1.0 : 1.0; }
__device__ inline double wave_pulse(double ang, double w, double high, double low) { return ang < w ? high : low; }
__device__ inline double wave_triangle(double ang) { return ang < 0.5 ? 2.0 * ang : ang < 1.5 ? 2.0 + -2.0 * ang : -4.0 + 2.0 * ang; }
__device__ inline double wave_saw(double ang) { return ang <= 1.0 ? ang : -2.0 + ang; }
__device__ inline double phase_to_normal_form(double phase) { // audioelement.h:418-428
  double nf = fmod(phase, 2.0);
  if (nf < 0) nf += 2.0;
  return nf;
}

struct CarX { double ph_angle, ph_inc, ph_alias; int nz; }; // the Phased oscillators (one shared angle, see above); the noise voice's index
// soundBufferWrapperAlgo::setStartAngle (audioelement.h:1544-1557), a in [-1, 1]
__device__ inline int noise_start_index(double a, int size) {
  int index = (int)(((a + 1.) * (double)size * 0.5) + 0.5);
  if (index < 0) index = 0;
  return index - 1; // "because index will be incremented in step before being used"
}
// soundBufferWrapperAlgo::step (:1568-1574), k >= 0 times: ++index; if (index >= size) index = 0
__device__ inline int noise_advance(int index, int k, int size) { return k > 0 ? (int)(((long long)index + k) % size) : index; }

// FreqCtrl_::angleIncrements (audioelement.h:3117-3124) over UnityGainMix::angleIncrements (:1948-1958): noise 0.5f, the four Phased
// increments, the sine's
__device__ inline double car_angle_increments(const Chan &c, const CarX &x) {
  double ai = (double)0.5f;
  ai = fmin(ai, fabs(x.ph_inc));
  ai = fmin(ai, fabs(c.inc));
  if (ai != 0.) return ai;
  return c.from;
}

constexpr int CT = 16; // samples per tile

__global__ void __launch_bounds__(RA_MAX_CHANNELS) ra_carrier_synth_serial_kernel(RaParams p, RaBuffers b, int64_t t0, int64_t n) {
  const int s = blockIdx.x, l = threadIdx.x, Ppad = p.car_Ppad, P = p.car_P;
  double *tile = reinterpret_cast<double *>(ra_smem_car);            // [Ppad][CT + 1]
  double *xch = tile + (size_t)Ppad * (CT + 1);                      // [4] broadcast slots
  int *xi = reinterpret_cast<int *>(xch + 4);                        // [4]
  Chan c;
  CarX x;
  double *sd = b.car_sd + (size_t)s * SD_COUNT * Ppad;
  int *si = b.car_si + (size_t)s * SI_COUNT * Ppad;
  double *ext = b.car_ext + (size_t)s * CX_COUNT * Ppad;
  load_chan(c, sd, si, Ppad, l);
  x.ph_angle = ext[CX_PH_ANGLE * Ppad + l]; x.ph_inc = ext[CX_PH_INC * Ppad + l]; x.ph_alias = ext[CX_PH_ALIAS * Ppad + l];
  x.nz = (int)ext[CX_NZ_INDEX * Ppad + l];
  int64_t noteid = b.car_noteid[(size_t)s * Ppad + l];
  int dropped = 0;
  const RaEnvParams ep = {p.envA, p.envH, p.envD, p.envR, p.SMinusOne};
  const uint64_t seed = p.seed + (uint64_t)s;
  // VocoderCarrierElementInitializer (lib.cpp:155-171): volume targets (float -> T) and PulseTrainAlgo_::setPulseWidth (audioelement.h:1701-1719)
  const double *nzb = b.car_noise; // null: noise volume 0
  const int nzn = b.car_noise_len;
  const double v_noise = nzb ? (double)p.car_vol[0] : 0.0;
  const double v_saw = (double)p.car_vol[1], v_tri = (double)p.car_vol[2], v_sq = (double)p.car_vol[3],
               v_sine = (double)p.car_vol[4], v_pulse = (double)p.car_vol[5];
  double pw = (double)p.car_pulse_width;
  pw = fmax(0., pw); pw = fmin(2., pw);
  const double p_high = 0.5 * (2. - pw), p_low = p_high - 1.;

  const RaCarEv *ev = b.car_ev;
  int e = b.car_ev_begin[s];
  const int e_end = b.car_ev_begin[s + 1];
  double *out_d = b.car_gen_d + (size_t)s * n;
  float *out_f = b.car_gen_f + (size_t)s * n;

  RaPhaseClock clk(b.dbg_clk, l == 0);
  int64_t t = t0;
  const int64_t t_end = t0 + n;
  // the next event's frame stays in a register: a global load per tile would have to wait for the tile's output stores to drain
  constexpr int64_t kNever = 0x7fffffffffffffffll;
  int64_t next_ev = (e < e_end) ? ev[e].sample : kNever;
  while (t < t_end) {
    if (next_ev <= t) { // the one-shots queued for this frame run before its step (simple_audio_context.h:74-76)
      const RaCarEv q = ev[e];
      ++e;
      next_ev = (e < e_end) ? ev[e].sample : kNever;
      if (l == 0) { xi[0] = 0x7fffffff; xi[1] = 0x7fffffff; }
      __syncthreads();
      if (q.type == 0) { // ---- onEvent(NoteOn), gen.crtp.h:389-449, and its one-shot :455-527 (run at once: same frame) ----
        if (l < P && c.state == ENV_DONE2) atomicMin(&xi[0], l); // tryAcquire: the first finished channel, :397-405
        __syncthreads();
        const int sel = xi[0];
        if (sel == 0x7fffffff) { if (l == 0) ++dropped; } // onDroppedNote
        else {
          if (l == sel) {
            c.state = ENV_SOON;
            noteid = q.noteid;                                        // channels.corresponding(*channel) = e.noteid
            chan_forget_past_signals(c);                              // envelope, control, volumes (the mix's have no state here)
            c.target = 1.0 * (double)q.vel; c.flags |= FL_HAS_TARGET; // Element::baseVolume (BaseVolumeDef::One) * velocity, :425
            c.cur = 0;                                                // initializer: ctrl.setup(100, LINEAR), no f_result yet (lib.cpp:157-158)
            const float inc_f = 2 * (q.freq * 1.f) / p.sample_rate;   // setupAudioElement, gen.sine.h:9-21
            chan_set_angle_increments(c, (double)inc_f, 0);
            xch[0] = car_angle_increments(c, x);
          }
          __syncthreads();
          // setPhase (gen.crtp.h:128-156): the first RT-active channel with an equal angleIncrements() donates its angles
          if (l != sel && l < P && rt_active(c) && car_angle_increments(c, x) == xch[0]) atomicMin(&xi[1], l);
          __syncthreads();
          const int src = xi[1];
          if (src != 0x7fffffff && l == src) { xch[1] = x.ph_angle; xch[2] = c.cre; xch[3] = c.cim; xi[2] = x.nz; }
          __syncthreads();
          if (l == sel) {
            if (src != 0x7fffffff) {
              x.ph_angle = phase_to_normal_form(xch[1]); // Phased::synchronizeAngles -> setStartAngle, audioelement.h:1433-1439
              c.cre = xch[2]; c.cim = xch[3];            // SineOscillatorAlgo::synchronizeAngles :2397-2399
              x.nz = xi[2];                              // soundBufferWrapperAlgo::synchronizeAngles :1532-1537
            } else {
              const float a = (p.phase_policy == RA_PHASE_HASH) ? hash_uniform_pm1(seed, (uint64_t)q.noteid, 2) : 0.f;
              x.ph_angle = phase_to_normal_form((double)a);
              double sn, cs;
              sincos(M_PI * (double)a, &sn, &cs);          // setStartAngle :2401-2403
              c.cre = cs; c.cim = sn;
              if (nzb) x.nz = noise_start_index((double)a, nzn);
            }
            env_on_key_pressed(c, q.delay);
          }
        }
      } else { // ---- NoteOff one-shot, gen.crtp.h:541-588: the first channel that last played this note id ----
        if (l < P && noteid == q.noteid) atomicMin(&xi[0], l);
        __syncthreads();
        if (l == xi[0] && env_can_release_now(c, q.delay)) env_on_key_released(c, q.delay);
      }
      __syncthreads();
      continue;
    }
    const int64_t seg_end = min(t_end, next_ev);
    if (!__syncthreads_or(l < P && rt_active(c))) { // nothing sounds until the next event: silence
      for (int64_t r = t + l; r < seg_end; r += blockDim.x) { out_d[r - t0] = 0.0; out_f[r - t0] = 0.f; }
      t = seg_end;
      continue;
    }
    const int m = (int)min((int64_t)CT, seg_end - t);
    double *mine = tile + (size_t)l * (CT + 1);
    clk.lap<0>();
    // A tile in which every sounding channel just holds its note — envelope at its sustain level with no release falling due, control
    // and oscillators at the note's increment, volume at its target — is the common case: its step is the oscillators and the mix, the
    // envelope only counts (EnvelopeCRT::step, audioelement.h:702-731,965-968).  Anything else takes the general step below.
    const bool act = l < P && rt_active(c);
    const bool holds = c.state == ENV_PRESSED && c.ahdstate < 0 && c.counter >= 1 && (c.reldelay < 0 || c.reldelay >= m) && (c.flags & FL_HAS_FRES) &&
                       c.fres == c.to && c.inc == c.to && (c.flags & FL_HAS_VOL) && c.vol == c.target;
    if (__syncthreads_and(!act || holds)) {
      if (act) {
        double ang = x.ph_angle, cre = c.cre, cim = c.cim;
        int nz = x.nz;
        const double inc = x.ph_inc, al = x.ph_alias, mre = c.mre, mim = c.mim, sal = c.alias, vol = c.vol, val = c.value;
        // a lane's samples form two recurrences (angle; phasor) with the mix hanging off them: unrolled, so that the mix of one sample
        // issues in the shadow of the next sample's recurrence (one wave per SIMD: nothing else hides the fp64 latency)
        auto one = [&](int i) {
          ang += inc;                                                // Phased::step :1452-1470
          if (ang > 2.0) ang -= 2.0;
          const double re = cre * mre - cim * mim, im = cre * mim + cim * mre; // SineOscillatorAlgo::step
          const double g = 0.5 * (3.0 - (re * re + im * im));
          cre = re * g; cim = im * g;
          double sum = 0.f;                                          // UnityGainMix::imag, as below
          if (v_noise != 0.) { nz = (nz + 1 >= nzn) ? 0 : nz + 1; sum += v_noise * nzb[nz]; } // soundBufferWrapperAlgo::step / imag :1563-1574
          if (v_saw != 0.) sum += v_saw * (al * wave_saw(ang));
          if (v_tri != 0.) sum += v_tri * (al * wave_triangle(ang));
          if (v_sq != 0.) sum += v_sq * (al * wave_square(ang));
          if (v_sine != 0.) sum += v_sine * (sal * cim);
          if (v_pulse != 0.) sum += v_pulse * (al * wave_pulse(ang, pw, p_high, p_low));
          mine[i] = vol * (sum * val);
        };
        if (m == CT) {
#pragma unroll
          for (int i = 0; i < CT; ++i) one(i);
        } else {
          for (int i = 0; i < m; ++i) one(i);
        }
        x.ph_angle = ang; c.cre = cre; c.cim = cim; x.nz = nz;
        c.counter += m;
        if (c.reldelay >= 0) c.reldelay -= m;
      } else {
        for (int i = 0; i < m; ++i) mine[i] = 0.0;
      }
      clk.lap<3>();
    } else
    for (int i = 0; i < m; ++i) {
      double v = 0.0;
      if (rt_active(c)) { // ImplCRTP::compute, gen.crtp.h:352-371
        // BaseVolumeAdjusted::step (:1195-1216) -> Enveloped::step (:197-200) -> FreqCtrl_::step (:3126-3129)
        // the envelope while it holds its sustain level (most of a note): EnvelopeCRT::step only counts (audioelement.h:702-731,965-968)
        if (c.state == ENV_PRESSED && c.ahdstate < 0 && c.reldelay < 0 && c.counter >= 1) ++c.counter;
        else env_step(c, ep);
        const double f = ctrl_step(c, 100.0);
        if (c.inc != f) { // both setAngleIncrements are functions of f alone: evaluated when f moves (the sine's own test, audioelement.h:2405)
          x.ph_inc = phase_to_normal_form(f); x.ph_alias = alias_mult(f); // Phased::setAngleIncrements :1443-1446
          osc_set_inc(c, f);
        }
        if (v_noise != 0.) x.nz = (x.nz + 1 >= nzn) ? 0 : x.nz + 1;   // soundBufferWrapperAlgo::step :1568-1574 (the mix's first voice)
        x.ph_angle += x.ph_inc;                                        // Phased::step :1452-1470
        if (x.ph_angle > 2.0) x.ph_angle -= 2.0;
        { // SineOscillatorAlgo::step
          const double re = c.cre * c.mre - c.cim * c.mim, im = c.cre * c.mim + c.cim * c.mre;
          const double g = 0.5 * (3.0 - (re * re + im * im));
          c.cre = re * g; c.cim = im * g;
        }
        // the outer volume filter: the first step takes the target (setInitialValue, then feed(target)); afterwards volume == target for
        // the whole note and feed(target) returns it unchanged, whatever the coefficient
        if (!(c.flags & FL_HAS_VOL)) {
          const double inc = fabs(car_angle_increments(c, x)), a = M_PI * inc;
          c.finit = inc; c.k = a / (1.0 + a);
          c.y = c.target + c.k * (c.target - c.target);
          c.vol = c.y;
          c.flags |= FL_HAS_VOL;
        } else if (c.vol != c.target) { // (unreachable while targets only change at note-on; kept for the reference's shape)
          const double inc = fabs(car_angle_increments(c, x));
          if (c.finit != inc) { c.finit = inc; const double a = M_PI * inc; c.k = a / (1.0 + a); }
          c.y = c.y + c.k * (c.target - c.y);
          c.vol = c.y;
        }
        if (rt_active(c)) {
          // UnityGainMix::imag (:1966-1972): sum = 0.f; sum += volume_i * (aliasingMult * wave_i(angle)), in tuple order.  A voice whose
          // volume is 0 adds +-0 to a sum that started at +0: skipped (workgroup-uniform tests)
          double sum = 0.f;
          if (v_noise != 0.) sum += v_noise * nzb[x.nz];
          if (v_saw != 0.) sum += v_saw * (x.ph_alias * wave_saw(x.ph_angle));
          if (v_tri != 0.) sum += v_tri * (x.ph_alias * wave_triangle(x.ph_angle));
          if (v_sq != 0.) sum += v_sq * (x.ph_alias * wave_square(x.ph_angle));
          if (v_sine != 0.) sum += v_sine * (c.alias * c.cim);
          if (v_pulse != 0.) sum += v_pulse * (x.ph_alias * wave_pulse(x.ph_angle, pw, p_high, p_low));
          v = c.vol * (sum * c.value); // BaseVolumeAdjusted::imag (:1126-1129) over Enveloped::imag (:227-231)
        }
      }
      mine[i] = v;
    }
    clk.lap<1>();
    __syncthreads();
    if (l < m) { // `buffer[i] += c.elem.imag()` over the channels in index order (silent channels add 0)
      double acc = 0.0;
#pragma unroll 8
      for (int ch = 0; ch < P; ++ch) acc += tile[(size_t)ch * (CT + 1) + l];
      out_d[t - t0 + l] = acc;
      out_f[t - t0 + l] = (float)acc;
    }
    __syncthreads();
    clk.lap<2>();
    t += m;
  }
  clk.flush(8);
  store_chan(c, sd, si, Ppad, l);
  ext[CX_PH_ANGLE * Ppad + l] = x.ph_angle; ext[CX_PH_INC * Ppad + l] = x.ph_inc; ext[CX_PH_ALIAS * Ppad + l] = x.ph_alias;
  ext[CX_NZ_INDEX * Ppad + l] = (double)x.nz;
  b.car_noteid[(size_t)s * Ppad + l] = noteid;
  if (l == 0 && dropped) b.car_dropped[s] += dropped;
}


// ---- K6, time-parallel form (the default) ------------------------------------------------------------------------------
// The serial kernel above runs 480 k dependent samples per stream on one wave: ~130 ms per 10 s block of 256 streams, whatever the chip.
// Of everything a channel carries only ONE recurrence has no closed form that keeps its bits: the shared angle of the phase-controlled
// oscillators.  The envelope's value is a function of its integer counters (env_jump), the control and the volume are constant after a
// note's first step, the sine's phasor is a phase (theta += inc).  So:
//   walk   (one workgroup per stream, lane = channel, samples in order): runs the events' one-shots, cuts the block into segments at
//          the events and every CAR_G samples, stores every channel's state at the start of each segment, and moves on to the next
//          segment by env_jump + `angle += inc; if (angle > 2) angle -= 2` (the exact sequence, four instructions per sample) + a phase
//          addition.  Steps that are not regular (a note's first step, envelope stage changes) are taken one at a time.
//   render (one wave per 64 / CH segments, lane = (segment, channel)): steps every segment from its stored state with the same code as the
//          serial kernel, mixes the voices and sums the channels in channel order.
// The angle, the envelope and the channel sum are the same operation sequences as in the serial kernel: the phase-controlled voices are
// bit-identical to it (and to the CPU restatement); the sine voice restarts every segment from sincospi(theta) instead of carrying the
// renormalised phasor product, a 1e-13 difference.
constexpr int CAR_G = 256; // samples per segment (segments are also cut at events)
enum { CS_VALUE, CS_TOP, CS_FROM, CS_TO, CS_CUR, CS_C, CS_FRES, CS_INC, CS_ALIAS, CS_VOL, CS_TARGET, CS_FINIT, CS_K, CS_Y, CS_THETA, CS_PH_ANGLE, CS_PH_INC,
       CS_PH_ALIAS, CS_NZ_INDEX, CS_COUNT };

__device__ inline void car_store_seg(const Chan &c, const CarX &x, double *d, int *i, int CH, int ch) {
  d[CS_VALUE * CH + ch] = c.value; d[CS_TOP * CH + ch] = c.top; d[CS_FROM * CH + ch] = c.from; d[CS_TO * CH + ch] = c.to; d[CS_CUR * CH + ch] = c.cur;
  d[CS_C * CH + ch] = c.C; d[CS_FRES * CH + ch] = c.fres; d[CS_INC * CH + ch] = c.inc; d[CS_ALIAS * CH + ch] = c.alias; d[CS_VOL * CH + ch] = c.vol;
  d[CS_TARGET * CH + ch] = c.target; d[CS_FINIT * CH + ch] = c.finit; d[CS_K * CH + ch] = c.k; d[CS_Y * CH + ch] = c.y; d[CS_THETA * CH + ch] = c.theta;
  d[CS_PH_ANGLE * CH + ch] = x.ph_angle; d[CS_PH_INC * CH + ch] = x.ph_inc; d[CS_PH_ALIAS * CH + ch] = x.ph_alias; d[CS_NZ_INDEX * CH + ch] = (double)x.nz;
  i[SI_ENV_STATE * CH + ch] = c.state; i[SI_ENV_COUNTER * CH + ch] = c.counter; i[SI_ENV_RELDELAY * CH + ch] = c.reldelay;
  i[SI_ENV_AHDSTATE * CH + ch] = c.ahdstate; i[SI_ENV_AHDCOUNTER * CH + ch] = c.ahdcounter; i[SI_ENV_MINCH * CH + ch] = c.minch;
  i[SI_ENV_NEXTMINCH * CH + ch] = c.nextminch; i[SI_FLAGS * CH + ch] = c.flags;
}
__device__ inline void car_load_seg(Chan &c, CarX &x, const double *d, const int *i, int CH, int ch) {
  c.value = d[CS_VALUE * CH + ch]; c.top = d[CS_TOP * CH + ch]; c.from = d[CS_FROM * CH + ch]; c.to = d[CS_TO * CH + ch]; c.cur = d[CS_CUR * CH + ch];
  c.C = d[CS_C * CH + ch]; c.fres = d[CS_FRES * CH + ch]; c.inc = d[CS_INC * CH + ch]; c.alias = d[CS_ALIAS * CH + ch]; c.vol = d[CS_VOL * CH + ch];
  c.target = d[CS_TARGET * CH + ch]; c.finit = d[CS_FINIT * CH + ch]; c.k = d[CS_K * CH + ch]; c.y = d[CS_Y * CH + ch]; c.theta = d[CS_THETA * CH + ch];
  x.ph_angle = d[CS_PH_ANGLE * CH + ch]; x.ph_inc = d[CS_PH_INC * CH + ch]; x.ph_alias = d[CS_PH_ALIAS * CH + ch]; x.nz = (int)d[CS_NZ_INDEX * CH + ch];
  c.state = i[SI_ENV_STATE * CH + ch]; c.counter = i[SI_ENV_COUNTER * CH + ch]; c.reldelay = i[SI_ENV_RELDELAY * CH + ch];
  c.ahdstate = i[SI_ENV_AHDSTATE * CH + ch]; c.ahdcounter = i[SI_ENV_AHDCOUNTER * CH + ch]; c.minch = i[SI_ENV_MINCH * CH + ch];
  c.nextminch = i[SI_ENV_NEXTMINCH * CH + ch]; c.flags = i[SI_FLAGS * CH + ch];
}

// everything a step does to a channel except the two oscillator recurrences and the mix (the walk counts the steps and applies the
// recurrences afterwards: the increments a step uses are those it has just set)
__device__ inline void car_step_state(Chan &c, CarX &x, const RaEnvParams &ep) {
  if (c.state == ENV_PRESSED && c.ahdstate < 0 && c.reldelay < 0 && c.counter >= 1) ++c.counter;
  else env_step(c, ep);
  const double f = ctrl_step(c, 100.0);
  if (c.inc != f) {
    x.ph_inc = phase_to_normal_form(f); x.ph_alias = alias_mult(f);
    c.inc = f; c.alias = alias_mult(f); // osc_set_inc without the rotation (the render derives it from the increment)
  }
  if (!(c.flags & FL_HAS_VOL)) {
    const double inc = fabs(car_angle_increments(c, x)), a = M_PI * inc;
    c.finit = inc; c.k = a / (1.0 + a);
    c.y = c.target + c.k * (c.target - c.target);
    c.vol = c.y;
    c.flags |= FL_HAS_VOL;
  } else if (c.vol != c.target) {
    const double inc = fabs(car_angle_increments(c, x));
    if (c.finit != inc) { c.finit = inc; const double a = M_PI * inc; c.k = a / (1.0 + a); }
    c.y = c.y + c.k * (c.target - c.y);
    c.vol = c.y;
  }
}

__global__ void __launch_bounds__(RA_MAX_CHANNELS) ra_carrier_walk_kernel(RaParams p, RaBuffers b, int64_t t0, int64_t n) {
  const int s = blockIdx.x, l = threadIdx.x, Ppad = p.car_Ppad, P = p.car_P, CH = p.car_CH;
  double *xch = reinterpret_cast<double *>(ra_smem_car); // [4] broadcast slots
  int *xi = reinterpret_cast<int *>(xch + 4);             // [4]
  Chan c;
  CarX x;
  double *sd = b.car_sd + (size_t)s * SD_COUNT * Ppad;
  int *si = b.car_si + (size_t)s * SI_COUNT * Ppad;
  double *ext = b.car_ext + (size_t)s * CX_COUNT * Ppad;
  load_chan(c, sd, si, Ppad, l);
  x.ph_angle = ext[CX_PH_ANGLE * Ppad + l]; x.ph_inc = ext[CX_PH_INC * Ppad + l]; x.ph_alias = ext[CX_PH_ALIAS * Ppad + l];
  x.nz = (int)ext[CX_NZ_INDEX * Ppad + l];
  const bool has_noise = b.car_noise != nullptr; // (noise volume != 0)
  const int nzn = b.car_noise_len;
  int64_t noteid = b.car_noteid[(size_t)s * Ppad + l];
  int dropped = 0;
  const RaEnvParams ep = {p.envA, p.envH, p.envD, p.envR, p.SMinusOne};
  const uint64_t seed = p.seed + (uint64_t)s;
  const RaCarEv *ev = b.car_ev;
  int e = b.car_ev_begin[s];
  const int e_end = b.car_ev_begin[s + 1];
  // this stream's slots of the segment table: one per CAR_G samples, one more for the block's ragged end, one per event
  const int64_t per_stream = (n + CAR_G - 1) / CAR_G + 1;
  int64_t seg = (int64_t)s * per_stream + b.car_ev_begin[s];
  const int64_t seg_stop = (int64_t)(s + 1) * per_stream + e_end;
  RaPhaseClock clk(b.dbg_clk, l == 0);
  int64_t t = t0;
  const int64_t t_end = t0 + n;
  constexpr int64_t kNever = 0x7fffffffffffffffll;
  int64_t next_ev = (e < e_end) ? ev[e].sample : kNever;
  while (t < t_end) {
    if (next_ev <= t) { // the one-shots queued for this frame run before its step (simple_audio_context.h:74-76)
      const RaCarEv q = ev[e];
      ++e;
      next_ev = (e < e_end) ? ev[e].sample : kNever;
      if (l == 0) { xi[0] = 0x7fffffff; xi[1] = 0x7fffffff; }
      __syncthreads();
      if (q.type == 0) { // ---- onEvent(NoteOn), gen.crtp.h:389-449, and its one-shot :455-527 (as in the serial kernel) ----
        if (l < P && c.state == ENV_DONE2) atomicMin(&xi[0], l);
        __syncthreads();
        const int sel = xi[0];
        if (sel == 0x7fffffff) { if (l == 0) ++dropped; }
        else {
          if (l == sel) {
            c.state = ENV_SOON;
            noteid = q.noteid;
            chan_forget_past_signals(c);
            c.target = 1.0 * (double)q.vel; c.flags |= FL_HAS_TARGET;
            c.cur = 0;
            const float inc_f = 2 * (q.freq * 1.f) / p.sample_rate;
            chan_set_angle_increments(c, (double)inc_f, 0);
            xch[0] = car_angle_increments(c, x);
          }
          __syncthreads();
          if (l != sel && l < P && rt_active(c) && car_angle_increments(c, x) == xch[0]) atomicMin(&xi[1], l);
          __syncthreads();
          const int src = xi[1];
          if (src != 0x7fffffff && l == src) { xch[1] = x.ph_angle; xch[2] = c.theta; xi[2] = x.nz; }
          __syncthreads();
          if (l == sel) {
            if (src != 0x7fffffff) {
              x.ph_angle = phase_to_normal_form(xch[1]);
              c.theta = xch[2];                          // SineOscillatorAlgo::synchronizeAngles :2397-2399, in the phase domain
              x.nz = xi[2];                              // soundBufferWrapperAlgo::synchronizeAngles :1532-1537
            } else {
              const float a = (p.phase_policy == RA_PHASE_HASH) ? hash_uniform_pm1(seed, (uint64_t)q.noteid, 2) : 0.f;
              x.ph_angle = phase_to_normal_form((double)a);
              c.theta = (double)a;                       // setStartAngle :2401-2403
              if (has_noise) x.nz = noise_start_index((double)a, nzn);
            }
            env_on_key_pressed(c, q.delay);
          }
        }
      } else { // ---- NoteOff one-shot, gen.crtp.h:541-588 ----
        if (l < P && noteid == q.noteid) atomicMin(&xi[0], l);
        __syncthreads();
        if (l == xi[0] && env_can_release_now(c, q.delay)) env_on_key_released(c, q.delay);
      }
      __syncthreads();
      continue;
    }
    const int64_t grid_end = t0 + ((t - t0) / CAR_G + 1) * CAR_G;
    const int64_t seg_end = min(min(t_end, next_ev), grid_end);
    const int len = (int)(seg_end - t);
    const bool act = l < P && rt_active(c);
    const int any = __syncthreads_or(act);
    if (seg < seg_stop) { // (always: the table has a slot for every cut)
      if (l == 0) b.car_seg_info[seg] = make_int4(s, (int)(t - t0), len, any);
      if (any && l < CH) car_store_seg(c, x, b.car_seg_d + (size_t)seg * CS_COUNT * CH, b.car_seg_i + (size_t)seg * SI_COUNT * CH, CH, l);
    }
    ++seg;
    if (any) {
      // the channel's steps of this segment: `na` of them, all with the increments the first one leaves behind
      int na = 0;
      {
        int left = len;
        while (left > 0 && rt_active(c) && l < P) {
          int j = env_regular_steps(c, ep);
          const bool regular = j > 0 && (c.flags & FL_HAS_VOL) && (c.flags & FL_HAS_FRES) && c.fres == c.to && c.inc == c.to && c.vol == c.target;
          if (!regular) {
            const double inc_before = x.ph_inc;
            car_step_state(c, x, ep);
            if (na > 0 && x.ph_inc != inc_before) atomicOr(b.error_flags, RA_ERR_BOOKKEEPING); // increments only move on a note's first step
            ++na; --left;
            continue;
          }
          j = min(j, left);
          env_jump(c, ep, j);
          na += j; left -= j;
        }
      }
      // Phased::step (audioelement.h:1452-1470), `na` times: the one recurrence that has to be run
      {
        double ang = x.ph_angle;
        const double inc = x.ph_inc;
        auto one = [&](int i) {
          const double t1 = ang + (i < na ? inc : 0.0); // (an angle that is not stepped stays <= 2)
          const double t2 = t1 - 2.0;
          ang = t1 > 2.0 ? t2 : t1;
        };
        int i = 0;
        for (; i + 8 <= len; i += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) one(i + u);
        }
        for (; i < len; ++i) one(i);
        x.ph_angle = ang;
      }
      // SineOscillatorAlgo::step, `na` times, as a phase
      if (na > 0) {
        double th = fma((double)na, c.inc, c.theta);
        th -= 2.0 * floor(0.5 * th);
        c.theta = th;
      }
      if (has_noise) x.nz = noise_advance(x.nz, na, nzn); // soundBufferWrapperAlgo::step, `na` times
    }
    t = seg_end;
  }
  // unused slots of this stream's part of the table
  for (int64_t q = seg + l; q < seg_stop; q += blockDim.x) b.car_seg_info[q] = make_int4(s, 0, 0, 0);
  store_chan(c, sd, si, Ppad, l);
  ext[CX_PH_ANGLE * Ppad + l] = x.ph_angle; ext[CX_PH_INC * Ppad + l] = x.ph_inc; ext[CX_PH_ALIAS * Ppad + l] = x.ph_alias;
  ext[CX_NZ_INDEX * Ppad + l] = (double)x.nz;
  b.car_noteid[(size_t)s * Ppad + l] = noteid;
  if (l == 0 && dropped) b.car_dropped[s] += dropped;
}

__global__ void __launch_bounds__(RA_MAX_CHANNELS) ra_carrier_render_kernel(RaParams p, RaBuffers b, int64_t n, int64_t total_segs) {
  const int l = threadIdx.x, nth = blockDim.x, P = p.car_P, CH = p.car_CH;
  const int SUB = nth / CH;                       // segments per workgroup (CH >= 64: one)
  const int g = l / CH, ch = l - g * CH;
  double *tile = reinterpret_cast<double *>(ra_smem_car);          // [nth][CT + 1]
  int4 *infos = reinterpret_cast<int4 *>(tile + (size_t)nth * (CT + 1)); // [SUB]
  int *smax = reinterpret_cast<int *>(infos + SUB);
  const int64_t seg = (int64_t)blockIdx.x * SUB + g;
  int4 info = make_int4(0, 0, 0, 0);
  if (seg < total_segs) info = b.car_seg_info[seg];
  if (l == 0) *smax = 0;
  __syncthreads();
  double *out_d = b.car_gen_d + (size_t)info.x * n + info.y;
  float *out_f = b.car_gen_f + (size_t)info.x * n + info.y;
  if (info.z > 0 && !info.w) { // nothing sounds in this segment
    for (int r = ch; r < info.z; r += CH) { out_d[r] = 0.0; out_f[r] = 0.f; }
    info.z = 0;
  }
  if (ch == 0) { infos[g] = info; if (info.z > 0) atomicMax(smax, info.z); }
  __syncthreads();
  const int maxlen = *smax;
  if (maxlen == 0) return;
  Chan c;
  CarX x;
  chan_init(c);
  x.ph_angle = 0; x.ph_inc = 0; x.ph_alias = 1; x.nz = -1;
  const int len = info.z;
  if (len > 0 && ch < P) {
    car_load_seg(c, x, b.car_seg_d + (size_t)seg * CS_COUNT * CH, b.car_seg_i + (size_t)seg * SI_COUNT * CH, CH, ch);
    double sn, cs;
    sincospi(c.theta, &sn, &cs);
    c.cre = cs; c.cim = sn;
    sincos(M_PI * c.inc, &sn, &cs); // osc_set_inc's rotation
    c.mre = cs; c.mim = sn;
  }
  const RaEnvParams ep = {p.envA, p.envH, p.envD, p.envR, p.SMinusOne};
  const double *nzb = b.car_noise; // null: noise volume 0
  const int nzn = b.car_noise_len;
  const double v_noise = nzb ? (double)p.car_vol[0] : 0.0;
  const double v_saw = (double)p.car_vol[1], v_tri = (double)p.car_vol[2], v_sq = (double)p.car_vol[3],
               v_sine = (double)p.car_vol[4], v_pulse = (double)p.car_vol[5];
  double pw = (double)p.car_pulse_width;
  pw = fmax(0., pw); pw = fmin(2., pw);
  const double p_high = 0.5 * (2. - pw), p_low = p_high - 1.;
  double *mine = tile + (size_t)l * (CT + 1);
  for (int tp = 0; tp < maxlen; tp += CT) {
    const int m = max(0, min(CT, len - tp));
    const bool act = m > 0 && ch < P && rt_active(c);
    const bool holds = c.state == ENV_PRESSED && c.ahdstate < 0 && c.counter >= 1 && (c.reldelay < 0 || c.reldelay >= m) && (c.flags & FL_HAS_FRES) &&
                       c.fres == c.to && c.inc == c.to && (c.flags & FL_HAS_VOL) && c.vol == c.target;
    if (__all(!act || holds)) { // every sounding channel of the wave just holds its note: oscillators and mix only
      double ang = x.ph_angle, cre = c.cre, cim = c.cim;
      int nz = x.nz;
      const double inc = x.ph_inc, al = x.ph_alias, mre = c.mre, mim = c.mim, sal = c.alias, vol = c.vol, val = c.value;
      auto one = [&](int i) {
        ang += inc;                                                // Phased::step :1452-1470
        if (ang > 2.0) ang -= 2.0;
        const double re = cre * mre - cim * mim, im = cre * mim + cim * mre; // SineOscillatorAlgo::step
        const double gn = 0.5 * (3.0 - (re * re + im * im));
        cre = re * gn; cim = im * gn;
        double sum = 0.f;                                          // UnityGainMix::imag
        if (v_noise != 0.) { nz = (nz + 1 >= nzn) ? 0 : nz + 1; sum += v_noise * nzb[nz]; } // soundBufferWrapperAlgo::step / imag :1563-1574
        if (v_saw != 0.) sum += v_saw * (al * wave_saw(ang));
        if (v_tri != 0.) sum += v_tri * (al * wave_triangle(ang));
        if (v_sq != 0.) sum += v_sq * (al * wave_square(ang));
        if (v_sine != 0.) sum += v_sine * (sal * cim);
        if (v_pulse != 0.) sum += v_pulse * (al * wave_pulse(ang, pw, p_high, p_low));
        mine[i] = vol * (sum * val);
      };
      if (act) {
        if (m == CT) {
#pragma unroll
          for (int i = 0; i < CT; ++i) one(i);
        } else {
          for (int i = 0; i < m; ++i) one(i);
        }
        x.ph_angle = ang; c.cre = cre; c.cim = cim; x.nz = nz;
        c.counter += m;
        if (c.reldelay >= 0) c.reldelay -= m;
      } else {
        for (int i = 0; i < m; ++i) mine[i] = 0.0;
      }
    } else {
      for (int i = 0; i < m; ++i) {
        double v = 0.0;
        if (ch < P && rt_active(c)) { // ImplCRTP::compute, gen.crtp.h:352-371 (the serial kernel's general step)
          if (c.state == ENV_PRESSED && c.ahdstate < 0 && c.reldelay < 0 && c.counter >= 1) ++c.counter;
          else env_step(c, ep);
          const double f = ctrl_step(c, 100.0);
          if (c.inc != f) {
            x.ph_inc = phase_to_normal_form(f); x.ph_alias = alias_mult(f);
            osc_set_inc(c, f);
          }
          if (v_noise != 0.) x.nz = (x.nz + 1 >= nzn) ? 0 : x.nz + 1;
          x.ph_angle += x.ph_inc;
          if (x.ph_angle > 2.0) x.ph_angle -= 2.0;
          {
            const double re = c.cre * c.mre - c.cim * c.mim, im = c.cre * c.mim + c.cim * c.mre;
            const double gn = 0.5 * (3.0 - (re * re + im * im));
            c.cre = re * gn; c.cim = im * gn;
          }
          if (!(c.flags & FL_HAS_VOL)) {
            const double inc = fabs(car_angle_increments(c, x)), a = M_PI * inc;
            c.finit = inc; c.k = a / (1.0 + a);
            c.y = c.target + c.k * (c.target - c.target);
            c.vol = c.y;
            c.flags |= FL_HAS_VOL;
          } else if (c.vol != c.target) {
            const double inc = fabs(car_angle_increments(c, x));
            if (c.finit != inc) { c.finit = inc; const double a = M_PI * inc; c.k = a / (1.0 + a); }
            c.y = c.y + c.k * (c.target - c.y);
            c.vol = c.y;
          }
          if (rt_active(c)) {
            double sum = 0.f;
            if (v_noise != 0.) sum += v_noise * nzb[x.nz];
            if (v_saw != 0.) sum += v_saw * (x.ph_alias * wave_saw(x.ph_angle));
            if (v_tri != 0.) sum += v_tri * (x.ph_alias * wave_triangle(x.ph_angle));
            if (v_sq != 0.) sum += v_sq * (x.ph_alias * wave_square(x.ph_angle));
            if (v_sine != 0.) sum += v_sine * (c.alias * c.cim);
            if (v_pulse != 0.) sum += v_pulse * (x.ph_alias * wave_pulse(x.ph_angle, pw, p_high, p_low));
            v = c.vol * (sum * c.value);
          }
        }
        mine[i] = v;
      }
    }
    __syncthreads();
    // `buffer[i] += c.elem.imag()` over the channels in index order (silent channels add 0), for every segment of the workgroup
    for (int o = l; o < SUB * CT; o += nth) {
      const int og = o / CT, i = o - og * CT;
      const int4 gi = infos[og];
      if (i < gi.z - tp) {
        double acc = 0.0;
        const double *col = tile + (size_t)og * CH * (CT + 1) + i;
        for (int k = 0; k < P; ++k) acc += col[(size_t)k * (CT + 1)];
        const size_t at = (size_t)gi.x * n + gi.y + tp + i;
        b.car_gen_d[at] = acc;
        b.car_gen_f[at] = (float)acc;
      }
    }
    __syncthreads();
  }
}

// ---- K7 -------------------------------------------------------------------------------------------------------------------
constexpr int BT = 64; // samples per tile = lanes
__global__ void __launch_bounds__(64) ra_voc_bank_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, const float *__restrict__ carrier_f,
                                                        const double *__restrict__ carrier_d, int64_t n) {
  const int s = blockIdx.x, l = threadIdx.x, B = p.voc_bands;
  double *tile = reinterpret_cast<double *>(ra_smem_car); // [BT][65]: row = sample, column = band
  const bool band = l < B;
  const double kA = b.bank_k[l], kB = b.bank_k[64 + l], kE = b.bank_k[128 + l];
  double *st = b.bank_state + (size_t)s * 5 * 64;
  double yA = st[l], yB = st[64 + l], yE = st[128 + l], yC = st[192 + l], yD = st[256 + l];
  const float *vs = voice ? voice + (size_t)s * n : nullptr;
  const float *cf = carrier_f ? carrier_f + (size_t)s * n : nullptr;
  const double *cd = carrier_d ? carrier_d + (size_t)s * n : nullptr;
  double *mix = b.voc_mix + (size_t)s * n;
  // this lane's sample of a tile (the modulator as the double read_direct_input widens it to, lib.cpp:1073-1079); the next tile's
  // samples are requested before the current tile is computed, so that no load has to wait behind the tile's output stores
  auto fetch = [&](int64_t t, double &mv, double &cv) {
    mv = 0.0; cv = 0.0;
    if (t + l < n) {
      if (vs) mv = (double)vs[t + l];
      if (cd) cv = cd[t + l]; else if (cf) cv = (double)cf[t + l];
    }
  };
  double mv_next, cv_next;
  fetch(0, mv_next, cv_next);
  for (int64_t t = 0; t < n; t += BT) {
    const int cnt = (int)min((int64_t)BT, n - t);
    const double mv = mv_next, cv = cv_next;
    fetch(t + BT, mv_next, cv_next);
    auto sample = [&](int i) {
      const double m = readlane_f64(mv, i), c = readlane_f64(cv, i); // wave-uniform: scalar operands of the recurrences
      // Modulator::feed (vocoder.cpp:358-380): BandPass::feed (:54-58) = high-pass then low-pass; EnvelopeFollower::feed (:71-74) of |.|
      yA = yA + kA * (m - yA);
      const double hp = m - yA;             // high-pass = input - low-pass
      yB = yB + kB * (hp - yB);
      yE = yE + kE * (fabs(yB) - yE);
      // Carrier::feed (:698-718)
      yC = yC + kA * (c - yC);
      const double hpc = c - yC;
      yD = yD + kB * (hpc - yD);
      tile[i * 65 + l] = band ? yE * yD : 0.0; // ponderation[i] * filtered
    };
    if (cnt == BT) {
#pragma unroll 8
      for (int i = 0; i < BT; ++i) sample(i);
    } else {
      for (int i = 0; i < cnt; ++i) sample(i);
    }
    // `res += ponderation[i] * filtered` in band order (one wave: its LDS operations execute in order)
    if (l < cnt) {
      double acc = 0.0;
      const double *row = tile + l * 65;
      for (int q = 0; q < B; ++q) acc += row[q];
      mix[t + l] = (double)p.vol_mod * mv + (double)p.vol_car * cv + (double)p.vol_voc * acc; // vocoder.cpp:799-803
    }
  }
  st[l] = yA; st[64 + l] = yB; st[128 + l] = yE; st[192 + l] = yC; st[256 + l] = yD;
}


// K7 as a pipeline of wavefronts.  The kernel above runs the five recurrences of a stream on one wave: ~23 dependent-ish fp64 instructions
// per sample at the rate a lone wave issues them (one per ~10 cycles): 38 ms per 10 s block, whatever the chip.  The recurrences form a
// chain of stages (modulator low-pass of the high-pass -> band low-pass -> follower; carrier the same without the follower), each a
// one-pole whose input is the previous stage's output of the SAME sample: six waves of one workgroup take one stage each and hand
// 16-sample tiles to each other through LDS, one barrier per tile, the consumer one tile behind its producer (double-buffered edges).
// Every stage evaluates the same expressions in the same order as the single-wave kernel: the results are bit-identical.
constexpr int PT = 16;                  // samples per pipeline tile
constexpr int PROW = 65;                // doubles per sample row ([sample][band], odd: the summing lanes walk rows)
constexpr int PEDGE = PT * PROW;        // one buffer of an edge
__global__ void __launch_bounds__(384) ra_voc_bank_pipe_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, const float *__restrict__ carrier_f,
                                                             const double *__restrict__ carrier_d, int64_t n, int perm) {
  const int s = blockIdx.x, l = threadIdx.x & 63, B = p.voc_bands;
  // the stage this wave runs.  A workgroup's waves go to the CU's four SIMDs round robin: waves 0 / 4 and 1 / 5 share one.  The two input
  // stages issue the most instructions per sample and get a SIMD of their own (RA_BANK_PERM: tuning aid, six digits = stage of wave 0..5)
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = (perm >> (4 * wid)) & 15;
  double *lds = reinterpret_cast<double *>(ra_smem_car);
  // edges: 0 hp of the modulator (stage 0 -> 1), 1 |band-passed modulator| (1 -> 2), 2 envelope (2 -> 4), 3 hp of the carrier (3 -> 4), 4 products (4 -> 5)
  auto edge = [&](int e, int64_t tile) -> double * { return lds + ((size_t)e * 2 + (size_t)(tile & 1)) * PEDGE; };
  const bool band = l < B;
  const double kA = b.bank_k[l], kB = b.bank_k[64 + l], kE = b.bank_k[128 + l];
  double *st = b.bank_state + (size_t)s * 5 * 64;
  const int delay = (w == 0) ? 0 : (w == 1) ? 1 : (w == 2 || w == 3) ? 2 : (w == 4) ? 3 : 4;
  double y = 0.0; // this stage's filter state
  if (w < 5) y = st[w * 64 + l];
  const float *vs = voice ? voice + (size_t)s * n : nullptr;
  const float *cf = carrier_f ? carrier_f + (size_t)s * n : nullptr;
  const double *cd = carrier_d ? carrier_d + (size_t)s * n : nullptr;
  double *mix = b.voc_mix + (size_t)s * n;
  const int64_t ntiles = (n + PT - 1) / PT;
  auto fetch_m = [&](int64_t tile) -> double { const int64_t at = tile * PT + l; return (vs && tile >= 0 && tile < ntiles && l < PT && at < n) ? (double)vs[at] : 0.0; };
  auto fetch_c = [&](int64_t tile) -> double {
    const int64_t at = tile * PT + l;
    if (!(tile >= 0 && tile < ntiles && l < PT && at < n)) return 0.0;
    return cd ? cd[at] : cf ? (double)cf[at] : 0.0;
  };
  // the input tile of the stages that read global memory, requested one step ahead
  double in_next = (w == 0 || w == 5) ? fetch_m(0) : (w == 3) ? fetch_c(0) : 0.0;
  double in2_next = (w == 5) ? fetch_c(0) : 0.0; // the summing stage mixes the dry inputs in: both requested a step ahead, like the stages' inputs
  unsigned long long clk_work = 0, clk_wait = 0, clk_t = b.dbg_clk ? __builtin_readcyclecounter() : 0; // tuning aid (RA_PHASE_CLOCKS): per stage, work / barrier
  for (int64_t step = 0; step < ntiles + 4; ++step) {
    const int64_t tile = step - delay;
    if (tile >= 0 && tile < ntiles) {
      const int cnt = (int)min((int64_t)PT, n - tile * PT);
      const bool full = cnt == PT; // (all but the block's last tile: straight-line code, the compiler interleaves the samples)
      if (w == 0 || w == 3) { // low-pass at the band's lower edge; out: input - low-pass (the high-pass)
        const double in = in_next;
        in_next = (w == 0) ? fetch_m(tile + 1) : fetch_c(tile + 1);
        double *out = edge(w == 0 ? 0 : 3, tile) + l;
        if (full) {
#pragma unroll
          for (int i = 0; i < PT; ++i) {
            const double x = readlane_f64(in, i);
            y = y + kA * (x - y);
            out[i * PROW] = x - y;
          }
        } else {
          for (int i = 0; i < cnt; ++i) {
            const double x = readlane_f64(in, i);
            y = y + kA * (x - y);
            out[i * PROW] = x - y;
          }
        }
      } else if (w == 1 || w == 2) { // 1: band low-pass of the modulator, out: its magnitude; 2: envelope follower
        const double *in = edge(w - 1, tile) + l;
        double *out = edge(w, tile) + l;
        const double k = (w == 1) ? kB : kE;
        if (full) {
          double v[PT];
#pragma unroll
          for (int i = 0; i < PT; ++i) v[i] = in[i * PROW];
#pragma unroll
          for (int i = 0; i < PT; ++i) {
            y = y + k * (v[i] - y);
            out[i * PROW] = (w == 1) ? fabs(y) : y;
          }
        } else {
          for (int i = 0; i < cnt; ++i) {
            y = y + k * (in[i * PROW] - y);
            out[i * PROW] = (w == 1) ? fabs(y) : y;
          }
        }
      } else if (w == 4) { // band low-pass of the carrier, times the envelope: ponderation[i] * filtered
        const double *in = edge(3, tile) + l, *env = edge(2, tile) + l;
        double *out = edge(4, tile) + l;
        if (full) {
          double v[PT], g[PT];
#pragma unroll
          for (int i = 0; i < PT; ++i) { v[i] = in[i * PROW]; g[i] = env[i * PROW]; }
#pragma unroll
          for (int i = 0; i < PT; ++i) {
            y = y + kB * (v[i] - y);
            out[i * PROW] = band ? g[i] * y : 0.0;
          }
        } else {
          for (int i = 0; i < cnt; ++i) {
            y = y + kB * (in[i * PROW] - y);
            out[i * PROW] = band ? env[i * PROW] * y : 0.0;
          }
        }
      } else { // w == 5: `res += ponderation[i] * filtered` in band order, and the mix (vocoder.cpp:799-803)
        const double mv = in_next, cv = in2_next;
        in_next = fetch_m(tile + 1); in2_next = fetch_c(tile + 1);
        if (l < cnt) {
          const int64_t at = tile * PT + l;
          const double *row = edge(4, tile) + l * PROW;
          double acc = 0.0;
          if ((B & 15) == 0) {
            for (int q0 = 0; q0 < B; q0 += 16) { // sixteen reads in flight, then their adds in band order
              double r[16];
#pragma unroll
              for (int j = 0; j < 16; ++j) r[j] = row[q0 + j];
#pragma unroll
              for (int j = 0; j < 16; ++j) acc += r[j];
            }
          } else {
            for (int q = 0; q < B; ++q) acc += row[q];
          }
          mix[at] = (double)p.vol_mod * mv + (double)p.vol_car * cv + (double)p.vol_voc * acc;
        }
      }
    }
    if (b.dbg_clk) { const unsigned long long now = __builtin_readcyclecounter(); clk_work += now - clk_t; clk_t = now; }
    __syncthreads();
    if (b.dbg_clk) { const unsigned long long now = __builtin_readcyclecounter(); clk_wait += now - clk_t; clk_t = now; }
  }
  if (b.dbg_clk && l == 0) { atomicAdd(b.dbg_clk + 96 + 2 * w, clk_work); atomicAdd(b.dbg_clk + 97 + 2 * w, clk_wait); }
  if (w < 5) st[w * 64 + l] = y;
}


// K7, time-parallel (the default since r04).  Every stage of the bank is a constant-coefficient one-pole, y[n] = y[n-1] + k (x[n] - y[n-1]):
// an affine map of its state per sample, so a run of samples composes into one affine map (a^len, zero-state response) and the state at
// every chunk boundary of a tile follows from a prefix scan over the chunks.  A tile is 1024 samples of one stream; lane = a chunk of 16
// consecutive samples; wave w of the workgroup's eight takes bands w * bpw ... (bpw = ceil(bands / 8)) one after the other.  Per band and
// stage: (1) the chunk's zero-state response by Horner's rule, (2) a weighted inclusive scan over the 64 lanes with the constant ratio
// P = a^16 — rows of 16 lanes by DPP row shifts, the four row ends chained through scalar registers, P^((lane mod 16) + 1) from a small LDS
// table; the tile's carry-in folded into lane 0 — gives every chunk its true start state, (3) the recurrence itself from that state, its
// multiply-add fused, produces the samples.  The `abs` between the band-pass and the follower is a source modifier of the follower's
// instructions, the high-pass `x - lp(x)` and the product with the envelope are elementwise between two scans.  The band sum runs over a
// wave's bands in band order (in the wave's own LDS slice), then over the eight waves in wave order.  A block's last, ragged tile takes the
// same path with per-lane lengths (a general affine scan over (a^len, response) pairs, by ds_bpermute: once per block).  Results equal the
// sample-serial recurrence to rounding (measured against the CPU restatement: <= 1e-8 RMS on the float output, tests/test_gpu_vocoder_bank.py).
// 480 k samples of a stream are 469 tiles instead of 480 k dependent steps: the pipeline kernel above (bit-identical to the reference's loop;
// reference_order = 1) needed 30.8 ms per configs[2] step at 27 % of the issue rate; this one 7.6 ms at 55 %, in 128 VGPRs (two sample arrays;
// inputs and running sum in LDS) so that the carrier synth's walk (249 VGPRs) still finds room beside it.
constexpr int SC = 16;              // samples per lane
constexpr int STILE = 64 * SC;      // samples per tile
constexpr int SROW = 65;            // LDS row stride in doubles: sample (lane, j) of a tile lives at j * SROW + lane

__device__ inline double uniform_f64(double v) { return readlane_f64(v, 0); } // a wave-uniform value into scalar registers
// Lane moves on the VALU (DPP): a ds_bpermute round trip costs ~100 cycles, and six of them in a row per stage were what a tile waited for.
template <int CTRL, bool ZERO>
__device__ inline double dpp_f64(double old, double v) { // ZERO: lanes without a source read 0; otherwise they keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, 0xf, ZERO);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, 0xf, ZERO);
  return __hiloint2double(hi, lo);
}
// Inclusive scan over the wave's lanes of y_l = P y_(l-1) + z_l with the constant ratio P: pw[s] = P^(2^s) (s = 0 .. 5), q = P^((lane mod 16) + 1).
// Rows of 16 lanes scan themselves with row_shr moves (lanes short of a source add 0); the four row ends are chained through scalar registers
// (E_r = R_r + P^16 E_(r-1)) and every later row adds q times its predecessor's end.  y_in (the state before lane 0) enters through lane 0's z.
// Returns the scan; carry = the value at lane 63.
__device__ inline double bank_scan_const(double z, const double *__restrict__ pw, double q, int lane, double &carry) {
  z = fma(pw[0], dpp_f64<0x111, true>(0.0, z), z);
  z = fma(pw[1], dpp_f64<0x112, true>(0.0, z), z);
  z = fma(pw[2], dpp_f64<0x114, true>(0.0, z), z);
  z = fma(pw[3], dpp_f64<0x118, true>(0.0, z), z);
  const double r0 = readlane_f64(z, 15), r1 = readlane_f64(z, 31), r2 = readlane_f64(z, 47), r3 = readlane_f64(z, 63);
  const double e1 = fma(pw[4], r0, r1), e2 = fma(pw[4], e1, r2);
  carry = fma(pw[4], e2, r3);
  double ep = r0;
  ep = lane >= 32 ? e1 : ep;
  ep = lane >= 48 ? e2 : ep;
  return lane >= 16 ? fma(q, ep, z) : z;
}
// the same for per-lane ratios (a ragged tile): the scan composes the affine maps (al, z)
__device__ inline double bank_scan_affine(double al, double z, int lane) {
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const double zt = __shfl_up(z, 1 << s), at = __shfl_up(al, 1 << s);
    if (lane >= (1 << s)) { z = fma(al, zt, z); al *= at; }
  }
  return z;
}
// One one-pole stage over the lane's chunk x[0 .. cnt), IN PLACE: x[j] becomes the stage's output sample — the low-pass y (OUT_LP), the
// high-pass x - y (OUT_HP).  Returns the tile's carry-out
// (wave-uniform).  k: the coefficient; with P = (1 - k)^16, the factor a state decays by over one lane's chunk: pw[s] = P^(2^s), s = 0 .. 5
// (pw[0] carries a state over one lane, pw[4] over a row of 16), q = P^((lane mod 16) + 1); y_in: the state before the tile's first sample (wave-uniform).
enum { OUT_LP, OUT_HP };
template <bool RAGGED, int OUT, bool IN_ABS>
__device__ inline double bank_stage(double (&x)[SC], double k, const double *__restrict__ pw, double q, double y_in, int cnt, int lane) {
  // IN_ABS: the stage is fed |x| (the envelope follower after the band-pass): a source modifier of the instructions that read x, no instruction
  // (1) the chunk's zero-state response z = k sum_j a^(len-1-j) x[j], by Horner's rule: one fused multiply-add per sample (it only feeds the
  // scan: its rounding is the carry's, not the samples')
  double z = 0.0, al = 1.0;
  const double a = 1.0 - k;
#pragma unroll
  for (int j = 0; j < SC; ++j) {
    if (!RAGGED || j < cnt) { z = fma(z, a, IN_ABS ? fabs(x[j]) : x[j]); if (RAGGED) al *= a; }
  }
  z *= k;
  // (2) the state at the end of every chunk, and from it every chunk's start state
  double Y, carry;
  if (RAGGED) {
    if (lane == 0) z = fma(al, y_in, z);
    Y = bank_scan_affine(al, z, lane);
    carry = readlane_f64(Y, 63); // (lanes past the ragged tile's end hold the identity map: the last lane carries the state after the last sample)
  } else {
    if (lane == 0) z = fma(pw[0], y_in, z);
    Y = bank_scan_const(z, pw, q, lane, carry);
  }
  double yy = dpp_f64<0x138, false>(y_in, Y); // wave_shr:1 — lane l takes lane l - 1's end state, lane 0 keeps y_in
  // (3) the recurrence y += k (x - y) from the chunk's true start state, the multiply-add fused (the library is built without contraction)
#pragma unroll
  for (int j = 0; j < SC; ++j) {
    const double xin = IN_ABS ? fabs(x[j]) : x[j];
    if (!RAGGED || j < cnt) yy = fma(k, xin - yy, yy);
    x[j] = OUT == OUT_HP ? xin - yy : yy;
  }
  return carry;
}

// one band of a tile for this lane's chunk.  in_m / in_c: the tile's inputs in LDS (sample (lane, j) at j * SROW + lane), read again for every
// band; out: the wave's running band sum, in its own slice of LDS.  Two 16-sample arrays are all the registers hold: with the inputs and the sum
// in registers too the kernel needed 239 VGPRs, which left the carrier synth's kernels it runs beside (next block's walk — 249 VGPRs — and render,
// on their own queue) no room on a SIMD: measured, both stretched three-fold.
template <bool RAGGED>
__device__ inline void bank_band(const double *__restrict__ in_m, const double *__restrict__ in_c, double *__restrict__ out, bool first, const double *__restrict__ qtab, double kA, double kB, double kE,
                                 const double *__restrict__ pwA, const double *__restrict__ pwB, const double *__restrict__ pwE, double *__restrict__ st, int cnt, int lane) {
  double v[SC], u[SC];
  // P^((lane mod 16) + 1) for the three coefficients: what a later row's lane owes its predecessor row's end state (bank_scan_const)
  const double qA = qtab[lane & 15], qB = qtab[16 + (lane & 15)], qE = qtab[32 + (lane & 15)];
  // Modulator::feed (vocoder.cpp:358-380): BandPass::feed (:54-58) = high-pass (input - low-pass) then low-pass; EnvelopeFollower::feed (:71-74) of |.|
#pragma unroll
  for (int j = 0; j < SC; ++j) v[j] = in_m[j * SROW + lane];
  st[0] = bank_stage<RAGGED, OUT_HP, false>(v, kA, pwA, qA, st[0], cnt, lane);
  st[1] = bank_stage<RAGGED, OUT_LP, false>(v, kB, pwB, qB, st[1], cnt, lane);
  st[2] = bank_stage<RAGGED, OUT_LP, true>(v, kE, pwE, qE, st[2], cnt, lane);
  // Carrier::feed (:698-718): the same band-pass; times the envelope
#pragma unroll
  for (int j = 0; j < SC; ++j) u[j] = in_c[j * SROW + lane];
  st[3] = bank_stage<RAGGED, OUT_HP, false>(u, kA, pwA, qA, st[3], cnt, lane);
  st[4] = bank_stage<RAGGED, OUT_LP, false>(u, kB, pwB, qB, st[4], cnt, lane);
  if (first) { // (wave-uniform) res += ponderation[i] * filtered, in band order
#pragma unroll
    for (int j = 0; j < SC; ++j) out[j * SROW + lane] = v[j] * u[j];
  } else {
#pragma unroll
    for (int j = 0; j < SC; ++j) out[j * SROW + lane] = fma(v[j], u[j], out[j * SROW + lane]);
  }
}

__global__ void __launch_bounds__(512, 4) ra_voc_bank_scan_kernel(RaParams p, RaBuffers b, const float *__restrict__ voice, const float *__restrict__ carrier_f,
                                                             const double *__restrict__ carrier_d, int64_t n) {
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, B = p.voc_bands;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bpw = (B + 7) >> 3;
  double *lds = reinterpret_cast<double *>(ra_smem_car);
  double *in_m = lds, *in_c = lds + SC * SROW, *part = lds + 2 * SC * SROW; // [SC][SROW] each; part: [8][SC][SROW]
  double *st_lds = part + 8 * SC * SROW;                                     // [64][5] the bands' filter states
  double *q_lds = st_lds + 5 * 64;                                           // [bands][3][16] P^(i + 1) per band and coefficient, P = (1 - k)^16
  double *gst = b.bank_state + (size_t)s * 5 * 64;
  for (int q = tid; q < 5 * 64; q += 512) st_lds[(q & 63) * 5 + (q >> 6)] = gst[q]; // global layout [5][64]
  const float *vs = voice ? voice + (size_t)s * n : nullptr;
  const float *cf = carrier_f ? carrier_f + (size_t)s * n : nullptr;
  const double *cd = carrier_d ? carrier_d + (size_t)s * n : nullptr;
  double *mix = b.voc_mix + (size_t)s * n;
  const double *pow_tab = b.bank_pow; // [3][8][64]: coefficient (band's lower edge, upper edge, follower), a^(16 * 2^e), band
  for (int q = tid; q < B * 3; q += 512) { // this entry's row powers P, P^2, ... P^16 by repeated multiplication (once per launch)
    const int band = q / 3, t = q - band * 3;
    const double P = pow_tab[(t * 8 + 0) * 64 + band];
    double acc = P;
    for (int i = 0; i < 16; ++i) { q_lds[(band * 3 + t) * 16 + i] = acc; acc *= P; }
  }
  auto sample_pos = [](int i) { return (i & (SC - 1)) * SROW + (i >> 4); };
  for (int64_t tb = 0; tb < n; tb += STILE) {
    const int cnt_tile = (int)min((int64_t)STILE, n - tb);
    // stage the tile's inputs (the modulator as the double read_direct_input widens it to, lib.cpp:1073-1079): coalesced loads, transposed in LDS
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = tid + 512 * h;
      double mv = 0.0, cv = 0.0;
      if (i < cnt_tile) {
        if (vs) mv = (double)vs[tb + i];
        if (cd) cv = cd[tb + i]; else if (cf) cv = (double)cf[tb + i];
      }
      in_m[sample_pos(i)] = mv; in_c[sample_pos(i)] = cv;
    }
    __syncthreads();
    double *out = part + (size_t)w * SC * SROW;
    if (w * bpw >= B) { // a wave without bands (fewer than eight bands' worth): its slice adds zeros
#pragma unroll
      for (int j = 0; j < SC; ++j) out[j * SROW + lane] = 0.0;
    }
    const int cnt = max(0, min(SC, cnt_tile - SC * lane));
    for (int bi = 0; bi < bpw; ++bi) {
      const int band = w * bpw + bi;
      if (band >= B) break; // wave-uniform
      // (everything per band is wave-uniform: kept in scalar registers, the vector registers belong to the two sample arrays)
      const double kA = uniform_f64(b.bank_k[band]), kB = uniform_f64(b.bank_k[64 + band]), kE = uniform_f64(b.bank_k[128 + band]);
      double pwA[6], pwB[6], pwE[6], st[5];
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        pwA[e] = uniform_f64(pow_tab[(0 * 8 + e) * 64 + band]); pwB[e] = uniform_f64(pow_tab[(1 * 8 + e) * 64 + band]); pwE[e] = uniform_f64(pow_tab[(2 * 8 + e) * 64 + band]);
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) st[q] = uniform_f64(st_lds[band * 5 + q]);
      if (cnt_tile == STILE) bank_band<false>(in_m, in_c, out, bi == 0, q_lds + band * 48, kA, kB, kE, pwA, pwB, pwE, st, cnt, lane);
      else bank_band<true>(in_m, in_c, out, bi == 0, q_lds + band * 48, kA, kB, kE, pwA, pwB, pwE, st, cnt, lane);
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) st_lds[band * 5 + q] = st[q];
      }
    }
    __syncthreads();
    // the band sum across the waves, in wave (= band) order, and the mix (vocoder.cpp:799-803)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = tid + 512 * h;
      if (i < cnt_tile) {
        const int at = sample_pos(i);
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += part[q * SC * SROW + at];
        mix[tb + i] = (double)p.vol_mod * in_m[at] + (double)p.vol_car * in_c[at] + (double)p.vol_voc * acc;
      }
    }
    // (the next tile's staging writes only this thread's own two input slots; its partials are written behind the next barrier)
  }
  __syncthreads();
  for (int q = tid; q < 5 * 64; q += 512) gst[q] = st_lds[(q & 63) * 5 + (q >> 6)];
}

} // namespace

int ra_carrier_seg_doubles() { return CS_COUNT; }

void ra_launch_carrier_synth(const RaParams &p, const RaBuffers &b, int64_t t0, int64_t n, int64_t total_segs, hipStream_t st) {
  if (n <= 0 || p.car_P <= 0) return;
  if (p.car_serial) { // reference_order 1: one sample-serial launch
    const size_t lds = (size_t)p.car_Ppad * (CT + 1) * sizeof(double) + 4 * sizeof(double) + 4 * sizeof(int);
    hipLaunchKernelGGL(ra_carrier_synth_serial_kernel, dim3((unsigned)p.n_streams), dim3((unsigned)p.car_Ppad), lds, st, p, b, t0, n);
    return;
  }
  hipLaunchKernelGGL(ra_carrier_walk_kernel, dim3((unsigned)p.n_streams), dim3((unsigned)p.car_Ppad), 4 * sizeof(double) + 4 * sizeof(int), st, p, b, t0, n);
  const int nth = std::max(64, p.car_CH), sub = nth / p.car_CH;
  const size_t lds = (size_t)nth * (CT + 1) * sizeof(double) + (size_t)sub * sizeof(int4) + 16;
  hipLaunchKernelGGL(ra_carrier_render_kernel, dim3((unsigned)((total_segs + sub - 1) / sub)), dim3((unsigned)nth), lds, st, p, b, n, total_segs);
}

void ra_launch_voc_bank(const RaParams &p, const RaBuffers &b, const float *d_voice, const float *d_carrier, const double *carrier_d, int64_t t0, int64_t n,
                        hipStream_t st) {
  if (n <= 0) return;
  (void)t0;
  static const bool force_pipe = std::getenv("RA_BANK_PIPE") != nullptr; // tuning aid: the bit-identical pipeline kernel whatever reference_order says
  if (!p.car_serial && !force_pipe && !std::getenv("RA_BANK_SINGLE_WAVE")) { // the default: time-parallel (results equal the reference's loop to rounding)
    static unsigned long long scan_attr_done = 0;
    const size_t lds = ((size_t)(2 + 8) * SC * SROW + 5 * 64 + 3 * 16 * 64) * sizeof(double);
    ra_once_per_device(scan_attr_done, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voc_bank_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RA_LDS_LIMIT_BYTES);
    });
    hipLaunchKernelGGL(ra_voc_bank_scan_kernel, dim3((unsigned)p.n_streams), dim3(512), lds, st, p, b, d_voice, d_carrier, carrier_d, n);
    return;
  }
  static const bool single_wave = std::getenv("RA_BANK_SINGLE_WAVE") != nullptr; // tuning aid: the one-wave kernel (same bits, ~3x the time)
  if (single_wave) {
    hipLaunchKernelGGL(ra_voc_bank_kernel, dim3((unsigned)p.n_streams), dim3(64), (size_t)BT * 65 * sizeof(double), st, p, b, d_voice, d_carrier, carrier_d, n);
    return;
  }
  static const int perm = [] {
    int v = 0x132540; // waves 0..5 run stages 0, 4, 5, 2, 3, 1 (measured on configs[2]: 47.2 ms per step; in stage order 51.7; in-kernel clocks:
                      // every stage then works ~1500 cycles per 16-sample tile, the floor of three dependent fp64 operations per sample is ~1100)
    if (const char *e = std::getenv("RA_BANK_PERM")) { v = 0; for (int i = 0; i < 6 && e[i]; ++i) v |= (e[i] - '0') << (4 * i); }
    return v;
  }();
  static unsigned long long attr_done = 0; // 81 KB of dynamic LDS: above the 64 KB default
  ra_once_per_device(attr_done, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra_voc_bank_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RA_LDS_LIMIT_BYTES);
  });
  hipLaunchKernelGGL(ra_voc_bank_pipe_kernel, dim3((unsigned)p.n_streams), dim3(384), (size_t)5 * 2 * PEDGE * sizeof(double), st, p, b, d_voice, d_carrier, carrier_d, n, perm);
}
