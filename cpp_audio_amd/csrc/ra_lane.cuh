// Cross-lane moves on the VALU (gfx950 permlane swaps / DPP): a ds_bpermute round trip costs ~100 cycles each.
#pragma once
#include <hip/hip_runtime.h>

__device__ inline double readlane_f64(double v, int lane) { // lane is wave-uniform; the result lives in scalar registers
  const long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffll), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ inline double wave_shr1_f64(double own, double v) { // lane l takes lane l - 1's v (DPP wave_shr:1), lane 0 keeps `own`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(own), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(own), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_shl1_f64(double own, double v) { // lane l takes lane l + 1's v (DPP wave_shl:1), lane 63 keeps `own`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(own), __double2loint(v), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(own), __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double lane_xor32_low(double x) { // value of lane ^ 32, valid in lanes 0..31
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ inline double lane_xor16_even(double x) { // value of lane ^ 16, valid in the even rows of 16 lanes
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ inline double lane_ror8(double x) { // value of lane (i + 8) mod 16 of the same row
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ inline double lane_ror(double x) { // value of lane (i + N) mod 16 of the same row
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over the wavefront's 64 lanes, valid in lanes 0..15 (fixed pairing order: reproducible)
__device__ inline double wave_sum_row0(double x) {
  x += lane_xor32_low(x);
  x += lane_xor16_even(x);
  x += lane_ror<8>(x);
  x += lane_ror<4>(x);
  x += lane_ror<2>(x);
  x += lane_ror<1>(x);
  return x;
}
