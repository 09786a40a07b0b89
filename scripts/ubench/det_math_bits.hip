// Checks on the GPU box that csrc/ra_math.cuh (device) and oracle/oracle_math.hpp (host) produce the same bits.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I cpp_audio_amd/csrc -I oracle scripts/ubench/det_math_bits.hip -o /tmp/det_math_bits && /tmp/det_math_bits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "ra_math.cuh"
#include "oracle_math.hpp"

__global__ void eval(const double *x, double *l2, double *l10, double *e10, double *dv, double *pw, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  l2[i] = ra_log2(x[i]);
  l10[i] = 10.0 * ra_log10(x[i] > 1e-300 ? x[i] : 1e-300);
  double y = 20.0 * (l2[i] * 0.01) - 3.0;
  e10[i] = ra_exp10(y / 20.0);
  dv[i] = (x[i] - 1.0) / (x[i] + 1.0);
  pw[i] = 440.0 * ra_pow(1.0594630943592953, 140.0 * (l2[i] - floor(l2[i])) - 69.0);
}

int main() {
  const int n = 1 << 22;
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> u(-690.0, 27.0);
  std::vector<double> x(n), l2(n), l10(n), e10(n), dv(n), pw(n);
  for (int i = 0; i < n; ++i) x[i] = std::exp(u(g));
  double *dx, *d2, *d10, *de, *dd, *dp;
  (void)hipMalloc(&dp, n * 8);
  (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&d2, n * 8); (void)hipMalloc(&d10, n * 8); (void)hipMalloc(&de, n * 8); (void)hipMalloc(&dd, n * 8);
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  eval<<<(n + 255) / 256, 256>>>(dx, d2, d10, de, dd, dp, n);
  (void)hipMemcpy(pw.data(), dp, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(l2.data(), d2, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(l10.data(), d10, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(e10.data(), de, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(dv.data(), dd, n * 8, hipMemcpyDeviceToHost);
  long b2 = 0, b10 = 0, be = 0, bd = 0, bp = 0;
  for (int i = 0; i < n; ++i) {
    double h2 = ora::det_log2(x[i]);
    double h10 = 10.0 * ora::det_log10(x[i] > 1e-300 ? x[i] : 1e-300);
    double y = 20.0 * (h2 * 0.01) - 3.0;
    double he = ora::det_exp10(y / 20.0);
    double hd = (x[i] - 1.0) / (x[i] + 1.0);
    double hp = 440.0 * ora::det_pow(1.0594630943592953, 140.0 * (h2 - std::floor(h2)) - 69.0);
    bp += std::memcmp(&hp, &pw[i], 8) != 0;
    b2 += std::memcmp(&h2, &l2[i], 8) != 0; b10 += std::memcmp(&h10, &l10[i], 8) != 0; be += std::memcmp(&he, &e10[i], 8) != 0; bd += std::memcmp(&hd, &dv[i], 8) != 0;
    if ((std::memcmp(&he, &e10[i], 8) != 0) && be < 4) printf("exp10 differs: y/20 = %a host %a dev %a\n", y / 20.0, he, e10[i]);
    if ((std::memcmp(&h2, &l2[i], 8) != 0) && b2 < 4) printf("log2 differs: x = %a host %a dev %a\n", x[i], h2, l2[i]);
  }
  printf("mismatching results of %d: log2 %ld, 10 log10 %ld, exp10 %ld, division %ld, pow %ld\n", n, b2, b10, be, bd, bp);
  return (b2 || b10 || be || bd || bp) ? 1 : 0;
}
