// Accuracy check of cpp_audio_amd/csrc/ra_math.cuh against the host libm (tuning aid).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off lean_math_check.hip -o lean_math_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../cpp_audio_amd/csrc/ra_math.cuh"

__global__ void k(const double *x, double *l2, double *e10, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { l2[i] = ra_log2(x[i]); e10[i] = ra_exp10(log10(x[i]) * 0.05); }
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), l2(n), e10(n);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> u(-300.0, 12.0), v(0.5, 2.0);
  for (int i = 0; i < n; ++i) x[i] = (i & 1) ? std::pow(10.0, u(g)) : v(g); // wide range and the neighbourhood of 1
  double *dx, *dl, *de;
  hipMalloc(&dx, n * 8); hipMalloc(&dl, n * 8); hipMalloc(&de, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dl, de, n);
  hipMemcpy(l2.data(), dl, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(e10.data(), de, n * 8, hipMemcpyDeviceToHost);
  double worst_l = 0, worst_e = 0;
  for (int i = 0; i < n; ++i) {
    long double ref = log2l((long double)x[i]);
    double ulp = std::fabs(std::nextafter((double)ref, INFINITY) - (double)ref);
    if (ref != 0) worst_l = std::fmax(worst_l, (double)(fabsl((long double)l2[i] - ref) / ulp));
    long double y = log10l((long double)x[i]) * 0.05L;   // the device formed y in double: compare against 10^(that double)
    double yd = std::log10(x[i]) * 0.05;
    long double re = powl(10.0L, (long double)yd);
    double ulpe = std::fabs(std::nextafter((double)re, INFINITY) - (double)re);
    worst_e = std::fmax(worst_e, (double)(fabsl((long double)e10[i] - re) / ulpe));
    (void)y;
  }
  std::printf("ra_log2 : worst error %.2f ulp\nra_exp10: worst error %.2f ulp (arguments from the device log10, so up to ~1 ulp of it is the argument)\n", worst_l, worst_e);
  return 0;
}
