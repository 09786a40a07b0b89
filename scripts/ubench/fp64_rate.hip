// Microbenchmark: issue rate and dependent latency of fp64 VALU ops on gfx950 (tuning aid, not part of the library).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off fp64_rate.hip -o fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K, int OP>
__global__ void chains(double *out, unsigned long long *cyc, int iters, double a, double b) {
  double x[K];
  for (int k = 0; k < K; ++k) x[k] = 1.0 + 1e-9 * (threadIdx.x + k);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (OP == 0) x[k] = fma(x[k], a, b);
      else if (OP == 1) x[k] = x[k] * a;
      else if (OP == 2) x[k] = x[k] + b;
      else { // complex rotation pair like the oscillator: 2 mul + 2 fma on a (re, im) pair per k
        double re = x[k], im = x[(k + 1) % K];
        double nr = fma(re, a, -(im * b)), ni = fma(re, b, im * a);
        x[k] = nr; x[(k + 1) % K] = ni;
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < K; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int OP>
void run(const char *name, int waves_per_simd) {
  int dev_cus = 256;
  int blocks = dev_cus * 4 * waves_per_simd;
  double *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(double) * blocks * 64);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  chains<K, OP><<<blocks, 64>>>(out, cyc, 100, 1.0000001, 1e-12);
  hipEventRecord(e0);
  chains<K, OP><<<blocks, 64>>>(out, cyc, iters, 1.0000001, 1e-12);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  int ops_per_iter = (OP == 3) ? 4 * K : K;
  printf("%-28s K=%2d waves/SIMD=%d : %.2f ticks/instr/wave, kernel %.3f ms, %.2f ns/instr/wave\n", name, K, waves_per_simd,
         avg / ((double)iters * ops_per_iter), ms, ms * 1e6 / ((double)iters * ops_per_iter));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<1, 0>("fma dependent", w);
    run<8, 0>("fma 8 independent", w);
    run<8, 1>("mul 8 independent", w);
    run<8, 2>("add 8 independent", w);
    run<2, 3>("rotation 1 pair (dependent)", w);
    run<8, 3>("rotation chained pairs", w);
  }
  return 0;
}
