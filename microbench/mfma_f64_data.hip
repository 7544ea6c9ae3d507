// Micro-benchmark: does the sustained v_mfma_f64_16x16x4_f64 rate depend on the
// operand DATA?  (DVFS: MI355X lowers its clock under load; near-constant
// operands toggle few bits.)  Register operands, 2 waves per SIMD, operands
// either ~1.0 in every lane or pseudo-random in [-1,1) per lane and per step.
// Reports TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ inline double rnd(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return (double)(int)x * (1.0 / 2147483648.0);
}

template <bool RANDOM>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *clk, int iters) {
  f64x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = RANDOM ? rnd(threadIdx.x * 16 + i + blockIdx.x * 8192) : 1.0 + 1e-9 * threadIdx.x;
    b[i] = RANDOM ? rnd(threadIdx.x * 16 + i + 8 + blockIdx.x * 8192) : 1.0 - 1e-9 * threadIdx.x;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[(j + i) & 7], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <bool RANDOM>
void run(int iters) {
  const int blocks = 256;
  double *out; unsigned long long *clk;
  hipMalloc(&out, sizeof(double) * blocks * 512);
  hipMalloc(&clk, sizeof(unsigned long long) * blocks * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<RANDOM><<<blocks, 512>>>(out, clk, iters);
  hipDeviceSynchronize();
  float ms = 0; int reps = 0;
  hipEventRecord(e0);
  do {
    for (int w = 0; w < 10; ++w) k<RANDOM><<<blocks, 512>>>(out, clk, iters);
    reps += 10;
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 2000.f);
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), clk, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
  std::vector<double> ghz(blocks);
  for (int i = 0; i < blocks; ++i) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
  std::sort(ghz.begin(), ghz.end());
  double flops = (double)blocks * 8 * iters * 32.0 * 2048.0;
  double t = ms * 1e-3 / reps;
  printf("%s operands: %.1f TFLOP/s, in-kernel clock median %.3f GHz (min %.3f max %.3f), %.1f cycles per MFMA per SIMD\n",
         RANDOM ? "random  " : "constant", flops / t / 1e12, ghz[blocks / 2], ghz[0], ghz[blocks - 1],
         (double)h[0] / (iters * 32.0) / 2.0);
  hipFree(out); hipFree(clk);
}

int main() {
  run<false>(1000);
  run<true>(1000);
  run<false>(1000);
  run<true>(1000);
  return 0;
}
