// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on the whole chip,
// the practical ceiling the projection kernel is priced against.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
// Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime) for
// 1, 2 and 4 waves per SIMD with NACC independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, unsigned long long *clk, int iters, double seed) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){seed * i, 0.5, 0.25, 1.0};
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
  int cus = 256;
  int blocks = cus * blocks_per_cu;
  double *out;
  unsigned long long *clk;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipMalloc(&clk, sizeof(unsigned long long) * blocks * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) mfma_loop<NACC><<<blocks, 256>>>(out, clk, iters, 0.001);
  hipDeviceSynchronize();
  // >= 1.5 s of back-to-back launches so the clock settles (DVFS)
  float ms = 0;
  int reps = 0;
  hipEventRecord(e0);
  do {
    for (int w = 0; w < 20; ++w) mfma_loop<NACC><<<blocks, 256>>>(out, clk, iters, 0.001);
    reps += 20;
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 1500.f);
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), clk, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
  std::vector<double> ghz(blocks);
  for (int i = 0; i < blocks; ++i) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;  // memrealtime = 100 MHz
  std::sort(ghz.begin(), ghz.end());
  double flops = (double)reps * blocks * 4 /*waves*/ * (double)iters * NACC * 2048.0;
  double tf = flops / (ms * 1e-3) / 1e12;
  double cyc_per_mfma = (double)h[0] / ((double)iters * NACC);
  printf("NACC=%d waves/SIMD=%d : %.2f TFLOP/s, in-kernel clock median %.3f GHz, %.1f cycles per MFMA per wave\n",
         NACC, blocks_per_cu, tf, ghz[blocks / 2], cyc_per_mfma);
  hipFree(out);
  hipFree(clk);
}

int main() {
  run<4>(1, 20000);
  run<8>(1, 10000);
  run<4>(2, 10000);
  run<4>(4, 5000);
  return 0;
}
