// Micro-benchmark: v_mfma_f64_16x16x4_f64 rate vs waves per SIMD, with operands
// in registers and with the B operand re-read from LDS before every MFMA (the
// projection kernel's pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ void k(double *out, int iters, double seed) {
  __shared__ double xs[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) xs[i] = seed * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f64x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f64x4){seed, 0.5, 0.25, 1.0};
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const double *row = xs + ((it & 15) * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double bb = LDS ? row[(lane >> 4) * 64 + i * 16 + (lane & 15)] : b;
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[i], 0, 0, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool LDS>
void run(int threads, int blocks_per_cu, int iters) {
  const int blocks = 256 * blocks_per_cu;
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<LDS><<<blocks, threads>>>(out, iters, 1e-3);
  hipDeviceSynchronize();
  float ms = 0;
  int reps = 0;
  hipEventRecord(e0);
  do {
    for (int w = 0; w < 10; ++w) k<LDS><<<blocks, threads>>>(out, iters, 1e-3);
    reps += 10;
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 700.f);
  double flops = (double)blocks * (threads / 64) * iters * 4.0 * 2048.0;
  double t = ms * 1e-3 / reps;
  printf("%s block=%4d threads x %d per CU (%d waves/SIMD): %.1f TFLOP/s\n", LDS ? "LDS-fed " : "register", threads,
         blocks_per_cu, threads / 256 * blocks_per_cu, flops / t / 1e12);
  hipFree(out);
}

int main() {
  run<false>(256, 1, 8000);
  run<false>(256, 2, 4000);
  run<false>(512, 1, 4000);
  run<false>(256, 3, 4000);
  run<false>(1024, 1, 2000);
  run<false>(256, 4, 2000);
  run<true>(256, 1, 8000);
  run<true>(256, 2, 4000);
  run<true>(512, 1, 4000);
  run<true>(256, 3, 4000);
  run<true>(256, 4, 2000);
  return 0;
}
