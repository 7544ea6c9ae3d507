// Micro-benchmark: sustained v_fma_f64 rate (vector pipe), alone and beside
// v_mfma_f64_16x16x4_f64 issued by OTHER waves of the same SIMD, and both
// interleaved in ONE wave.  Answers: can the fp64 vector pipe add throughput
// on top of the matrix pipe?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// mode 0: all waves VALU; 1: all waves MFMA; 2: even waves MFMA, odd waves VALU;
// 3: every wave interleaves 1 MFMA : NV VALU fmas
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(double *out, int iters, double seed) {
  const int wave = threadIdx.x >> 6;
  double v[16];
  f64x4 acc[4];
  for (int i = 0; i < 16; ++i) v[i] = seed * (i + threadIdx.x);
  for (int i = 0; i < 4; ++i) acc[i] = (f64x4){seed, 0.5, 0.25, 1.0};
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  const bool do_mfma = MODE == 1 || (MODE == 2 && (wave & 1) == 0) || MODE == 3;
  const bool do_valu = MODE == 0 || (MODE == 2 && (wave & 1) == 1) || MODE == 3;
  if (MODE == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[(i * NV + j) & 15] = fma(v[(i * NV + j) & 15], a, b);
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(const char *name, int iters) {
  const int blocks = 256;  // one 8-wave block per CU = 2 waves per SIMD
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<MODE, NV><<<blocks, 512>>>(out, iters, 1e-3);
  hipDeviceSynchronize();
  float ms = 0;
  int reps = 0;
  hipEventRecord(e0);
  do {
    for (int w = 0; w < 10; ++w) k<MODE, NV><<<blocks, 512>>>(out, iters, 1e-3);
    reps += 10;
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 1000.f);
  double waves = (double)blocks * 8;
  double mf = 0, vf = 0;
  if (MODE == 0) vf = waves * iters * 64.0 * 128.0;
  if (MODE == 1) mf = waves * iters * 4.0 * 2048.0;
  if (MODE == 2) { mf = waves / 2 * iters * 4.0 * 2048.0; vf = waves / 2 * iters * 64.0 * 128.0; }
  if (MODE == 3) { mf = waves * iters * 4.0 * 2048.0; vf = waves * iters * 4.0 * NV * 128.0; }
  double t = ms * 1e-3 / reps;
  printf("%-44s mfma %.1f TF + valu %.1f TF = %.1f TF  (%.3f ms/launch)\n", name, mf / t / 1e12, vf / t / 1e12,
         (mf + vf) / t / 1e12, t * 1e3);
  hipFree(out);
}

int main() {
  run<0, 0>("VALU v_fma_f64 only, 2 waves/SIMD", 4000);
  run<1, 0>("MFMA f64 only, 2 waves/SIMD", 4000);
  run<2, 0>("split: 1 MFMA wave + 1 VALU wave per SIMD", 4000);
  run<3, 4>("interleaved in each wave, 4 fma per mfma", 4000);
  run<3, 8>("interleaved in each wave, 8 fma per mfma", 4000);
  run<3, 16>("interleaved in each wave, 16 fma per mfma", 4000);
  run<3, 24>("interleaved in each wave, 24 fma per mfma", 2000);
  return 0;
}
