// Micro-benchmark + layout probe for v_mfma_f64_4x4x4_4b_f64 on gfx950:
//  (1) sustained rate vs v_mfma_f64_16x16x4_f64 (is the small shape cheaper per
//      instruction? 512 vs 2048 flop);
//  (2) operand / result lane layout, found with one-hot inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void rate4(double *out, int iters) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = 0.001 * i;
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void probe(const double *A, const double *B, double *D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}

int main() {
  const int blocks = 256;
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int w = 0; w < 3; ++w) rate4<<<blocks, 512>>>(out, iters);
  hipDeviceSynchronize();
  float ms = 0;
  int reps = 0;
  hipEventRecord(e0);
  do {
    for (int w = 0; w < 10; ++w) rate4<<<blocks, 512>>>(out, iters);
    reps += 10;
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 700.f);
  double ninst = (double)blocks * 8 * iters * 8.0 * reps;
  double t = ms * 1e-3;
  printf("v_mfma_f64_4x4x4_4b: %.2f TFLOP/s (512 flop each), %.1f cycles per instruction per SIMD at 2.39 GHz\n",
         ninst * 512.0 / t / 1e12, t * 2.39e9 / (ninst / 1024.0));

  // layout probe: one-hot A lane x, all-ones B, and vice versa
  double *dA, *dB, *dD;
  hipMalloc(&dA, 64 * 8);
  hipMalloc(&dB, 64 * 8);
  hipMalloc(&dD, 64 * 8);
  std::vector<double> hA(64), hB(64), hD(64);
  printf("A one-hot lane -> result lanes that see it (B = lane id + 1 so the partner k is visible):\n");
  for (int x = 0; x < 64; x += 1) {
    for (int i = 0; i < 64; ++i) { hA[i] = (i == x); hB[i] = 1 + i; }
    hipMemcpy(dA, hA.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), 512, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD.data(), dD, 512, hipMemcpyDeviceToHost);
    printf("A[%2d]:", x);
    for (int i = 0; i < 64; ++i) if (hD[i] != 0) printf(" D%d=B%d", i, (int)hD[i] - 1);
    printf("\n");
  }
  return 0;
}
