// Micro-benchmark for a register-resident B operand: the X fragments of a wave's
// 64 voxels (nk = 15 k-steps x 4 voxel tiles = 60 doubles per lane) stay in
// registers, the A operand (operator fragment) is read from LDS once per k-step
// and feeds four MFMAs.  Question: does this reach the register-fed rate
// (77 TFLOP/s) where one ds_read_b64 per MFMA gives 58-71?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int NK = 15, NT = 4;

__global__ __launch_bounds__(256, 2) void k(const double *xin, double *out, int tiles, double seed) {
  __shared__ double as[64 * NK * 8];           // 8 tiles of fragments
  for (int i = threadIdx.x; i < 64 * NK * 8; i += blockDim.x) as[i] = seed * (i % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double xb[NK][NT];
#pragma unroll
  for (int s = 0; s < NK; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xb[s][nt] = xin[(s * NT + nt) * 64 + lane];
  double tot = 0.0;
  for (int t = 0; t < tiles; ++t) {
    f64x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const double *frag = as + (t & 7) * NK * 64 + lane;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const double a = frag[s * 64];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[s][nt], acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) tot += acc[nt][0] + acc[nt][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}

void run(int blocks_per_cu, int tiles) {
  const int blocks = 256 * blocks_per_cu, threads = 256;
  double *out, *xin;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&xin, sizeof(double) * NK * NT * 64);
  hipMemset(xin, 0, sizeof(double) * NK * NT * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<<<blocks, threads>>>(xin, out, tiles, 1e-3);
  hipDeviceSynchronize();
  float ms = 0;
  int reps = 0;
  hipEventRecord(e0);
  do {
    for (int w = 0; w < 10; ++w) k<<<blocks, threads>>>(xin, out, tiles, 1e-3);
    reps += 10;
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 700.f);
  const double flops = (double)blocks * 4 * tiles * NK * NT * 2048.0;
  printf("B in registers, A from LDS once per k-step, %d waves/SIMD: %.1f TFLOP/s\n", blocks_per_cu,
         flops / (ms * 1e-3 / reps) / 1e12);
  hipFree(out);
  hipFree(xin);
}

int main() {
  run(1, 400);
  run(2, 200);
  return 0;
}
