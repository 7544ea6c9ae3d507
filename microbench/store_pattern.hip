// Micro-benchmark: HBM write rate of the VS^T store pattern of K4a / K4b.  Every workgroup (256
// threads = 64 voxels) writes, per item, 48 rows x 512 B; rows are ldv * 8 bytes apart (row
// layout) or contiguous per workgroup (blocked layout).  No compute.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int BLOCKED>
__global__ __launch_bounds__(256) void k(double *out, long p, int items, int rows, int it0) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, g = lane >> 4;
  const long v = (long)blockIdx.x * 64 + wave * 16 + (col & ~1);
  for (int it = it0; it < it0 + items; ++it) {
    for (int q = 0; q < rows / 8; ++q) {                 // one 16-byte store instruction per 8 rows, as the kernels do
      const int j = q * 8 + g + 4 * (col & 1);
      double *dst;
      if (BLOCKED) dst = out + (((long)it * gridDim.x + blockIdx.x) * rows + j) * 64 + wave * 16 + (col & ~1);
      else dst = out + ((long)it * rows + j) * p + v;
      *(d2 *)dst = (d2){(double)it, (double)j};
    }
  }
}

int main() {
  const long p = 200000;
  const int rows = 48, items = 64, wgs = 3125;
  double *out;
  if (hipMalloc(&out, sizeof(double) * p * rows * items) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocked = 0; blocked < 2; ++blocked) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (blocked) k<1><<<wgs, 256>>>(out, p, items, rows, 0);
      else k<0><<<wgs, 256>>>(out, p, items, rows, 0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s layout: %.3f ms for %.2f GB = %.2f TB/s\n", blocked ? "blocked" : "row", ms, p * rows * items * 8e-9,
             p * rows * items * 8e-12 / (ms * 1e-3) * 1e0);
    }
  }
  return 0;
}
