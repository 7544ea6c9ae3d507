// Micro-benchmark: achievable HBM3E bandwidth on this box for the two patterns the
// reductions use -- a streaming read (sum) and a streaming copy -- on buffers far
// larger than the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void read_sum(const double2 *in, double *out, size_t n2) {
  double a = 0.0, b = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const double2 v = in[i];
    a += v.x;
    b += v.y;
  }
  if (a + b == 12345.678) out[0] = a;          // keep the loads
}

__global__ __launch_bounds__(256) void copy(const double2 *in, double2 *out, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

int main() {
  const size_t bytes = (size_t)8 << 30;        // 8 GiB per buffer
  const size_t n2 = bytes / sizeof(double2);
  double2 *a, *b;
  double *o;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 8) != hipSuccess) return 1;
  (void)hipMemset(a, 0, bytes);
  (void)hipMemset(b, 0, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int blocks : {2048, 8192, 32768}) {
    float ms;
    read_sum<<<blocks, 256>>>(a, o, n2);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) read_sum<<<blocks, 256>>>(a, o, n2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("read  %6d blocks: %.2f TB/s\n", blocks, 5.0 * bytes / (ms * 1e-3) / 1e12);
    copy<<<blocks, 256>>>(a, b, n2);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) copy<<<blocks, 256>>>(a, b, n2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("copy  %6d blocks: %.2f TB/s (read + write)\n", blocks, 5.0 * 2 * bytes / (ms * 1e-3) / 1e12);
  }
  return 0;
}
