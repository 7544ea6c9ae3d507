// Does the memory system come back slowly after the device has been idle?  Short streaming kernels (each reads
// 2 GiB, writes 1 GiB) in a row after an idle period; per kernel the hipEvent duration -> GB/s.
// Usage: mem_ramp [idle_ms ...]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ __launch_bounds__(256) void stream(const double2 *a, const double2 *b, double2 *c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 x = a[i], y = b[i];
    c[i] = double2{x.x + y.x, x.y + y.y};
  }
}

int main(int argc, char **argv) {
  const size_t n = (size_t)1 << 26;                 // 64 M double2 = 1 GiB per array
  double2 *a, *b, *c;
  hipMalloc(&a, n * 16);
  hipMalloc(&b, n * 16);
  hipMalloc(&c, n * 16);
  hipMemset(a, 0, n * 16);
  hipMemset(b, 0, n * 16);
  hipMemset(c, 0, n * 16);
  const int K = 40;
  std::vector<hipEvent_t> ev(K + 1);
  for (auto &e : ev) hipEventCreate(&e);
  auto run = [&](const char *tag) {
    hipEventRecord(ev[0]);
    for (int i = 0; i < K; ++i) {
      stream<<<2048, 256>>>(a, b, c, n);
      hipEventRecord(ev[i + 1]);
    }
    hipDeviceSynchronize();
    printf("%s: TB/s per kernel:", tag);
    float t = 0;
    for (int i = 0; i < K; ++i) {
      float ms;
      hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
      t += ms;
      if (i < 12 || i % 8 == 7) printf(" %.2f", 3.0 * n * 16 / (ms * 1e-3) / 1e12);
    }
    printf("  (%.1f ms in all)\n", t);
  };
  run("cold start       ");
  run("back to back     ");
  for (int i = 1; i < argc; ++i) {
    const int ms = atoi(argv[i]);
    std::this_thread::sleep_for(std::chrono::milliseconds(ms));
    char tag[64];
    snprintf(tag, sizeof tag, "after %4d ms idle", ms);
    run(tag);
  }
  return 0;
}
