// How fast does the shader clock come back after the device has been idle?  A grid that fills the chip with
// fp64 MFMA work records, per ~40 us slice, core cycles (clock64) against the 100 MHz wall clock (wall_clock64):
// their ratio is the clock the CUs actually ran at.  Usage: clock_ramp [idle_ms ...]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void burn(long long *log, int slices, int per_slice, double *sink) {
  f64x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int s = 0; s < slices; ++s) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      log[2 * s] = clock64();
      log[2 * s + 1] = wall_clock64();
    }
    for (int it = 0; it < per_slice; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double t = 0;
  for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * 256 + threadIdx.x] = t;
}

int main(int argc, char **argv) {
  const int slices = 1500, per_slice = 400;         // 1600 MFMAs of 64 cycles per slice = 43 us at 2.4 GHz
  long long *log;
  double *sink;
  hipMalloc(&log, sizeof(long long) * 2 * slices);
  hipMalloc(&sink, sizeof(double) * 1024 * 256);
  std::vector<long long> h(2 * slices);
  auto run = [&](const char *tag) {
    burn<<<1024, 256>>>(log, slices, per_slice, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), log, sizeof(long long) * 2 * slices, hipMemcpyDeviceToHost);
    printf("%s: MHz at t =", tag);
    const double t0 = h[1];
    int next_ms = 0;
    for (int s = 1; s < slices; ++s) {
      const double t_ms = (h[2 * s + 1] - t0) / 1e5;     // 100 MHz ticks -> ms
      if (t_ms >= next_ms) {
        const int w = s >= 20 ? 20 : s;                  // average over the last slices
        const double mhz = (double)(h[2 * s] - h[2 * (s - w)]) / ((h[2 * s + 1] - h[2 * (s - w) + 1]) / 100.0);
        printf(" %dms:%.0f", next_ms, mhz);
        next_ms += next_ms < 10 ? 1 : 5;
      }
    }
    printf("  (kernel %.1f ms)\n", (h[2 * slices - 1] - t0) / 1e5);
  };
  run("cold start        ");
  run("back to back      ");
  for (int i = 1; i < argc; ++i) {
    const int ms = atoi(argv[i]);
    std::this_thread::sleep_for(std::chrono::milliseconds(ms));
    char tag[64];
    snprintf(tag, sizeof tag, "after %4d ms idle ", ms);
    run(tag);
    run("back to back      ");
  }
  return 0;
}
