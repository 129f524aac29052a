// Device-to-device copy loop for tools/r4_power.sh-style sampling: what a pure HBM stream (read + write, no matrix work) draws.
//   tools/copy_power [seconds=5] [GiB per copy=2]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
  const size_t bytes = (size_t)((argc > 2 ? atof(argv[2]) : 2.0) * (1ull << 30));
  char *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const auto t0 = std::chrono::steady_clock::now();
  float ms = 0; long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); n += 20;
  }
  printf("device-to-device copy of %.1f GiB: %ld copies, the last 20: %.3f ms each = %.2f TB/s read + written\n", bytes / 1073741824.0, n, ms / 20,
         2.0 * bytes / (ms / 20 * 1e-3) / 1e12);
  return 0;
}
