// What shader clock does a lone small workgroup get?  (tools/, tuning only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long* out, int iters) {
  const long long c0 = clock64(), r0 = wall_clock64();
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;
  const long long c1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (long long)a; }
}
int main() {
  long long* d; hipMalloc(&d, 24);
  long long h[3];
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, 20000);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("lone workgroup: %lld shader cycles in %.1f us -> %.2f GHz (%.1f cycles per dependent fp64 FMA)\n", h[0], h[1] / 100.0,
           h[0] / (h[1] / 100.0) / 1e3, (double)h[0] / 20000);
  }
  return 0;
}
