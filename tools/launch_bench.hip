// Host launch rate and dependent-kernel spacing of this runtime: N launches of a trivial kernel on one stream.
// build: hipcc --offload-arch=gfx950 -O3 tools/launch_bench.hip -o tools/launch_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(double* p, int n) { if (threadIdx.x == 0 && n < 0) p[0] = 1.0; }
__global__ void small64(double* p) {           // ~64x64 worth of work in one workgroup
  __shared__ double s[256];
  double a = p[threadIdx.x];
  for (int i = 0; i < 200; ++i) a = a * 1.0000001 + 1e-9;
  s[threadIdx.x] = a; __syncthreads();
  p[threadIdx.x] = s[(threadIdx.x + 1) & 255];
}
int main() {
  double* d; hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  for (int variant = 0; variant < 4; ++variant) {
    const int N = 2000;
    hipStreamSynchronize(st); hipStreamSynchronize(s2);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      if (variant == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, i);
      if (variant == 1) hipLaunchKernelGGL(small64, dim3(1), dim3(256), 0, st, d);
      if (variant == 2) { hipLaunchKernelGGL(small64, dim3(1), dim3(256), 0, st, d); (void)hipGetLastError(); }
      if (variant == 3) hipLaunchKernelGGL(small64, dim3(1), dim3(256), 0, (i & 1) ? st : s2, d + 512 * (i & 1));
    }
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(st); hipStreamSynchronize(s2);
    auto t2 = std::chrono::steady_clock::now();
    const double host = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
    const double all = std::chrono::duration<double, std::micro>(t2 - t0).count() / N;
    printf("variant %d: host %.2f us per launch, launch+drain %.2f us per kernel\n", variant, host, all);
  }
  return 0;
}
