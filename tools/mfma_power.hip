// Register-only fp64 matrix loop for tools/r4_power.sh: v_mfma_f64_4x4x4_4b_f64 on 8 independent accumulators per wave, 2 waves per SIMD
// on every CU, launched back to back for `seconds` so that rocm-smi can be sampled beside it: what the matrix pipeline draws and how it
// is clocked when nothing but registers moves.   tools/mfma_power [seconds=6] [random operands: 0 | 1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_mfma4(int iters, double* sink) {
  double acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0;
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  sink[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

// the same loop on operands that change from one MFMA to the next: eight (a, b) pairs of pseudo-random doubles per lane, read from
// memory once (so that the compiler cannot fold them), one pair per accumulator
__global__ __launch_bounds__(256) void k_mfma4_rand(int iters, double* sink, const double* ops) {
  double acc[8], a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = 0;
    a[i] = ops[(threadIdx.x * 16 + 2 * i) & 4095];
    b[i] = ops[(threadIdx.x * 16 + 2 * i + 1) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[i], acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  sink[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 6.0;
  const int random_ops = argc > 2 ? atoi(argv[2]) : 0;
  double* ops;
  {
    double h[4096];
    unsigned long long x = 88172645463325252ull;
    for (int i = 0; i < 4096; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = ((double)(x >> 11) / 9007199254740992.0 - 0.5) * 1e-3; }
    hipMalloc(&ops, sizeof(h));
    hipMemcpy(ops, h, sizeof(h), hipMemcpyHostToDevice);
  }
  int cu = 256;
  hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cu * 2, iters = 40000;        // 2 waves per SIMD; 8 x 40000 MFMAs per wave and launch
  double* sink;
  hipMalloc(&sink, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if (random_ops) hipLaunchKernelGGL(k_mfma4_rand, dim3(blocks), dim3(256), 0, 0, iters, sink, ops);
    else hipLaunchKernelGGL(k_mfma4, dim3(blocks), dim3(256), 0, 0, iters, sink);
  };
  launch();
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  float last_ms = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&last_ms, e0, e1);
    launches += 20;
  }
  const double flops = (double)blocks * 4 * 8 * iters * 512.0;
  printf("register-only v_mfma_f64_4x4x4_4b_f64 (%s operands), %d CUs x 2 waves per SIMD: %ld launches, the last 20: %.3f ms each = %.1f TFLOP/s\n",
         random_ops ? "eight pseudo-random pairs of" : "one pair of constant", cu, launches,
         last_ms / 20, flops / (last_ms / 20 * 1e-3) / 1e12);
  return 0;
}
