// Micro-benchmark of fp64 matrix/vector issue rates on gfx950 (roofline denominator evidence).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bench.hip -o tools/mfma_bench && ./tools/mfma_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define STAMP_BEGIN long long t0 = clock64(), r0 = wall_clock64();
#define STAMP_END                                                   \
  long long t1 = clock64(), r1 = wall_clock64();                    \
  if ((threadIdx.x & 63) == 0) {                                    \
    cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;             \
    rt[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0;              \
  }

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(int iters, double* sink, long long* cyc, long long* rt) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  STAMP_BEGIN
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  STAMP_END
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(int iters, double* sink, long long* cyc, long long* rt) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  STAMP_BEGIN
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  STAMP_END
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  sink[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(int iters, double* sink, long long* cyc, long long* rt) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-9 * i;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  STAMP_BEGIN
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  STAMP_END
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  sink[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int blocks, int iters, int nacc, double flops_per_inst) {
  double* sink;
  long long *cyc, *rt;
  hipMalloc(&sink, (size_t)blocks * 256 * 8);
  hipMalloc(&cyc, blocks * 4 * 8);
  hipMalloc(&rt, blocks * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, sink, cyc, rt);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, sink, cyc, rt);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> hc(blocks * 4), hr(blocks * 4);
  hipMemcpy(hc.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), rt, blocks * 4 * 8, hipMemcpyDeviceToHost);
  std::sort(hc.begin(), hc.end());
  std::sort(hr.begin(), hr.end());
  const double medc = hc[hc.size() / 2], medr = hr[hr.size() / 2];
  const double inst = (double)iters * nacc;
  const double tf = (double)blocks * 4 * inst * flops_per_inst / (ms * 1e-3) / 1e12;
  printf("%-28s blocks=%5d  %6.1f TFLOP/s  cycles/inst/wave=%6.1f  shader clock=%.2f GHz  kernel %.2f ms\n", name, blocks, tf,
         medc / inst, medc / (medr / 100e6) / 1e9, ms);
  hipFree(sink);
  hipFree(cyc);
  hipFree(rt);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cu = p.multiProcessorCount;
  printf("device %s CUs=%d clockRate=%d kHz\n", p.gcnArchName, cu, p.clockRate);
  for (int wps : {1, 2, 4}) {
    const int blocks = cu * wps;   // 4 waves per block -> wps waves per SIMD
    printf("--- %d wave(s) per SIMD on every CU\n", wps);
    run("mfma_f64_16x16x4 nacc=1", k_mfma16<1>, blocks, 20000, 1, 2048.0);
    run("mfma_f64_16x16x4 nacc=2", k_mfma16<2>, blocks, 10000, 2, 2048.0);
    run("mfma_f64_16x16x4 nacc=4", k_mfma16<4>, blocks, 5000, 4, 2048.0);
    run("mfma_f64_16x16x4 nacc=8", k_mfma16<8>, blocks, 2500, 8, 2048.0);
    run("mfma_f64_4x4x4_4b nacc=4", k_mfma4<4>, blocks, 20000, 4, 512.0);
    run("mfma_f64_4x4x4_4b nacc=8", k_mfma4<8>, blocks, 10000, 8, 512.0);
    run("v_fma_f64 nacc=8", k_fma<8>, blocks, 40000, 8, 128.0);
    run("v_fma_f64 nacc=16", k_fma<16>, blocks, 20000, 16, 128.0);
  }
  printf("--- 8 blocks only (1 wave/SIMD on 8 CUs): does the clock rise when the chip is mostly idle?\n");
  run("mfma_f64_16x16x4 nacc=4", k_mfma16<4>, 8, 20000, 4, 2048.0);
  run("v_fma_f64 nacc=16", k_fma<16>, 8, 40000, 16, 128.0);
  return 0;
}
