// Does v_mfma_f64_4x4x4_4b keep its 16-cycle issue rate with the operand pattern of the GEMM loop?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define STAMP_BEGIN long long t0 = clock64();
#define STAMP_END long long t1 = clock64(); if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;

// FM x FN fragment tile: FM A regs, 4*FN B regs, 4*FM*FN accumulators, no memory traffic
template <int FM, int FN, int ORDER>
__global__ __launch_bounds__(256) void k_tile(int iters, const double* in, double* sink, long long* cyc) {
  double acc[FM][FN][4];
  double fa[FM], fb[FN][4];
#pragma unroll
  for (int i = 0; i < FM; ++i) fa[i] = in[threadIdx.x + 64 * i];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) fb[j][r] = in[threadIdx.x + 64 * (8 + 4 * j + r)];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0;
  STAMP_BEGIN
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[i], fb[j][r], acc[i][j][r], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < FM; ++i) acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[i], fb[j][r], acc[i][j][r], 0, 0, 0);
    }
    // perturb operands so the loop cannot be hoisted
#pragma unroll
    for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(fa[i]));
  }
  STAMP_END
  double s = 0;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  sink[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int blocks, int iters, int per_iter) {
  double *sink, *in; long long* cyc;
  hipMalloc(&sink, (size_t)blocks * 256 * 8); hipMalloc(&in, 64 * 64 * 8); hipMalloc(&cyc, blocks * 4 * 8);
  hipMemset(in, 0, 64 * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, in, sink, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, in, sink, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> hc(blocks * 4);
  hipMemcpy(hc.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  std::sort(hc.begin(), hc.end());
  const double inst = (double)iters * per_iter;
  printf("%-40s blocks=%4d  %6.1f TFLOP/s  cycles/inst/wave=%6.1f\n", name, blocks, (double)blocks * 4 * inst * 512 / (ms * 1e-3) / 1e12,
         hc[hc.size() / 2] / inst);
}

int main() {
  for (int wps : {1, 2}) {
    const int b = 256 * wps;
    printf("--- %d wave(s)/SIMD\n", wps);
    run("tile 4x2 order j,i,r (GEMM loop)", k_tile<4, 2, 0>, b, 2000, 32);
    run("tile 4x2 order j,r,i", k_tile<4, 2, 1>, b, 2000, 32);
    run("tile 4x4 order j,i,r", k_tile<4, 4, 0>, b, 1000, 64);
    run("tile 2x2 order j,i,r", k_tile<2, 2, 0>, b, 4000, 16);
    run("tile 1x1", k_tile<1, 1, 0>, b, 8000, 4);
  }
  return 0;
}
