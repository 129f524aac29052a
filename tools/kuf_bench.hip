// Where does the Kuf assembly's time go?  Variants of rbf_kuf_kernel<8> at P = 1e6, Mp = 256 (2 GB written):
//   0 as built   1 no exp (stores r2)   2 exp, result summed instead of stored   3 two columns per thread, 16-byte stores,
//   block walks two point rows at a time   4 as 3 with nontemporal stores
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dgp-toolbox_amd/csrc tools/kuf_bench.hip -o tools/kuf_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dgp_internal.h"
using namespace dgp;

template <int V>
__global__ __launch_bounds__(256) void kuf_var(const double* __restrict__ X, long P, const double* __restrict__ Z, const double* __restrict__ var,
                                               const double* __restrict__ ls, int Mp, double* __restrict__ Kt, int ppb, double* sink) {
  constexpr int DIN = 8;
  if constexpr (V <= 2) {
    const int m = threadIdx.x;
    double zs[DIN], il[DIN];
#pragma unroll
    for (int j = 0; j < DIN; ++j) { il[j] = 1.0 / ls[j]; zs[j] = Z[(long)m * DIN + j] * il[j]; }
    const double v0 = var[0];
    const long p0 = (long)blockIdx.x * ppb, p1 = min(P, p0 + ppb);
    double acc = 0.0;
    for (long p = p0; p < p1; ++p) {
      const double* __restrict__ x = X + p * DIN;
      double r2 = 0.0;
#pragma unroll
      for (int j = 0; j < DIN; ++j) { const double d = x[j] * il[j] - zs[j]; r2 += d * d; }
      if constexpr (V == 0) Kt[p * Mp + m] = v0 * exp_nonpos(-0.5 * r2);
      if constexpr (V == 1) Kt[p * Mp + m] = r2;
      if constexpr (V == 2) acc += v0 * exp_nonpos(-0.5 * r2);
    }
    if constexpr (V == 2) if (acc == 1.2345) sink[0] = acc;
  } else {
    // two adjacent columns per thread; waves 0-1 take row p, waves 2-3 row p + 1
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
    const int m = 2 * (threadIdx.x & 127);
    double zs[2][DIN], il[DIN];
#pragma unroll
    for (int j = 0; j < DIN; ++j) { il[j] = 1.0 / ls[j]; zs[0][j] = Z[(long)m * DIN + j] * il[j]; zs[1][j] = Z[(long)(m + 1) * DIN + j] * il[j]; }
    const double v0 = var[0];
    const long p0 = (long)blockIdx.x * ppb, p1 = min(P, p0 + ppb);
    for (long p = p0 + half; p < p1; p += 2) {
      const double* __restrict__ x = X + p * DIN;
      double r0 = 0.0, r1 = 0.0;
#pragma unroll
      for (int j = 0; j < DIN; ++j) {
        const double xs = x[j] * il[j];
        const double d0 = xs - zs[0][j], d1 = xs - zs[1][j];
        r0 += d0 * d0; r1 += d1 * d1;
      }
      d2_t v = {v0 * exp_nonpos(-0.5 * r0), v0 * exp_nonpos(-0.5 * r1)};
      if constexpr (V == 4) __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(Kt + p * Mp + m));
      else *reinterpret_cast<d2_t*>(Kt + p * Mp + m) = v;
    }
  }
}

int main() {
  const long P = 1000064; const int Mp = 256, D = 8;
  double *X, *Z, *var, *ls, *Kt, *sink;
  hipMalloc(&X, P * D * 8); hipMalloc(&Z, Mp * D * 8); hipMalloc(&var, 8); hipMalloc(&ls, D * 8); hipMalloc(&Kt, P * Mp * 8); hipMalloc(&sink, 8);
  std::vector<double> h(P * D);
  for (auto& v : h) v = rand() / (double)RAND_MAX;
  hipMemcpy(X, h.data(), P * D * 8, hipMemcpyHostToDevice);
  hipMemcpy(Z, h.data(), Mp * D * 8, hipMemcpyHostToDevice);
  double one[8] = {1, 1, 1, 1, 1, 1, 1, 1};
  hipMemcpy(var, one, 8, hipMemcpyHostToDevice); hipMemcpy(ls, one, 64, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int v, int ppb) {
    dim3 grid((unsigned)((P + ppb - 1) / ppb));
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      switch (v) {
        case 0: hipLaunchKernelGGL(kuf_var<0>, grid, dim3(256), 0, 0, X, P, Z, var, ls, Mp, Kt, ppb, sink); break;
        case 1: hipLaunchKernelGGL(kuf_var<1>, grid, dim3(256), 0, 0, X, P, Z, var, ls, Mp, Kt, ppb, sink); break;
        case 2: hipLaunchKernelGGL(kuf_var<2>, grid, dim3(256), 0, 0, X, P, Z, var, ls, Mp, Kt, ppb, sink); break;
        case 3: hipLaunchKernelGGL(kuf_var<3>, grid, dim3(256), 0, 0, X, P, Z, var, ls, Mp, Kt, ppb, sink); break;
        case 4: hipLaunchKernelGGL(kuf_var<4>, grid, dim3(256), 0, 0, X, P, Z, var, ls, Mp, Kt, ppb, sink); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("variant %d ppb %3d: %.3f ms  (%.2f TB/s written)\n", v, ppb, best, P * Mp * 8.0 / best / 1e9);
  };
  for (int ppb : {64, 16, 256}) for (int v = 0; v < 5; ++v) run(v, ppb);
  return 0;
}
