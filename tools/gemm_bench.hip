// Standalone tuning harness for the fp64 MFMA GEMM engine (includes the kernel template directly).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dgp-toolbox_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#include "gemm_f64.h"
using namespace dgp;

template <bool TA, bool TB, int BM, int BN, int BK, int WR, int WC, bool FAST = false>
double run(const char* name, long M, long N, long K, int splits, int tri, long triblk, double* dA, double* dB, double* dC,
           long lda, long ldb, long ldc, double work_frac) {
  GemmArgs a;
  a.A = dA; a.B = dB; a.C = dC; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.sA = a.sB = a.sC = 0; a.batch = 1; a.splits = splits; a.alpha = 1.0; a.beta = splits > 1 ? 1 : 0; a.tri = tri; a.triblk = triblk;
  long per = (K + splits - 1) / splits; per = ((per + BK - 1) / BK) * BK; a.ksplit = per; a.splits = (int)((K + per - 1) / per);
  const long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  dim3 grid((unsigned)(a.splits == 1 && tiles > 4088 ? 4088 : tiles), (unsigned)a.splits);   // persistent, as the library launches
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, BK, WR, WC, 2, 2, FAST, false>), grid, dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, BK, WR, WC, 2, 2, FAST, false>), grid, dim3(256), 0, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
#ifdef DGP_CLOCK_STAMPS
  {
    const long nwg = (long)grid.x * grid.y;
    long long* st; hipMalloc(&st, nwg * 16); hipMemset(st, 0, nwg * 16);
    GemmArgs b = a; b.emul = reinterpret_cast<const double*>(st); b.C2 = nullptr;
    hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, BK, WR, WC, 2, 2, FAST, false>), grid, dim3(256), 0, 0, b);
    hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, BK, WR, WC, 2, 2, FAST, false>), grid, dim3(256), 0, 0, b);
    hipDeviceSynchronize();
    std::vector<long long> h(2 * nwg); hipMemcpy(h.data(), st, nwg * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (long t = 0; t < nwg; ++t) if (h[2 * t + 1] > 0) ghz.push_back((double)h[2 * t] / ((double)h[2 * t + 1] / 100e6) / 1e9);
    std::sort(ghz.begin(), ghz.end());
    if (!ghz.empty()) printf("      in-kernel shader clock: median %.2f GHz (p10 %.2f, p90 %.2f) over %zu workgroups\n", ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], ghz.size());
    hipFree(st);
  }
#endif
  const double dense = 2.0 * M * N * K;
  printf("%-44s %s %4dx%3dx%2d w%dx%d  %8.3f ms  dense-equiv %6.1f TF  executed(~%.0f%%) %6.1f TF\n", name, FAST ? "FAST" : "gen ", BM, BN, BK, WR, WC, ms,
         dense / ms / 1e9, work_frac * 100, dense * work_frac / ms / 1e9);
  return ms;
}

template <bool SC>
void run_g(const char* name, long P, long Mp, int D, int splits, int tri, double* Ct, double* vbar, double* G, long Ncols = 0) {
  GemmArgs a;
  a.A = Ct; a.B = Ct; a.C = G; a.lda = Mp; a.ldb = Mp; a.ldc = Mp; a.M = Mp; a.N = Ncols ? Ncols : Mp; a.K = P;
  a.sA = 0; a.sB = 0; a.sC = Mp * Mp; a.batch = D; a.splits = splits; a.alpha = 1.0; a.beta = 1; a.tri = tri; a.triblk = Mp;
  if (SC) { a.ascale = vbar; a.as_ld = D; a.ascale_mode = 2; }
  long per = (P + splits - 1) / splits; per = ((per + 15) / 16) * 16; a.ksplit = per; a.splits = (int)((P + per - 1) / per);
  if (splits % 8 == 0) a.splits = ((a.splits + 7) / 8) * 8;
  const long tiles = ((Mp + 127) / 128) * ((a.N + 63) / 64);
  dim3 grid((unsigned)tiles, (unsigned)(a.batch * a.splits));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((gemm_f64_kernel<true, false, 128, 64, 16, 2, 2, 2, 2, true, SC>), grid, dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((gemm_f64_kernel<true, false, 128, 64, 16, 2, 2, 2, 2, true, SC>), grid, dim3(256), 0, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double frac = tri ? 0.75 : 1.0;
  printf("%-56s splits=%4d  %8.3f ms  executed %6.1f TF\n", name, a.splits, ms, 2.0 * P * Mp * a.N * D * frac / ms / 1e9);
}

// the production dC product: C[P x Mp] = [2 vbar .* Ct] * Scat, K = D*Mp wrapping Ct's columns, rank-D epilogue
void run_cbar(long P, long Mp, int D, double* Ct, double* vbar, double* Scat, double* C, double* u, bool rank_epi) {
  GemmArgs a;
  a.A = Ct; a.B = Scat; a.C = C; a.lda = Mp; a.ldb = Mp; a.ldc = Mp; a.M = P; a.N = Mp; a.K = (long)D * Mp;
  a.sA = a.sB = a.sC = 0; a.batch = 1; a.splits = 1; a.ksplit = 0; a.alpha = 2.0; a.beta = 0; a.tri = TRI_NONE; a.triblk = 0;
  a.ascale = vbar; a.as_ld = D; a.a_kblk = Mp; a.ascale_mode = 1;
  if (rank_epi) { a.rowf = vbar; a.colf = u; a.rank = D; }
  const long tiles = (P / 128) * (Mp / 64);
  dim3 grid((unsigned)(tiles > 4088 ? 4088 : tiles), 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((gemm_f64_kernel<false, false, 128, 64, 16, 2, 2, 2, 2, true, true>), grid, dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 128, 64, 16, 2, 2, 2, 2, true, true>), grid, dim3(256), 0, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("NN scaled dC=[2v.*Ct]*Scat (K wraps Ct, rank epilogue %d)      %8.3f ms  executed %6.1f TF\n", (int)rank_epi, ms,
         2.0 * P * Mp * Mp * D / ms / 1e9);
}

int main(int argc, char** argv) {
  // the FAST kernels carry no bounds checks: the harness must only hand them interior shapes (the library's
  // host dispatcher guarantees this; here P is rounded to a multiple of the largest tile height)
  const long P = (((argc > 1 ? atol(argv[1]) : 200000) + 255) / 256) * 256, Mp = 256, D = 8;
  const long DM = D * Mp;
  double *A, *B, *C;
  hipMalloc(&A, P * DM * 8); hipMalloc(&B, (size_t)Mp * DM * 8 + P * Mp * 8); hipMalloc(&C, P * DM * 8);
  std::vector<double> h(1 << 20);
  for (auto& x : h) x = (double)rand() / RAND_MAX - 0.5;
  for (size_t off = 0; off < (size_t)P * DM; off += h.size()) hipMemcpy(A + off, h.data(), std::min(h.size(), (size_t)P * DM - off) * 8, hipMemcpyHostToDevice);
  for (size_t off = 0; off < (size_t)Mp * DM + P * Mp; off += h.size()) hipMemcpy(B + off, h.data(), std::min(h.size(), (size_t)Mp * DM + P * Mp - off) * 8, hipMemcpyHostToDevice);
  printf("P=%ld Mp=%ld D=%ld\n", P, Mp, D);
#define NN_DENSE(BM, BN, BK, WR, WC) run<false, false, BM, BN, BK, WR, WC>("NN  Tt=Ct*Wcat dense", P, DM, Mp, 1, 0, 0, A, B, C, Mp, DM, DM, 1.0)
#define NN_TRI(BM, BN, BK, WR, WC) run<false, false, BM, BN, BK, WR, WC>("NN  Tt=Ct*Wcat tri", P, DM, Mp, 1, TRI_B_LOWER, Mp, A, B, C, Mp, DM, DM, 0.5 + 0.5 * BN / Mp)
#define NT_TRI(BM, BN, BK, WR, WC) run<false, true, BM, BN, BK, WR, WC>("NT  Cbar=sTt*Wcat^T tri (K=D*Mp)", P, Mp, DM, 1, TRI_B_UPPER, Mp, A, B, C, DM, DM, Mp, 0.5 + 0.5 * BN / Mp)
#define TN_GRAM(BM, BN, BK, WR, WC) run<true, false, BM, BN, BK, WR, WC>("TN  dW=Ct^T*sTt (K=P) splits", Mp, DM, P, 48, TRI_OUT_LOWER, Mp, B + Mp * DM, A, C, Mp, DM, DM, 0.75)
#define CB(BM, BN, BK, WR, WC) run<false, false, BM, BN, BK, WR, WC, true>("NN  Cbar=C'*Scat dense K=D*Mp", P, Mp, DM, 1, 0, 0, A, B, C, DM, Mp, Mp, 1.0)
#define TT(BM, BN, BK, WR, WC) run<false, false, BM, BN, BK, WR, WC, true>("NN  Tt=Ct*Wcat tri K=Mp", P, DM, Mp, 1, TRI_B_LOWER, Mp, A, B, C, Mp, DM, DM, 0.5 + 0.5 * BN / Mp)
#define GR(BM, BN, BK, WR, WC) run<true, false, BM, BN, BK, WR, WC, true>("TN  G=Ct^T*Ct (K=P) splits", Mp, DM, P, 48, TRI_OUT_LOWER, Mp, B + Mp * DM, A, C, Mp, DM, DM, 0.75)
  if (argc > 2 && std::string(argv[2]) == "tiles") {   // k-depth / occupancy sweep of the interior kernel
    CB(128, 64, 16, 2, 2);
    TT(128, 64, 16, 2, 2);
    GR(128, 64, 16, 2, 2);
    {
      double* vb; hipMalloc(&vb, P * D * 8); hipMemset(vb, 0, P * D * 8);
      run_cbar(P, Mp, (int)D, B + Mp * DM, vb, B, C, B, false);
      run_cbar(P, Mp, (int)D, B + Mp * DM, vb, B, C, B, true);
    }
    return 0;
  }
  {
    double* vb; hipMalloc(&vb, P * D * 8); hipMemset(vb, 0, P * D * 8);
    double* Ct = B + Mp * DM;   // [P x Mp]
    run_g<true>("G production: batch 8, scaled, lower, grouped", P, Mp, (int)D, 248, TRI_OUT_LOWER, Ct, vb, C);
    run_g<false>("G unscaled,  batch 8, lower, grouped", P, Mp, (int)D, 248, TRI_OUT_LOWER, Ct, vb, C);
    run_g<false>("G unscaled,  batch 8, dense, grouped", P, Mp, (int)D, 248, 0, Ct, vb, C);
    run_g<false>("G unscaled,  batch 8, dense, ungrouped(s=47)", P, Mp, (int)D, 47, 0, Ct, vb, C);
    run_g<false>("G unscaled,  batch 8, dense, grouped s=1000", P, Mp, (int)D, 1000, 0, Ct, vb, C);
    run_g<false>("G unscaled,  batch 1, dense, grouped", P, Mp, 1, 248, 0, Ct, vb, C);
    run_g<true>("G scaled,    batch 8, dense, grouped", P, Mp, (int)D, 248, 0, Ct, vb, C);
    run_g<false>("G unscaled, batch 8, dense N=192 (6 of 8 tiles)", P, Mp, (int)D, 248, 0, Ct, vb, C, 192);
    run_g<false>("G unscaled, batch 8, dense N=128 (4 of 8 tiles)", P, Mp, (int)D, 248, 0, Ct, vb, C, 128);
    run_g<false>("G unscaled, batch 8, dense N=64  (2 of 8 tiles)", P, Mp, (int)D, 248, 0, Ct, vb, C, 64);
  }
  return 0;
}
