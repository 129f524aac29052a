// Check + timing harness of the weighted Gram kernel (dgp-toolbox_amd/csrc/gemm_gram.h) against the 128 x 64 engine.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I dgp-toolbox_amd/csrc tools/gram_bench.hip \
//         dgp-toolbox_amd/csrc/gemm_f64.hip dgp-toolbox_amd/csrc/gemm_wide.hip dgp-toolbox_amd/csrc/gemm_gram.hip -o tools/gram_bench
//   tools/gram_bench [P_check] [P_time]
// G_d += sum_p s[p][d] c_p c_p^T (lower triangle, Mp = 256) for D = 1, 8, 16, weighted and unweighted: (1) both kernels on
// P_check points, every element of the lower triangles compared; (2) both timed on P_time points.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <functional>
#include "gemm_wide.h"
using namespace dgp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static double* dalloc(size_t n) { double* p; CK(hipMalloc(&p, n * 8)); return p; }
static void fill_rand(double* d, size_t n, unsigned seed) {
  std::vector<double> h(1 << 22);
  srand(seed);
  for (auto& x : h) x = (double)rand() / RAND_MAX - 0.5;
  for (size_t off = 0; off < n; off += h.size()) CK(hipMemcpy(d + off, h.data(), std::min(h.size(), n - off) * 8, hipMemcpyHostToDevice));
}
static float time_ms(hipStream_t st, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

static double* g_ws = nullptr;
static GemmArgs gram_args(const double* Ct, const double* s, double* G, long P, int D, bool weighted) {
  GemmArgs a{};
  a.gram_ws = g_ws; a.gram_ws_bytes = gemm_gram_ws_bytes(0);
  a.A = Ct; a.B = Ct; a.C = G;
  a.lda = a.ldb = a.ldc = 256;
  a.M = a.N = 256; a.K = P;
  a.batch = D; a.sC = 65536;
  a.alpha = 1.0; a.beta = 1;
  a.tri = TRI_OUT_LOWER; a.triblk = 256;
  a.splits = 1;
  if (weighted) { a.ascale = s; a.as_ld = D; a.ascale_mode = 2; }
  return a;
}

int main(int argc, char** argv) {
  const long Pc = argc > 1 ? atol(argv[1]) : 70000 / 16 * 16, Pt = argc > 2 ? atol(argv[2]) : 1000000;
  hipStream_t st; CK(hipStreamCreate(&st));
  const long Pmax = std::max(Pc, Pt);
  double* Ct = dalloc((size_t)Pmax * 256);
  double* s = dalloc((size_t)Pmax * 16);
  double* G0 = dalloc(16 * 65536), *G1 = dalloc(16 * 65536);
  g_ws = dalloc(gemm_gram_ws_bytes(0) / 8);
  fill_rand(Ct, (size_t)Pmax * 256, 1);
  fill_rand(s, (size_t)Pmax * 16, 2);
  int bad = 0;
  const int Ds[3] = {1, 4, 8};
  for (int w = 1; w >= 0; --w)
    for (int D : Ds) {
      if (!w && D > 1) continue;
      GemmArgs a = gram_args(Ct, s, G1, Pc, D, w);
      if (!gemm_gram_ok(a)) { printf("D=%d weighted=%d: not eligible?\n", D, w); bad = 1; continue; }
      // pre-filled accumulators: both kernels ADD
      fill_rand(G0, (size_t)D * 65536, 7);
      CK(hipMemcpy(G1, G0, (size_t)D * 65536 * 8, hipMemcpyDeviceToDevice));
      GemmArgs r = a; r.C = G0; r.no_wide = 1;
      r.splits = 64; r.ksplit = ((Pc + 63) / 64 + 15) / 16 * 16;
      CK(gemm_f64(st, GEMM_TN, r));
      CK(gemm_f64(st, GEMM_TN, a));
      CK(hipStreamSynchronize(st));
      std::vector<double> h0((size_t)D * 65536), h1((size_t)D * 65536);
      CK(hipMemcpy(h0.data(), G0, h0.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1.data(), G1, h1.size() * 8, hipMemcpyDeviceToHost));
      double mx = 0, md = 0; size_t nb = 0; int shown = 0;
      for (int d = 0; d < D; ++d)
        for (int i = 0; i < 256; ++i)
          for (int j = 0; j <= i; ++j) mx = std::max(mx, std::fabs(h0[(size_t)d * 65536 + i * 256 + j]));
      for (int d = 0; d < D; ++d)
        for (int i = 0; i < 256; ++i)
          for (int j = 0; j <= i; ++j) {
            const size_t x = (size_t)d * 65536 + i * 256 + j;
            const double df = std::fabs(h0[x] - h1[x]);
            if (!(df <= 1e-11 * mx)) { ++nb; if (shown++ < 6) printf("   diff d=%d (%d,%d): gram %.17g old %.17g\n", d, i, j, h1[x], h0[x]); }
            md = std::max(md, df);
          }
      printf("check D=%2d weighted=%d P=%ld: max |diff| / max = %.3g, %zu bad  %s\n", D, w, Pc, md / mx, nb, nb ? "FAIL" : "ok");
      if (nb) bad = 1;
    }
  {  // two sources: Q = A^T C (lower), unweighted
    double* A2 = dalloc((size_t)Pmax * 256);
    fill_rand(A2, (size_t)Pmax * 256, 5);
    GemmArgs a = gram_args(Ct, s, G1, Pc, 1, false);
    a.A = A2;
    if (!gemm_gram_ok(a)) { printf("two-source: not eligible?\n"); bad = 1; }
    else {
      fill_rand(G0, 65536, 7);
      CK(hipMemcpy(G1, G0, 65536 * 8, hipMemcpyDeviceToDevice));
      GemmArgs r = a; r.C = G0; r.no_wide = 1; r.splits = 64;
      CK(gemm_f64(st, GEMM_TN, r));
      CK(gemm_f64(st, GEMM_TN, a));
      CK(hipStreamSynchronize(st));
      std::vector<double> h0(65536), h1(65536);
      CK(hipMemcpy(h0.data(), G0, h0.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1.data(), G1, h1.size() * 8, hipMemcpyDeviceToHost));
      double mx = 0, md = 0; size_t nb = 0; int shown = 0;
      for (int i = 0; i < 256; ++i) for (int j = 0; j <= i; ++j) mx = std::max(mx, std::fabs(h0[i * 256 + j]));
      for (int i = 0; i < 256; ++i)
        for (int j = 0; j <= i; ++j) {
          const double df = std::fabs(h0[i * 256 + j] - h1[i * 256 + j]);
          if (!(df <= 1e-11 * mx)) { ++nb; if (shown++ < 6) printf("   diff (%d,%d): gram %.17g old %.17g\n", i, j, h1[i * 256 + j], h0[i * 256 + j]); }
          md = std::max(md, df);
        }
      printf("check two-source P=%ld: max |diff| / max = %.3g, %zu bad  %s\n", Pc, md / mx, nb, nb ? "FAIL" : "ok");
      if (nb) bad = 1;
      GemmArgs at = gram_args(Ct, s, G1, Pt, 1, false); at.A = A2;
      GemmArgs rt = at; rt.C = G0; rt.no_wide = 1;
      long nsp = std::max(1L, Pt / 2048); rt.splits = (int)nsp;
      const float t_old = time_ms(st, 5, [&] { CK(gemm_f64(st, GEMM_TN, rt)); });
      const float t_new = time_ms(st, 5, [&] { CK(gemm_f64(st, GEMM_TN, at)); });
      const double fl = 2.0 * Pt * 256.0 * 257.0 / 2.0;
      printf("time two-source P=%ld: 128x64 engine %.3f ms (%.1f TF alg)   gram %.3f ms (%.1f TF alg, %.1f TF executed)\n", Pt, t_old, fl / t_old * 1e-9, t_new,
             fl / t_new * 1e-9, 2.0 * Pt * 136 * 256.0 / t_new * 1e-9);
    }
  }
  if (bad) { printf("MISMATCH\n"); if (!getenv("GB_IGNORE")) return 1; }
  for (int w = 1; w >= 0; --w)
    for (int D : Ds) {
      if (!w && D > 1) continue;
      GemmArgs a = gram_args(Ct, s, G1, Pt, D, w);
      GemmArgs r = a; r.C = G0; r.no_wide = 1;
      // the split the library picks for this shape (dgp_ctx.h pick_splits_tiles): ~8 MB of A per chunk
      long nsp = std::max(1L, Pt / 4096); r.splits = (int)nsp; r.ksplit = ((Pt + nsp - 1) / nsp + 15) / 16 * 16;
      const float t_old = time_ms(st, 5, [&] { CK(gemm_f64(st, GEMM_TN, r)); });
      const float t_new = time_ms(st, 5, [&] { CK(gemm_f64(st, GEMM_TN, a)); });
      const double fl = 2.0 * Pt * 256.0 * 257.0 / 2.0 * D;     // algorithmic: the lower triangle incl. diagonal
      printf("time D=%2d weighted=%d P=%ld: 128x64 engine %.3f ms (%.1f TF alg)   gram %.3f ms (%.1f TF alg, %.1f TF executed)\n", D, w, Pt, t_old,
             fl / t_old * 1e-9, t_new, fl / t_new * 1e-9, 2.0 * Pt * 136 * 256.0 * D / t_new * 1e-9);
    }
  return 0;
}
