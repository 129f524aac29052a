// Check + timing harness of the wide-tile kernel (dgp-toolbox_amd/csrc/gemm_wide.h) against the 128 x 64 engine.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I dgp-toolbox_amd/csrc tools/wide_bench.hip \
//         dgp-toolbox_amd/csrc/gemm_f64.hip dgp-toolbox_amd/csrc/gemm_wide.hip -o tools/wide_bench
//   tools/wide_bench [P_check] [P_time]
// For every product shape of the SVGP-layer path: (1) both kernels on P_check rows, every output element compared;
// (2) both kernels timed on P_time rows (the headline configuration has 10^6 rows per hidden layer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
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
// B [K x N] with the block-triangular structure of `tri` (zeros elsewhere)
static void fill_tri(double* d, long K, long N, int tri, long triblk, unsigned seed) {
  std::vector<double> h((size_t)K * N);
  srand(seed);
  for (long k = 0; k < K; ++k)
    for (long n = 0; n < N; ++n) {
      double v = (double)rand() / RAND_MAX - 0.5;
      if (tri == TRI_B_UPPER && (k % triblk) > (n % triblk)) v = 0.0;
      if (tri == TRI_B_LOWER && (k % triblk) < (n % triblk)) v = 0.0;
      h[k * N + n] = v;
    }
  CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
}
static double max_rel_diff(const double* a, const double* b, size_t n) {
  std::vector<double> ha(n), hb_(n);
  CK(hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb_.data(), b, n * 8, hipMemcpyDeviceToHost));
  double md = 0.0, mx = 0.0;
  for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(hb_[i]));
  int shown = 0;
  size_t nbad = 0;
  for (size_t i = 0; i < n; ++i) {
    if (std::isnan(ha[i]) || std::isnan(hb_[i])) { if (shown++ < 6) printf("      NaN at %zu: wide %g old %g\n", i, ha[i], hb_[i]); md = 1e300; continue; }
    const double d = std::fabs(ha[i] - hb_[i]);
    if (d > 1e-12 * mx) { ++nbad; if (shown++ < 6) printf("      diff at %zu: wide %.17g old %.17g\n", i, ha[i], hb_[i]); }
    md = std::max(md, d);
  }
  if (nbad) {
    printf("      %zu of %zu elements differ\n", nbad, n);
    if (getenv("WB_HIST")) {      // blocked layout [panel][col block][128 rows][16]: where do the differences sit?
      size_t hc[16] = {0}, hr[8] = {0}, hb[16] = {0};
      for (size_t i = 0; i < n; ++i) {
        const double d = std::fabs(ha[i] - hb_[i]);
        if (d > 1e-12 * mx) { hc[i % 16]++; hr[(i / 16) % 128 / 16]++; hb[(i / 2048) % 16]++; }
      }
      printf("      by column in block:"); for (int k = 0; k < 16; ++k) printf(" %zu", hc[k]);
      printf("\n      by 16-row group of the panel:"); for (int k = 0; k < 8; ++k) printf(" %zu", hr[k]);
      printf("\n      by column block mod 16:"); for (int k = 0; k < 16; ++k) printf(" %zu", hb[k]);
      printf("\n");
    }
  }
  return md / (mx > 0 ? mx : 1.0);
}
static float time_ms(hipStream_t st, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

struct Case {
  std::string name;
  GemmArgs a;
  size_t c_elems, rs_elems;
  double alg_flops;    // algorithmic: triangular M(M+1) per block
};

int main(int argc, char** argv) {
  const long Pc = ((argc > 1 ? atol(argv[1]) : 4000) + 127) / 128 * 128;
  const long Pt = ((argc > 2 ? atol(argv[2]) : 1000000) + 127) / 128 * 128;
  const int reps = argc > 3 ? atoi(argv[3]) : 3;
  hipStream_t st = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const bool timing = pass == 1;
    const long P = timing ? Pt : Pc;
    if (P <= 0) continue;
    for (long Mp : {256L, 512L}) {
      const int dalt = getenv("WB_D") ? atoi(getenv("WB_D")) : 1;      // the second block count (1..8)
      for (int D : {8, dalt}) {
        if (timing && Mp == 512 && D == 1) continue;
        if (timing && Mp == 512 && P > 250000) { /* keep the big arrays bounded */ }
        const long Pm = (timing && Mp == 512) ? std::min(P, 256000L) : P;
        const long DM = D * Mp;
        double* Kt = dalloc(Pm * Mp); fill_rand(Kt, Pm * Mp, 1);
        double* Ct = dalloc(Pm * Mp); fill_rand(Ct, Pm * Mp, 2);
        double* Tt = dalloc(Pm * DM); fill_rand(Tt, Pm * DM, 3);
        double* vbar = dalloc(Pm * D); fill_rand(vbar, Pm * D, 4);
        double* mbar = dalloc(Pm * D); fill_rand(mbar, Pm * D, 5);
        double* u = dalloc(Mp * D); fill_rand(u, Mp * D, 6);
        double* LinvT = dalloc(Mp * Mp); fill_tri(LinvT, Mp, Mp, TRI_B_UPPER, Mp, 7);
        double* Linv = dalloc(Mp * Mp); fill_tri(Linv, Mp, Mp, TRI_B_LOWER, Mp, 8);
        double* Wcat = dalloc(Mp * DM); fill_tri(Wcat, Mp, DM, TRI_B_LOWER, Mp, 9);
        double* Scat = dalloc(DM * Mp); fill_tri(Scat, DM, Mp, TRI_B_UPPER, Mp, 10);
        const size_t cmax = (size_t)Pm * DM;
        double* C0 = dalloc(cmax); double* C1 = dalloc(cmax);
        double* G0 = dalloc((size_t)Pm * Mp); double* G1 = dalloc((size_t)Pm * Mp);
        const size_t rsn = (size_t)Pm * (Mp / 32) * D;
        double* R0 = dalloc(rsn); double* R1 = dalloc(rsn);
        const double tri1 = (double)Pm * Mp * (Mp + 1.0);

        std::vector<Case> cases;
        {  // Ct = Kt * Linv^T (as an NN product with the transposed copy), |c|^2 partials
          Case c; c.name = "Ct  = Kt * LinvT      (upper, K=Mp, rowsq)";
          GemmArgs a; a.A = Kt; a.B = LinvT; a.lda = Mp; a.ldb = Mp; a.ldc = Mp; a.M = Pm; a.N = Mp; a.K = Mp;
          a.sA = a.sB = a.sC = 0; a.batch = 1; a.splits = 1; a.ksplit = 0; a.alpha = 1.0; a.beta = 0;
          a.tri = TRI_B_UPPER; a.triblk = Mp; a.epi = 2; a.rowsq_ld = Pm;
          c.a = a; c.c_elems = (size_t)Pm * Mp; c.rs_elems = (size_t)Pm * (Mp / 32); c.alg_flops = tri1; cases.push_back(c);
        }
        {  // Tt = Ct * Wcat, blocked output, |t_d|^2 partials
          Case c; c.name = "Tt  = Ct * Wcat       (lower, K=Mp, N=D*Mp, blocked C, rowsq)";
          GemmArgs a; a.A = Ct; a.B = Wcat; a.lda = Mp; a.ldb = DM; a.ldc = DM; a.M = Pm; a.N = DM; a.K = Mp;
          a.sA = a.sB = a.sC = 0; a.batch = 1; a.splits = 1; a.ksplit = 0; a.alpha = 1.0; a.beta = 0;
          a.tri = TRI_B_LOWER; a.triblk = Mp; a.epi = 2; a.rowsq_ld = Pm; a.c_blocked = 1;
          c.a = a; c.c_elems = (size_t)Pm * DM; c.rs_elems = (size_t)Pm * (Mp / 32) * D; c.alg_flops = tri1 * D; cases.push_back(c);
        }
        {  // dC = [2 vbar .* Tt] * WTcat - (sum 2 vbar) Ct + mbar u^T
          Case c; c.name = "dCt = [2v.*Tt] * WTcat (upper, K=D*Mp, blocked scaled A, eadd, rank)";
          GemmArgs a; a.A = Tt; a.B = Scat; a.lda = DM; a.ldb = Mp; a.ldc = Mp; a.M = Pm; a.N = Mp; a.K = DM;
          a.sA = a.sB = a.sC = 0; a.batch = 1; a.splits = 1; a.ksplit = 0; a.alpha = 2.0; a.beta = 0;
          a.tri = TRI_B_UPPER; a.triblk = Mp; a.ascale = vbar; a.as_ld = D; a.a_kblk = Mp; a.ascale_mode = 1; a.a_wrap = 0;
          a.a_blocked = 1; a.eadd = Ct; a.eadd_nsc = D; a.rowf = mbar; a.colf = u; a.rank = D;
          if (getenv("WB_NORANK")) { a.rank = 0; a.rowf = a.colf = nullptr; }
          if (getenv("WB_NOEADD")) { a.eadd = nullptr; a.eadd_nsc = 0; }
          c.a = a; c.c_elems = (size_t)Pm * Mp; c.rs_elems = 0; c.alg_flops = tri1 * D; cases.push_back(c);
        }
        {  // dK = dC * Linv, g = dK .* k
          Case c; c.name = "dKt = dCt * Linv      (lower, K=Mp, second output g)";
          GemmArgs a; a.A = Ct; a.B = Linv; a.lda = Mp; a.ldb = Mp; a.ldc = Mp; a.M = Pm; a.N = Mp; a.K = Mp;
          a.sA = a.sB = a.sC = 0; a.batch = 1; a.splits = 1; a.ksplit = 0; a.alpha = 1.0; a.beta = 0;
          a.tri = TRI_B_LOWER; a.triblk = Mp; a.emul = Kt;
          c.a = a; c.c_elems = (size_t)Pm * Mp; c.rs_elems = 0; c.alg_flops = tri1; cases.push_back(c);
        }
        if (!timing || getenv("WB_DENSE")) {  // dense product (no hint)
          Case c; c.name = "dense NN K=D*Mp";
          GemmArgs a; a.A = Tt; a.B = Scat; a.lda = DM; a.ldb = Mp; a.ldc = Mp; a.M = Pm; a.N = Mp; a.K = DM;
          a.sA = a.sB = a.sC = 0; a.batch = 1; a.splits = 1; a.ksplit = 0; a.alpha = 1.0; a.beta = 0; a.tri = TRI_NONE; a.triblk = 0;
          c.a = a; c.c_elems = (size_t)Pm * Mp; c.rs_elems = 0; c.alg_flops = 2.0 * Pm * Mp * DM; cases.push_back(c);
        }
        printf("---- %s: P=%ld Mp=%ld D=%d\n", timing ? "TIMING" : "CHECK", Pm, Mp, D);
        for (auto& c : cases) {
          GemmArgs w = c.a, o = c.a;
          w.C = C1; o.C = C0; w.rowsq = R1; o.rowsq = R0; o.no_wide = 1;
          if (c.a.emul) { w.C2 = G1; o.C2 = G0; }
          // the old engine's row-norm planes: Mp/32 per 256 columns... it writes (tile_col*2 + wave_col) with 64-column
          // tiles, the wide kernel (tile_col*2 + wave_col) with 256-column tiles: compare the plane SUMS per (row, block)
          if (!gemm_wide_ok(w)) { printf("  %-70s NOT eligible for the wide kernel\n", c.name.c_str()); continue; }
          if (!timing) {
            CK(hipMemset(C0, 0, cmax * 8)); CK(hipMemset(C1, 0xff, cmax * 8));
            CK(hipMemset(R0, 0, rsn * 8)); CK(hipMemset(R1, 0, rsn * 8));
            CK(gemm_f64(st, GEMM_NN, o));
            CK(gemm_f64(st, GEMM_NN, w));      // (dispatcher: the tall-tile kernel takes the T product when eligible)
            CK(hipDeviceSynchronize());
            double dc = max_rel_diff(C1, C0, c.c_elems), dg = 0.0, dr = 0.0;
            if (c.a.emul) dg = max_rel_diff(G1, G0, c.c_elems);
            if (c.a.epi) {
              // sum the planes per (row, 256-column block) on the host
              const long nblk = c.a.N / Mp, po = Mp / 32, pw = Mp / 128;
              std::vector<double> h0(c.rs_elems), h1((size_t)Pm * pw * nblk);
              CK(hipMemcpy(h0.data(), R0, h0.size() * 8, hipMemcpyDeviceToHost));
              CK(hipMemcpy(h1.data(), R1, h1.size() * 8, hipMemcpyDeviceToHost));
              double md = 0, mx = 0;
              for (long b = 0; b < nblk; ++b)
                for (long r = 0; r < Pm; ++r) {
                  double s0 = 0, s1 = 0;
                  for (long q = 0; q < po; ++q) s0 += h0[(b * po + q) * Pm + r];
                  for (long q = 0; q < pw; ++q) s1 += h1[(b * pw + q) * Pm + r];
                  md = std::max(md, std::fabs(s0 - s1)); mx = std::max(mx, std::fabs(s0));
                }
              dr = md / (mx > 0 ? mx : 1);
            }
            printf("  %-70s max rel diff C %.2e  g %.2e  rowsq %.2e  %s\n", c.name.c_str(), dc, dg, dr,
                   (dc < 1e-12 && dg < 1e-12 && dr < 1e-12) ? "OK" : "MISMATCH");
          } else {
            const float t_old = time_ms(st, reps, [&]() { CK(gemm_f64(st, GEMM_NN, o)); });
            const float t_new = time_ms(st, reps, [&]() { CK(gemm_f64(st, GEMM_NN, w)); });
            printf("  %-70s 128x64 engine %8.3f ms (%5.1f TF alg)   wide %8.3f ms (%5.1f TF alg)   x%.2f\n", c.name.c_str(), t_old,
                   c.alg_flops / t_old / 1e9, t_new, c.alg_flops / t_new / 1e9, t_old / t_new);
          }
        }
        for (double* p : {Kt, Ct, Tt, vbar, mbar, u, LinvT, Linv, Wcat, Scat, C0, C1, G0, G1, R0, R1}) CK(hipFree(p));
      }
    }
  }
  return 0;
}
