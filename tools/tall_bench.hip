// Attribution harness for the tall-tile kernels: the forward T product as built (blocked stores with
// the nontemporal hint), with plain stores, and with the stores removed (row sums only), the backward dC product and the
// weighted Gram kernel, each launched `reps` times back to back on 10^6 rows (steady power state).  Run it plain for the
// event-timed durations, or under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace`
// (clock = GRBM_GUI_ACTIVE / duration; issue share = MFMA busy / (4 SIMDs x CUs x active cycles)).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I dgp-toolbox_amd/csrc tools/tall_bench.hip \
//         dgp-toolbox_amd/csrc/gemm_f64.hip dgp-toolbox_amd/csrc/gemm_wide.hip dgp-toolbox_amd/csrc/gemm_gram.hip \
//         dgp-toolbox_amd/csrc/gemm_tall.hip dgp-toolbox_amd/csrc/gemm_tallu.hip dgp-toolbox_amd/csrc/gemm_small.hip \
//         dgp-toolbox_amd/csrc/gemm_mid.hip -o tools/tall_bench
//   tools/tall_bench [rows] [reps] [D] [which: bitmask 1 T/NT, 2 T/plain, 4 T/no store, 8 dC, 16 Gram, 32 g row-panel, 64 dC row-panel, 128 Gram two sources, 1024 Ct solve]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <functional>
#include "gemm_tall.h"
#include "gemm_tallu.h"
#include "gemm_dcpanel.h"
#include "dgp_internal.h"
using namespace dgp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static double* dalloc(size_t n) { double* p; CK(hipMalloc(&p, n * 8)); return p; }
static void fill_rand(double* d, size_t n, unsigned seed, double scale = 1.0) {
  std::vector<double> h(1 << 22);
  srand(seed);
  for (auto& x : h) x = scale * ((double)rand() / RAND_MAX - 0.5);
  for (size_t off = 0; off < n; off += h.size()) CK(hipMemcpy(d + off, h.data(), std::min(h.size(), n - off) * 8, hipMemcpyHostToDevice));
}
static void fill_tri_blocks(double* d, long rows, long cols, bool lower, unsigned seed) {   // 256 x 256 triangular blocks
  std::vector<double> h((size_t)rows * cols);
  srand(seed);
  for (long r = 0; r < rows; ++r)
    for (long c = 0; c < cols; ++c) {
      const long rr = r % 256, cc = c % 256;
      const bool keep = lower ? cc <= rr : cc >= rr;
      h[(size_t)r * cols + c] = keep ? ((double)rand() / RAND_MAX - 0.5) : 0.0;
    }
  CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
}
static float time_ms(hipStream_t st, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f(); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const long P = ((argc > 1 ? atol(argv[1]) : 1000000) + 255) / 256 * 256;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const int D = argc > 3 ? atoi(argv[3]) : 8;
  const int which = argc > 4 ? atoi(argv[4]) : 31;
  hipStream_t st; CK(hipStreamCreate(&st));
  int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const long DM = 256L * D;
  double* Ct = dalloc((size_t)P * 256); fill_rand(Ct, (size_t)P * 256, 1);
  double* T = dalloc((size_t)P * DM); fill_rand(T, (size_t)P * DM, 2);
  double* W = dalloc(256 * DM); fill_tri_blocks(W, 256, DM, true, 3);
  double* WT = dalloc(DM * 256); fill_tri_blocks(WT, DM, 256, false, 4);
  double* rs = dalloc((size_t)2 * D * P);
  double* vbar = dalloc((size_t)P * D); fill_rand(vbar, (size_t)P * D, 5);
  double* mbar = dalloc((size_t)P * D); fill_rand(mbar, (size_t)P * D, 6);
  double* u = dalloc(256 * D); fill_rand(u, 256 * D, 7);
  double* Cb = dalloc((size_t)P * 256);
  double* G = dalloc((size_t)D * 65536); CK(hipMemset(G, 0, (size_t)D * 65536 * 8));
  double* ws = dalloc(gemm_gram_ws_bytes(0) / 8);
  const double flopsT = (double)P * D * 256.0 * 257.0;

  TallArgs g{};
  g.A = Ct; g.lda = 256; g.B = W; g.ldb = DM; g.C = T; g.ldc = DM; g.rowsq = rs; g.rowsq_ld = P; g.M = P; g.D = D;
  const long ntile = ((P + 255) / 256) * D;
  const unsigned gridT = (unsigned)std::min<long>(ntile, cus);
#ifdef TL_TIMING
  // (diagnostic build: EVERY launch of the T kernel writes its clock records - the buffer exists before any of them)
  const size_t nts = (size_t)cus * 8 * TL_TS_PER_WG;
  unsigned* ts; CK(hipMalloc(&ts, nts * 4));
  g.ts = ts;
  if ((ntile + cus - 1) / cus + 2 > TL_TS_PER_WG / 48) { printf("too many pairs per workgroup for the clock records\n"); return 1; }
#endif
  printf("rows %ld, D %d, %d CUs, %d launches per variant\n", P, D, cus, reps);
  if (which & 1) {
    const float t = time_ms(st, reps, [&]() { hipLaunchKernelGGL((gemm_tall_kernel<true, true>), dim3(gridT), dim3(512), 0, st, g); });
    printf("T  = Ct W   blocked stores, nontemporal   %8.3f ms  %5.1f TFLOP/s algorithmic\n", t, flopsT / t / 1e9);
  }
  if (which & 2) {
    const float t = time_ms(st, reps, [&]() { hipLaunchKernelGGL((gemm_tall_kernel<true, false>), dim3(gridT), dim3(512), 0, st, g); });
    printf("T  = Ct W   blocked stores, plain         %8.3f ms  %5.1f TFLOP/s algorithmic\n", t, flopsT / t / 1e9);
  }
  if (which & 4) {
    TallArgs n = g; n.C = nullptr;
    const float t = time_ms(st, reps, [&]() { hipLaunchKernelGGL((gemm_tall_kernel<false, false>), dim3(gridT), dim3(512), 0, st, n); });
    printf("T  = Ct W   row sums only (no stores)     %8.3f ms  %5.1f TFLOP/s algorithmic\n", t, flopsT / t / 1e9);
  }
  if (which & 8) {
    TallUArgs a{};
    a.A = T; a.lda = DM; a.B = WT; a.ldb = 256; a.C = Cb; a.ldc = 256; a.s = vbar; a.as_ld = D; a.eadd = Ct; a.rowf = mbar; a.colf = u;
    a.rank = D; a.alpha = 2.0; a.M = P; a.D = D;
    const unsigned gridU = (unsigned)std::min<long>((P + 255) / 256, cus);
    float t;
    if (D == 8) t = time_ms(st, reps, [&]() { hipLaunchKernelGGL(gemm_tallu_kernel<8>, dim3(gridU), dim3(512), 0, st, a); });
    else t = time_ms(st, reps, [&]() { hipLaunchKernelGGL(gemm_tallu_kernel<-1>, dim3(gridU), dim3(512), 0, st, a); });
    printf("dC = [2v.T] WT - (s2v) Ct + mbar uT       %8.3f ms  %5.1f TFLOP/s algorithmic\n", t, flopsT / t / 1e9);
  }
  if (which & 16) {
    GemmArgs a{};
    a.gram_ws = ws; a.gram_ws_bytes = gemm_gram_ws_bytes(0);
    a.A = Ct; a.B = Ct; a.C = G; a.lda = a.ldb = a.ldc = 256; a.M = a.N = 256; a.K = P; a.batch = D; a.sC = 65536;
    a.alpha = 1.0; a.beta = 1; a.tri = TRI_OUT_LOWER; a.triblk = 256; a.splits = 1; a.ascale = vbar; a.as_ld = D; a.ascale_mode = 2;
    if (!gemm_gram_ok(a)) printf("Gram: not eligible\n");
    else {
      const float t = time_ms(st, reps, [&]() { CK(gemm_f64(st, GEMM_TN, a)); });
      printf("G_d = sum_p v_pd c_p c_pT (Gram kernel)   %8.3f ms  %5.1f TFLOP/s algorithmic\n", t, flopsT / t / 1e9);
      // the form the training step launches: du_d = sum_p mbar_pd c_p rides on waves 0 and 1
      double* du = dalloc(256 * D); CK(hipMemset(du, 0, 256 * D * 8));
      a.gram_mb = mbar; a.gram_du = du;
      if (gemm_gram_ok(a)) {
        const float t2 = time_ms(st, reps, [&]() { CK(gemm_f64(st, GEMM_TN, a)); });
        printf("G_d and du_d = sum_p mbar_pd c_p          %8.3f ms  %5.1f TFLOP/s algorithmic\n", t2, flopsT / t2 / 1e9);
      }
    }
  }
  if (which & 1024) {
    // the forward solve  Ct = Kt Linv^T  (upper-triangular B, row sums |c|^2, mean0 = Kt alpha riding on it): what the training step
    // launches through gemm_f64 (dgp_ctx.h: args_Ct) - the wide-tile kernel <1, false, 4>
    double* LinvT = dalloc(256 * 256); fill_tri_blocks(LinvT, 256, 256, false, 9);
    double* alpha = dalloc(256 * D); fill_rand(alpha, 256 * D, 10);
    double* mean0 = dalloc((size_t)P * D);
    double* cnp = dalloc((size_t)2 * P);
    GemmArgs a{};
    a.A = T; a.lda = 256; a.B = LinvT; a.ldb = 256; a.C = Cb; a.ldc = 256; a.M = P; a.N = 256; a.K = 256; a.batch = 1; a.splits = 1;
    a.alpha = 1.0; a.beta = 0; a.tri = TRI_B_UPPER; a.triblk = 256; a.epi = 2; a.rowsq = cnp; a.rowsq_ld = P;
    a.mean_alpha = alpha; a.mean_out = mean0; a.mean_d = D;
    a.cu_count = cus;
    const float t = time_ms(st, reps, [&]() { CK(gemm_f64(st, GEMM_NN, a)); });
    printf("Ct = Kt LinvT (+ |c|^2, mean0)            %8.3f ms  %5.1f TFLOP/s algorithmic; 4.1 GB moved -> %.2f TB/s\n", t, (double)P * 256.0 * 257.0 / t / 1e9,
           ((double)P * 256 * 8 * 2 + (double)P * D * 8) / t / 1e9);
  }
#ifdef TL_TIMING
  if (which & 256) {
    // per-position clocks of the T kernel (diagnostic build: -DTL_TIMING): mean over workgroups (wave 0) and pairs (the first pair
    // of a workgroup left out) of the shader clocks from the previous barrier to the position's wait (work) and inside wait + barrier
    std::vector<unsigned> h(nts);
    for (int variant = 0; variant < 2; ++variant) {
      TallArgs n = g; n.ts = ts; if (variant == 1) n.C = nullptr;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ts, 0, nts * 4));
        if (variant == 0) hipLaunchKernelGGL((gemm_tall_kernel<true, true>), dim3(gridT), dim3(512), 0, st, n);
        else hipLaunchKernelGGL((gemm_tall_kernel<false, false>), dim3(gridT), dim3(512), 0, st, n);
        CK(hipStreamSynchronize(st));
      }
      CK(hipMemcpy(h.data(), ts, nts * 4, hipMemcpyDeviceToHost));
      for (int wv = 0; wv < 8; wv += 7) {
        double work[24] = {0}, wait[24] = {0}; long cnt = 0;
        for (unsigned b = 0; b < gridT; ++b) {
          const unsigned* q = &h[((size_t)b * 8 + wv) * TL_TS_PER_WG];
          // record i (position i of the stream of positions): q[2 i] = clock at its wait, q[2 i + 1] = clock behind the barrier of i - 1
          for (int i = 24; i + 24 < TL_TS_PER_WG / 2 && q[2 * (i + 23)] != 0; i += 24) {
            for (int p = 0; p < 24; ++p) {
              work[p] += (double)(unsigned)(q[2 * (i + p)] - q[2 * (i + p) + 1]);
              wait[p] += (double)(unsigned)(q[2 * (i + p + 1) + 1] - q[2 * (i + p)]);
            }
            ++cnt;
          }
        }
        printf("%s, wave %d: %ld pairs; per position: blocks, clocks of work, clocks in wait + barrier\n", variant == 0 ? "T with stores (nt)" : "T row sums only", wv, cnt);
        double tw = 0, tb = 0;
        for (int p = 0; p < 24; ++p) {
          printf("  p %2d  n %d  work %8.0f  wait %7.0f\n", p, tl_nact(p), work[p] / cnt, wait[p] / cnt);
          tw += work[p] / cnt; tb += wait[p] / cnt;
        }
        printf("  pair: work %.0f + wait %.0f = %.0f clocks\n", tw, tb, tw + tb);
      }
    }
  }
#endif
  if (which & 128) {
    // two-source form (rows from Cb, columns from Ct: one 256 x 256 lower triangle of a rectangular block of a 512 x 512 Gram matrix)
    GemmArgs a{};
    a.gram_ws = ws; a.gram_ws_bytes = gemm_gram_ws_bytes(0);
    a.A = Cb; a.B = Ct; a.C = G; a.lda = a.ldb = a.ldc = 256; a.M = a.N = 256; a.K = P; a.batch = 1;
    a.alpha = 1.0; a.beta = 1; a.tri = TRI_OUT_LOWER; a.triblk = 256; a.splits = 1;
    if (!gemm_gram_ok(a)) printf("Gram, two sources: not eligible\n");
    else {
      const float t = time_ms(st, reps, [&]() { CK(gemm_f64(st, GEMM_TN, a)); });
      printf("G = A^T C lower triangle, two sources     %8.3f ms  %5.1f TFLOP/s algorithmic\n", t, (double)P * 256.0 * 257.0 / t / 1e9);
    }
  }
  if (which & 64) {
    DcPanelArgs a{};
    a.T = T; a.ldt = DM; a.B = WT; a.C = Cb; a.s = vbar; a.as_ld = D; a.eadd = Ct; a.rowf = mbar; a.colf = u; a.rank = D; a.alpha = 2.0; a.P = P; a.D = D;
    const unsigned gridP = (unsigned)std::min<long>((P + 127) / 128, cus);
    float t;
    if (D == 8) t = time_ms(st, reps, [&]() { hipLaunchKernelGGL(dc_panel_kernel<8>, dim3(gridP), dim3(512), 0, st, a); });
    else t = time_ms(st, reps, [&]() { hipLaunchKernelGGL(dc_panel_kernel<-1>, dim3(gridP), dim3(512), 0, st, a); });
    printf("dC (row-panel kernel, row-major T)        %8.3f ms  %5.1f TFLOP/s algorithmic\n", t, flopsT / t / 1e9);
  }
  if (which & 32) {
    double* Z1 = dalloc(256 * 9); fill_rand(Z1, 256 * 9, 8);
    double* X1 = dalloc((size_t)P * 9); fill_rand(X1, (size_t)P * 9, 9);
    double* R1 = dalloc((size_t)P * 9);
    double* GXa = dalloc(256 * 9); CK(hipMemset(GXa, 0, 256 * 9 * 8));
    double* gws = dalloc(512L * 256 * 9);
    double* Lv = dalloc(65536); fill_tri_blocks(Lv, 256, 256, true, 10);
    const float t = time_ms(st, reps, [&]() { CK(g_panel(st, Ct, Lv, T, P, Z1, X1, 9, R1, GXa, gws, cus, 0)); });
    const double fl = (double)P * 256.0 * 257.0 + 4.0 * P * 256.0 * 9.0;
    printf("R1, GX of g = (dC Linv) .* k (row-panel kernel) %8.3f ms  %5.1f TFLOP/s algorithmic; inputs %.2f GB -> %.2f TB/s\n", t, fl / t / 1e9,
           2.0 * P * 2048 / 1e9, 2.0 * P * 2048 / t / 1e9);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
