// Host-side dispatch of the fp64 MFMA GEMM engine (see gemm_f64.h).
#include "gemm_f64.h"
#include "gemm_wide.h"
#include "chain.h"
#include <cstdlib>
#include <algorithm>

namespace dgp {

template <bool TA, bool TB, int BM, int BN, int BK, int WR, int WC, int VA, int VB, bool FAST = false, bool SCALED = false>
static hipError_t launch(hipStream_t st, GemmArgs a) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.batch <= 0) return hipSuccess;
  if (a.splits > 1) {   // ksplit must be a multiple of BK so that split boundaries coincide with k-tiles
    const bool grouped = a.splits % 8 == 0;
    while ((long)a.batch * a.splits > 65528 && a.splits > 8) a.splits = ((a.splits / 2 + 7) / 8) * 8;   // grid.y limit
    long per = (a.K + a.splits - 1) / a.splits;
    per = ((per + BK - 1) / BK) * BK;
    a.ksplit = per;
    a.splits = (int)((a.K + per - 1) / per);
    if (grouped) a.splits = ((a.splits + 7) / 8) * 8;   // keep whole XCD groups (empty trailing splits exit at once)
  }
  if (a.splits < 1) a.splits = 1;
  long tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  a.active_tiles = 0;
  if (a.tri == TRI_OUT_LOWER && a.triblk > 0 && (a.triblk % BN == 0 || a.N <= a.triblk) && tiles <= 4096) {
    long act = 0;                                        // launch only the tiles on or below the diagonal
    for (long m0 = 0; m0 < a.M; m0 += BM)
      for (long n0 = 0; n0 < a.N; n0 += BN)
        if (a.tri_row0 + m0 + BM - 1 >= n0 % a.triblk) ++act;
    if (act < tiles) { a.active_tiles = (int)act; tiles = act; }
  }
  // persistent grid: at most 8*255 workgroups walk the tiles (a multiple of 8 keeps a workgroup's tiles on its
  // XCD's contiguous run; 255 is odd so that successive tiles of one workgroup cycle through the column tiles).
  // XCD-grouped split-K launches keep one workgroup per (tile, batch, split).
  const bool grouped = a.splits > 1 && a.splits % 8 == 0;
  long gx = tiles;
  static long gmax = 0;
  static long gmin = 0, per = 4;
  if (gmax == 0) {
    const char* e = getenv("DGP_GEMM_GRID"); gmax = e ? atol(e) : 4088; if (gmax < 8) gmax = 8;
    e = getenv("DGP_GEMM_GRID_MIN"); gmin = e ? atol(e) : 1016; if (gmin > gmax) gmin = gmax;
    e = getenv("DGP_GEMM_TILES_PER_WG"); per = e ? atol(e) : 4; if (per < 1) per = 1;
  }
  static int tri_half = -1;
  if (tri_half < 0) { const char* e = getenv("DGP_TRI_HALF"); tri_half = e ? atoi(e) : 1; }
  a.tri_half = tri_half;
  if (!grouped && gx > gmin) {
    // ~`per` tiles per workgroup amortise the launch cost of a workgroup, few enough workgroups leave the
    // balancing of unequal (triangular) tiles to the hardware dispatcher: 8 * odd, between gmin and gmax
    long want = ((tiles / per / 8) | 1) * 8;
    gx = want < gmin ? gmin : (want > gmax ? gmax : want);
  }
  if constexpr (FAST && BM == 128 && BN == 64 && WR == 2) {
    // lower-triangular output: the tiles that need only their last 64 rows go to the half-MFMA instantiation
    if (a.active_tiles > 0 && a.tri_half) {
      long n_half = 0;
      for (long m0 = 0; m0 < a.M; m0 += BM)
        for (long n0 = 0; n0 < a.N; n0 += BN)
          if (a.tri_row0 + m0 + BM - 1 >= n0 % a.triblk && a.tri_row0 + m0 + BM / 2 <= n0 % a.triblk) ++n_half;
      // (a half part of fewer workgroups than the chip has slots would run as an under-filled launch of its own:
      //  small shards and single-output layers keep the one-launch form)
      static long min_half = -1;
      if (min_half < 0) { const char* e = getenv("DGP_HALF_MIN_WG"); min_half = e ? atol(e) : 384; }
      if (n_half > 0 && n_half < tiles && n_half * a.batch * a.splits >= min_half) {
        GemmArgs f = a, h = a;
        f.tri_part = 1; f.active_tiles = (int)(tiles - n_half);
        h.tri_part = 2; h.active_tiles = (int)n_half;
        long gf = grouped ? tiles - n_half : std::min<long>(gx, tiles - n_half), gh = grouped ? n_half : std::min<long>(gx, n_half);
        hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, BK, WR, WC, VA, VB, FAST, SCALED, 0>),
                           dim3((unsigned)gf, (unsigned)(a.batch * a.splits), 1), dim3(256), 0, st, f);
        hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, BK, WR, WC, VA, VB, FAST, SCALED, 1>),
                           dim3((unsigned)gh, (unsigned)(a.batch * a.splits), 1), dim3(256), 0, st, h);
        return hipGetLastError();
      }
    }
  }
  dim3 grid((unsigned)gx, (unsigned)(a.batch * a.splits), 1);
  hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, BK, WR, WC, VA, VB, FAST, SCALED>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

static inline bool vec2_ok(const double* p, long ld, long extent, long bstride, int batch) {
  return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) && (ld % 2 == 0) && (extent % 2 == 0) &&
         (batch <= 1 || bstride % 2 == 0);
}

// generic (predicated) path for any shape
template <bool TA, bool TB>
static hipError_t launch_generic(hipStream_t st, const GemmArgs& a) {
  const bool va = vec2_ok(a.A, a.lda, TA ? a.M : ((a.ascale_mode == 1 && a.a_wrap) ? a.a_kblk : a.K), a.sA, a.batch);
  const bool vb = vec2_ok(a.B, a.ldb, TB ? a.K : a.N, a.sB, a.batch);
  if (a.N > 16) {
    if (va && vb) return launch<TA, TB, 128, 64, 16, 2, 2, 2, 2>(st, a);
    return launch<TA, TB, 128, 64, 16, 2, 2, 1, 1>(st, a);
  }
  if (va) return launch<TA, TB, 128, 16, 16, 4, 1, 2, 1>(st, a);
  return launch<TA, TB, 128, 16, 16, 4, 1, 1, 1>(st, a);
}

template <bool TA, bool TB>
static hipError_t launch_fast(hipStream_t st, const GemmArgs& a) {
  if (a.ascale_mode == 0) return launch<TA, TB, 128, 64, 16, 2, 2, 2, 2, true, false>(st, a);
  if constexpr (!TB) return launch<TA, TB, 128, 64, 16, 2, 2, 2, 2, true, true>(st, a);   // scaled A: NN (mode 1) / TN (mode 2)
  else return hipErrorInvalidValue;
}

// Splits a product into an interior part run by the lean FAST kernel and edge strips run by the generic one.
template <bool TA, bool TB>
static hipError_t dispatch(hipStream_t st, const GemmArgs& a) {
  constexpr long BM = 128, BN = 64, BK = 16;
  const bool va = vec2_ok(a.A, a.lda, TA ? a.M : ((a.ascale_mode == 1 && a.a_wrap) ? a.a_kblk : a.K), a.sA, a.batch);
  const bool vb = vec2_ok(a.B, a.ldb, TB ? a.K : a.N, a.sB, a.batch);
  bool fast = va && vb && a.N > 16 && a.N % BN == 0 && (a.tri == TRI_NONE || (a.triblk % BN == 0 && a.triblk % BK == 0)) &&
              (a.ascale_mode != 1 || (!TA && a.a_kblk % BK == 0 && a.K % a.a_kblk == 0)) && (a.ascale_mode != 2 || TA) &&
              !(TB && a.ascale_mode != 0) &&
              !(a.ascale_mode == 1 && !a.a_wrap && a.tri != TRI_NONE && a.triblk != a.a_kblk);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  // the interior kernel stores 16-byte pieces of C (and reads/writes its epilogue companions the same way)
  if (!(al16(a.C) && a.ldc % 2 == 0 && (a.batch <= 1 || a.sC % 2 == 0) && al16(a.C2) && al16(a.emul) && al16(a.eadd)))
    fast = false;
  const long Mf = (a.M / BM) * BM;
  if (!TA && a.K % BK != 0) fast = false;        // K is the contiguous direction of A: no K tail handling there
  const long Kf = TA ? (a.K / BK) * BK : a.K;
  if (a.a_blocked || a.c_blocked) {   // blocked operands exist for whole-tile interior products only (dgp_ctx.h sizes them so)
    if (!fast || Mf != a.M || TA || a.beta || a.splits > 1 || a.C2 != nullptr || (a.c_blocked && a.eadd != nullptr) ||
        (a.a_blocked && (a.ascale_mode == 1 && a.a_wrap)))
      return hipErrorInvalidValue;
  }
  if (!fast || Mf == 0 || Kf == 0) return launch_generic<TA, TB>(st, a);
  hipError_t e;
  {  // interior rows x aligned K
    GemmArgs f = a;
    f.M = Mf;
    f.K = Kf;
    if ((e = launch_fast<TA, TB>(st, f)) != hipSuccess) return e;
  }
  if (TA && Kf < a.K) {   // K tail of a reduction over points: accumulate on top of the fast part
    GemmArgs t = a;
    t.M = Mf;
    t.A = a.A + Kf * a.lda;
    t.B = a.B + Kf * a.ldb;
    if (a.ascale_mode == 2) t.ascale = a.ascale + Kf * a.as_ld;
    t.K = a.K - Kf;
    t.splits = 1;
    t.beta = 1;
    if ((e = launch_generic<TA, TB>(st, t)) != hipSuccess) return e;
  }
  if (Mf < a.M) {         // remaining rows, full K
    GemmArgs r = a;
    r.M = a.M - Mf;
    r.tri_row0 = a.tri_row0 + Mf;
    r.A = TA ? a.A + Mf : a.A + Mf * a.lda;
    if (a.C) r.C = a.C + Mf * a.ldc;
    if (a.C2) { r.C2 = a.C2 + Mf * a.ldc; r.emul = a.emul + Mf * a.ldc; }
    if (a.rank > 0) r.rowf = a.rowf + Mf * a.rank;
    if (a.eadd) r.eadd = a.eadd + Mf * a.ldc;
    if (a.rowsq) r.rowsq = a.rowsq + Mf;
    if (a.ascale_mode == 1) r.ascale = a.ascale + Mf * a.as_ld;
    if ((e = launch_generic<TA, TB>(st, r)) != hipSuccess) return e;
  }
  return hipSuccess;
}

// CUs the persistent kernels (one workgroup per CU: tall tiles, Gram) leave free.  A workgroup of theirs fills its CU, so
// with one on every CU nothing else is scheduled until the kernel ends - which is fine for a single process (NOTES.md
// par. 10: the L2 sharing of a full grid is worth more than the free CUs) but makes a collective issued beside the
// backward pass wait for the end of a kernel.  dgp_comm_init sets it for its context (GemmArgs::reserve_cus).
int gemm_device_cus(const GemmArgs& a) {
  if (a.cu_count > 0) return a.cu_count;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n;
  return 256;
}
int gemm_persistent_grid(const GemmArgs& a, int cus) {
  if (cus < 64) return cus;
  const int want = a.reserve_cus < 0 ? 0 : a.reserve_cus;
  const int r = want < cus / 4 ? want : cus / 4;
  return cus - r;
}

int gemm_engine_of(GemmOp op, const GemmArgs& args) {
  GemmArgs a = args;
  if (a.splits < 1) a.splits = 1;
  if (a.batch < 1) a.batch = 1;
  if (gemm_small_ok(op, a)) return 5;      // M, N, K <= 64: one workgroup, operands staged once
  if (gemm_mid_ok(op, a)) return 6;        // the chains' Mp x Mp products (K <= 256, few engine tiles): 32 x 32 tiles, k range staged once
  // row-panel products with a (block-)triangular or dense Mp-wide B: the tall-tile kernels, else the wide-tile kernel
  if (op == GEMM_NN && !a.no_wide && gemm_tall_ok(a)) return 2;      // T = Ct * Wcat at Mp = 256: 256 x 128 tiles
  if (op == GEMM_NN && !a.no_wide && dc_panel_ok(a)) return 7;       // dC from a ROW-MAJOR T at Mp = 256: 128-row panels, A straight into registers (gemm_dcpanel.h)
  if (op == GEMM_NN && !a.no_wide && gemm_tallu_ok(a)) return 3;     // dC = [2 vbar .* T] * W^T ... at Mp = 256 from the blocked T, 256 x 128 tiles
  if (op == GEMM_NN && !a.no_wide && gemm_wide_ok(a)) return 1;
  // weighted Gram products over the points (lower triangle, Mp = 256): the single-staging kernel of gemm_gram.h
  if (op == GEMM_TN && !a.no_wide && gemm_gram_ok(a)) return 4;
  return 0;
}

hipError_t gemm_f64(hipStream_t st, GemmOp op, const GemmArgs& args) {
  GemmArgs a = args;
  if (a.splits < 1) a.splits = 1;
  if (a.batch < 1) a.batch = 1;
  if (ChainRec* r = chain_recorder()) {
    // inside a recording (chain.h) only the one-workgroup product can become a step; anything else must not be issued
    // out of order: the recording is marked unusable and its owner falls back to launch by launch
    if (gemm_engine_of(op, a) != 5) { r->failed = true; return hipSuccess; }
  }
  switch (gemm_engine_of(op, a)) {
    case 5: return gemm_small(st, op, a);
    case 6: return gemm_mid(st, op, a);
    case 2: return gemm_tall(st, a);
    case 3: return gemm_tallu(st, a);
    case 7: return dc_panel(st, a);
    case 1: return gemm_wide(st, a);
    case 4: return gemm_gram(st, a);
    default: break;
  }
  switch (op) {
    case GEMM_NN: return dispatch<false, false>(st, a);
    case GEMM_NT: return dispatch<false, true>(st, a);
    case GEMM_TN: return dispatch<true, false>(st, a);
  }
  return hipErrorInvalidValue;
}

}  // namespace dgp
