// Host side of the wide-tile kernel (gemm_wide.h): eligibility test and launch.
#include "gemm_wide.h"

#include <cstdlib>

namespace dgp {

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool gemm_wide_ok(const GemmArgs& a) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("DGP_WIDE"); enabled = e ? atoi(e) : 1; }
  if (!enabled) return false;
  if (a.c2_only) return false;       // (tried as a mode of its own: 2.00 ms against 1.82 ms on the 128 x 64 engine)
  // one workgroup per CU walks the tiles: fewer than a few rounds of tiles leave CUs idle (small models, the deduplicated
  // first layer of a small shard): those stay on the 128 x 64 engine with its four times finer tiles
  static long min_tiles = -1;
  // (768 until round 3 - three rounds of tiles; measured on the shards of a strongly-scaled run: a 25 008-point first layer (196
  //  tiles, one workgroup on 196 of 256 CUs) already runs Ct / dC faster here than on the engine: N/4 14.74 -> 14.52 ms, N/2 25.83
  //  -> 25.40 ms; a 12 496-point one (98 tiles) does not: 8.62 -> 9.01 ms)
  if (min_tiles < 0) { const char* e = getenv("DGP_WIDE_MIN_TILES"); min_tiles = e ? atol(e) : 192; }
  if ((a.M / WBM) * (a.N / WBN) < min_tiles) return false;
  if (a.M <= 0 || a.M % WBM != 0 || a.N <= 0 || a.N % WBN != 0 || a.K <= 0 || a.K % WBK != 0) return false;
  if (a.batch > 1 || a.splits > 1 || a.beta != 0) return false;
  // second output C2 = C .* emul (dK -> g): only as the interleaved mode 2 (B lower, one k block, plain epilogue), and off by
  // default: that product moves 8 GB per 10^6 rows (A, emul in; C, C2 out) for 0.13 TFLOP, i.e. it is HBM-bound, and the
  // 128 x 64 engine with two workgroups per CU streams it faster (2.06 ms against 2.20 ms; DGP_WIDE_EM2=1 selects mode 2)
  if (a.C2 != nullptr || a.emul != nullptr) {
    static int em2 = -1;
    if (em2 < 0) { const char* e = getenv("DGP_WIDE_EM2"); em2 = e ? atoi(e) : 0; }
    if (!em2 || a.tri != TRI_B_LOWER || a.K != a.triblk || a.alpha != 1.0 || a.rank != 0 || a.eadd != nullptr || a.epi != 0 ||
        a.c_blocked || a.ascale_mode != 0 || a.C2 == nullptr || a.emul == nullptr)
      return false;
  }
  if (a.tri != TRI_NONE && a.tri != TRI_B_UPPER && a.tri != TRI_B_LOWER) return false;
  long kblen = a.K;
  if (a.tri != TRI_NONE) {
    if (a.triblk <= 0 || a.triblk % WBN != 0) return false;
    kblen = a.K > a.triblk ? a.triblk : a.K;
    if (a.K % kblen != 0) return false;
    if (a.K < a.triblk) return false;            // a k range shorter than the triangular block: generic engine
  }
  if (a.ascale_mode != 0) {
    // row scale per k block only, all K columns stored (a_wrap 0), scale blocks = triangular blocks
    if (a.ascale_mode != 1 || a.a_wrap != 0 || a.tri != TRI_B_UPPER || a.a_kblk != kblen || a.K / kblen > W_MAXD) return false;
    if (a.ascale == nullptr) return false;
  }
  if (!al16(a.A) || !al16(a.B) || a.ldb % 2 != 0 || a.lda % 2 != 0) return false;
  if (a.epi != 1) {
    if (a.C == nullptr || !al16(a.C) || a.ldc % 2 != 0) return false;
    if (!al16(a.C2) || !al16(a.emul) || !al16(a.eadd)) return false;
    if ((a.C2 != nullptr) != (a.emul != nullptr)) return false;
    if (a.c_blocked && (a.eadd != nullptr || a.C2 != nullptr)) return false;
    if (a.eadd != nullptr && a.ascale == nullptr) return false;
  }
  if (a.epi != 0 && a.rowsq == nullptr) return false;
  if (a.rank < 0 || a.rank > 32 || (a.rank > 0 && (a.rowf == nullptr || a.colf == nullptr))) return false;
  // 32-bit per-lane byte offsets of the DMA sources
  if ((a.a_blocked ? (long)WBK : a.lda) * 8 * WBM >= (1L << 31) || a.ldb * 8 * WBK >= (1L << 31)) return false;
  return true;
}

// mode 4 (gemm_wide.h): the plain forward product with the row sums (epi 2), one 256-column tile per row panel, K = 256
bool gemm_wide_mean_ok(const GemmArgs& a) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("DGP_WIDE_MEAN"); enabled = e ? atoi(e) : 1; }
  return enabled && a.mean_alpha != nullptr && a.mean_out != nullptr && a.mean_d >= 1 && a.mean_d <= 8 && gemm_wide_ok(a) &&
         a.tri == TRI_B_UPPER && a.N == WBN && a.K == WBN && a.triblk == WBN && a.alpha == 1.0 && a.rank == 0 && a.eadd == nullptr &&
         a.epi == 2 && a.C2 == nullptr && a.ascale_mode == 0 && !a.c_blocked && !a.a_blocked &&
         (reinterpret_cast<uintptr_t>(a.mean_alpha) & 7u) == 0 && (reinterpret_cast<uintptr_t>(a.mean_out) & 7u) == 0;
}

hipError_t gemm_wide(hipStream_t st, const GemmArgs& a) {
  static int grid_env = -1;
  if (grid_env < 0) { const char* e = getenv("DGP_WIDE_GRID"); grid_env = (e && atoi(e) > 0) ? atoi(e) : 0; }
  const int cus = grid_env ? grid_env : gemm_device_cus(a);
  const long tiles = (a.M / WBM) * (a.N / WBN);
  long gmax = gemm_persistent_grid(a, cus); // one persistent workgroup per CU (147 KB of LDS each), less those left to a collective
  // With several 256-column tiles per triangular block the tiles of a row panel cost 1 : 2.8 (diagonal block only /
  // diagonal + dense).  A workgroup walks tiles b, b + grid, ...: with grid a multiple of the tiles per panel it would
  // see one kind only.  8 * 31 workgroups make its column tile cycle through all of them.
  if (a.tri != TRI_NONE && a.triblk > WBN && gmax > 248) gmax = 248;
  const unsigned grid = (unsigned)(tiles < gmax ? tiles : gmax);
  const int dir = a.tri == TRI_B_UPPER ? 1 : (a.tri == TRI_B_LOWER ? 2 : 0);
  const bool sc = a.ascale_mode == 1;
  // epilogue mode (gemm_wide.h): the interleaved block-column forms for the plain products with one k block per tile
  int em = 0;
  const bool plain = dir != 0 && !sc && a.alpha == 1.0 && a.rank == 0 && a.eadd == nullptr && a.K == a.triblk;
  if (plain && a.epi == 2 && a.C2 == nullptr) em = (dir == 1 && gemm_wide_mean_ok(a)) ? 4 : 1;
  else if (plain && a.epi == 0 && a.C2 != nullptr && dir == 2) em = 2;
  else if (plain && a.epi == 1) em = 3;
#define W_LAUNCH(D, S, E) hipLaunchKernelGGL((gemm_wide_kernel<D, S, E>), dim3(grid), dim3(512), 0, st, a)
  if (sc) W_LAUNCH(1, true, 0);
  else if (dir == 0) W_LAUNCH(0, false, 0);
  else if (dir == 1) {
    if (em == 4) W_LAUNCH(1, false, 4);
    else if (em == 1) W_LAUNCH(1, false, 1);
    else W_LAUNCH(1, false, 0);
  } else {
    if (em == 1) W_LAUNCH(2, false, 1);
    else if (em == 2) W_LAUNCH(2, false, 2);
    else if (em == 3) W_LAUNCH(2, false, 3);
    else W_LAUNCH(2, false, 0);
  }
#undef W_LAUNCH
  return hipGetLastError();
}

}  // namespace dgp
