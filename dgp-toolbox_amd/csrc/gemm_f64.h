// fp64 MFMA GEMM engine for gfx950 (v_mfma_f64_4x4x4_4b_f64), LDS-staged, 4 waves / workgroup.
//
// Every dense contraction of the SVGP-layer path runs through this one kernel family:
//   forward   Ct  = Kt  * Linv^T        (reference: tf.linalg.triangular_solve, layers.py:245-247)
//             Tt  = Ct  * Wcat          (reference: tf.matmul(SK, A_tiled),      layers.py:263)
//   backward  dCt = sTt * Wcat^T,  dKt = dCt * Linv, and the reductions over points
//             dWcat = Ct^T * sTt,  Q = dKt^T * Ct,  du = Ct^T * dm    (reference: tf.GradientTape)
//   plus the M x M x M products of the small-matrix chain (Cholesky backward, natural gradient).
//
// Layout rules (all row-major, leading dimension in elements):
//   C[M x N] (+)= alpha * op(A)[M x K] * op(B)[K x N]
//   TA = false: A stored [M][K] (k contiguous);  TA = true: A stored [K][M] (m contiguous)
//   TB = false: B stored [K][N] (n contiguous);  TB = true: B stored [N][K] (k contiguous)
// LDS images keep the global orientation of each tile (so global loads and LDS stores are contiguous).  Fragment
// reads are 16-byte (ds_read_b128) wherever the layout allows: tools/lds_bench.hip measures ~210 B/clk/CU for
// conflict-free b128 reads against ~120 for b64 on gfx950, and the engine was LDS-bound before (96 % LDS busy,
// 21 % of it bank conflicts: rocprofv3 SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT).  Measured conflict-free pitches:
//   k-contiguous image (A not transposed, B transposed): BK+4 doubles, a lane reads its two k values as one b128
//     (pitch BK+2 is 2-way conflicted for both b64 and b128);
//   n-contiguous B image: BN+8 doubles, natural column order: lane c feeds column 4 (c mod 4) + e of its 16-column
//     block to the e-th of the four MFMAs, i.e. 32 contiguous bytes (the MFMA then yields, per lane, four
//     CONSECUTIVE output columns, which the epilogue stores as one 32-byte piece);
//   m-contiguous A image (A transposed): BM+8 doubles, b64 reads of consecutive rows.
//   (In both, the lane groups lk and lk+1 of a 32-lane half read image rows two apart: a pitch of 8 mod 16
//   doubles puts them on opposite halves of the 64 banks; BN+16 / BM+16 were 2-way conflicted.)
//
// Matrix instruction: v_mfma_f64_4x4x4_4b_f64 issues at 16-18 cycles (72-74 TFLOP/s chip-wide, 94 % of
// the 78.6 TFLOP/s fp64 spec) while v_mfma_f64_16x16x4_f64 sustains only ~100 cycles per instruction
// (47 TFLOP/s) on gfx950 — measured by tools/mfma_bench.hip.  Lane maps, probed by tools/mfma_probe.hip:
// lane l supplies A[c = l&15][k = l>>4] and B[k = l>>4][c = l&15] (the 16x16x4 operand layout) and
// receives D[4*(c>>2) + (l>>4)][c], i.e. only the four diagonal 4x4 blocks of the 16x16 product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace dgp {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

enum TriMode : int {
  TRI_NONE = 0,
  TRI_B_UPPER = 1,    // op(B)[k][n] == 0 unless (k mod triblk) <= (n mod triblk)
  TRI_B_LOWER = 2,    // op(B)[k][n] == 0 unless (k mod triblk) >= (n mod triblk)
  TRI_OUT_LOWER = 3,  // only C[m][n] with m >= (n mod triblk) is needed (tiles above are skipped)
};

struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  long lda, ldb, ldc;
  long M, N, K;
  long sA, sB, sC;   // batch strides (elements)
  int batch;
  int splits;        // split-K factor (>1 => atomic accumulation into C, C must be pre-initialised)
  long ksplit;       // K range per split (multiple of BK)
  double alpha;
  int beta;          // 0: overwrite, 1: accumulate (non-atomic; ignored when splits > 1)
  int tri;
  long triblk;
  // TRI_OUT_LOWER launches: number of output tiles that survive the skip when the grid is COMPACT (only those
  // tiles are launched: early-exit workgroups were measured to cost as much as running ones); 0 = full grid
  int active_tiles = 0;
  // Blocked layout of a point-major matrix that only this engine produces and consumes (the stored t_d): panels of BM
  // rows, inside a panel blocks of BK columns, each block BM x BK row-major (pitch BK) - the A tile of a k-tile and the
  // 16-column blocks of an output tile are then single contiguous 16 KB pieces instead of 128 row segments 16 KB apart.
  // lda / ldc keep meaning "columns of the whole matrix".  FAST kernels, non-transposed A / vector epilogue only.
  int a_blocked = 0, c_blocked = 0;
  int tri_half = 1;    // half-MFMA k-tiles on the diagonal block of a triangular B (0 = off, tuning)
  int tri_part = 0;    // TRI_OUT_LOWER with a compact grid: 0 = every active tile; 1 = only tiles that need all their
                       // rows; 2 = only tiles whose first BM/2 rows lie strictly above the diagonal (kernel ROWSEL = 1
                       // runs those with half of the MFMAs).  The host launches parts 1 and 2 back to back.
  long tri_row0 = 0;   // absolute row of C's first row (set when the host launches a row strip of a larger product)
  // optional row scaling of the physical A tile (fused elementwise work, no extra HBM pass):
  //   ascale_mode 1 (A not transposed): A[m][k] = ascale[m*as_ld + k / a_kblk] * Aphys[m][k % a_kblk]
  //                                     (K = nblk * a_kblk re-reads the same physical columns per block)
  //                                     a_wrap = 2: as 1, block index innermost on the interior path (see there)
  //                                     a_wrap = 0: A is stored with all K columns (no re-reading): A[m][k] =
  //                                     ascale[m*as_ld + k / a_kblk] * Aphys[m][k]; with a triangular hint the
  //                                     host passes triblk == a_kblk (scale blocks = triangular blocks)
  //   ascale_mode 2 (A transposed):     A[k][m] = ascale[k*as_ld + batch] * Aphys[k][m]
  const double* ascale = nullptr;
  long as_ld = 0, a_kblk = 1;
  int ascale_mode = 0;
  int a_wrap = 1;
  // epilogue: 0 store C; 1 only row sums of squares; 2 both.  rowsq[(tile_col*WC + wave_col)*rowsq_ld + row]
  int epi = 0;
  double* rowsq = nullptr;
  long rowsq_ld = 0;
  // optional rank-R update in the epilogue: C += rowf[M x R] * colf[N x R]^T  (fuses dC += mbar u^T)
  const double* rowf = nullptr;
  const double* colf = nullptr;
  int rank = 0;
  // optional epilogue term (ascale_mode 1): C[m][n] -= alpha * (sum_{q < eadd_nsc} ascale[m*as_ld + q]) * eadd[m][n]
  // (same leading dimension as C; fuses the "- c" of  dC = sum_d 2 vbar_d (W_d t_d - c))
  const double* eadd = nullptr;
  int eadd_nsc = 0;
  // optional second output C2 = (alpha * A B) .* E  (same shape / leading dimension as C; used for g = dK .* K)
  const double* emul = nullptr;
  double* C2 = nullptr;
  int c2_only = 0;     // 1: only C2 = C .* emul is stored, C itself is not (overwrite semantics only: beta = 0, splits = 1)
  double* gram_ws = nullptr;   // scratch for gemm_gram.h's partial triangles (gram_ws_bytes >= gemm_gram_ws_bytes())
  long gram_ws_bytes = 0;
  // weighted Gram launches only (gemm_gram.h, form DU): also gram_du[m][d] += sum_k gram_mb[k][d] B[k][m]  (batch = d; the
  // adjoint of mean = c^T u_d riding on the launch that streams Ct for G_d).  Ignored - and the caller must issue that product
  // itself - when the launch does not go to the Gram kernel (gemm_engine_of(...) != 4).
  const double* gram_mb = nullptr;
  double* gram_du = nullptr;
  // the forward product Ct = Kt LinvT on the wide-tile kernel only (gemm_wide.h, mode 4): also mean_out[m][d] = sum_k A[m][k]
  // mean_alpha[k][d] (d < mean_d <= 8; alpha = LinvT u, so that this is mean0 = Ct u).  Ignored - the caller issues that product
  // itself - unless gemm_wide_mean_ok(a).
  const double* mean_alpha = nullptr;
  double* mean_out = nullptr;
  int mean_d = 0;
  int no_wide = 0;     // 1: keep this product on the 128 x 64 engine even where the wide-tile kernel (gemm_wide.h) applies
  // the device this launch is for, as its context knows it (0: ask the runtime for the current device's CU count), and the
  // CUs a persistent one-workgroup-per-CU kernel (wide-tile, tall-tile, Gram) leaves free for a collective that runs beside
  // it (set per context by dgp_comm_init: state of the context, not of the process)
  int cu_count = 0, reserve_cus = 0;
};

template <bool TA, bool TB, int BM, int BN, int BK, int WR, int WC, int VA, int VB>
struct GemmCfg {
  static constexpr int WM = BM / WR, WN = BN / WC;
  static constexpr int FM = WM / 16, FN = WN / 16;
  static constexpr int LDA_S = TA ? (BM + 8) : (BK + 4);
  static constexpr int LDB_S = TB ? (BK + 4) : (BN + 8);
  static constexpr int A_ROWS = TA ? BK : BM, A_COLS = TA ? BM : BK;
  static constexpr int B_ROWS = TB ? BN : BK, B_COLS = TB ? BK : BN;
  static constexpr int AS_SZ = A_ROWS * LDA_S, BS_SZ = B_ROWS * LDB_S;
  static constexpr int A_TPR = A_COLS / VA, B_TPR = B_COLS / VB;          // threads per tile row
  static constexpr int A_RPP = 256 / A_TPR, B_RPP = 256 / B_TPR;          // rows per pass
  static constexpr int A_PASS = (A_ROWS + A_RPP - 1) / A_RPP, B_PASS = (B_ROWS + B_RPP - 1) / B_RPP;
  static_assert(WR * WC == 4, "4 waves per workgroup");
  static_assert(WM % 16 == 0 && WN % 16 == 0 && BK % 16 == 0, "MFMA tile granularity (two 8-deep k steps per k-tile)");
  static_assert(256 % A_TPR == 0 && 256 % B_TPR == 0, "loader shape");
};

// global -> registers for one operand tile (rows x contiguous cols), zero-filled outside [rmax, cmax)
template <int ROWS, int COLS, int V, int TPR, int RPP, int PASS>
__device__ __forceinline__ void tile_load(double (&reg)[PASS][V], const double* __restrict__ base, long ld,
                                          long r0, long c0, long rmax, long cmax, int tid,
                                          const double* __restrict__ sc = nullptr, long sc_ld = 0, long sc_col = 0) {
  const int tr = tid / TPR, tc = (tid % TPR) * V;
#pragma unroll
  for (int p = 0; p < PASS; ++p) {
    const int r = p * RPP + tr;
    const long gr = r0 + r, gc = c0 + tc;
    const bool ok = (r < ROWS) && (gr < rmax) && (gc < cmax);
    double f = 1.0;
    if (sc != nullptr) f = ok ? sc[gr * sc_ld + sc_col] : 0.0;
    if constexpr (V == 2) {
      d2_t v = {0.0, 0.0};
      if (ok) v = *reinterpret_cast<const d2_t*>(base + gr * ld + gc);
      reg[p][0] = v[0] * f;
      reg[p][1] = v[1] * f;
    } else {
      reg[p][0] = (ok ? base[gr * ld + gc] : 0.0) * f;
    }
  }
}

template <int ROWS, int COLS, int V, int TPR, int RPP, int PASS, int LDS_LD>
__device__ __forceinline__ void tile_store(const double (&reg)[PASS][V], double* __restrict__ s, int tid) {
  const int tr = tid / TPR, tc = (tid % TPR) * V;
#pragma unroll
  for (int p = 0; p < PASS; ++p) {
    const int r = p * RPP + tr;
    if (r < ROWS) {
      if constexpr (V == 2) {
        d2_t v = {reg[p][0], reg[p][1]};
        *reinterpret_cast<d2_t*>(s + r * LDS_LD + tc) = v;
      } else {
        s[r * LDS_LD + tc] = reg[p][0];
      }
    }
  }
}

// FAST: the host guarantees that every tile of the launch is interior (M % BM == 0, N % BN == 0, every
// k-range a multiple of BK, 16-byte aligned operands): the loader then keeps one running pointer per operand,
// issues unconditional 16-byte loads and carries no predicates — the generic loader's ~330 non-MFMA
// instructions per k-tile made the loop issue-bound (two waves share a SIMD's issue port with the MFMAs).
// SCALED (FAST only): A-operand row scaling of GemmArgs::ascale_mode.
template <bool TA, bool TB, int BM, int BN, int BK, int WR, int WC, int VA, int VB, bool FAST = false, bool SCALED = false,
          int ROWSEL = 0 /* 1: only the half-needed tiles of a lower-triangular output, see tri_part */>
#ifndef DGP_WAVES_PER_EU   // two workgroups per CU (LDS allows no more): cap the register budget at 256 per lane
#define DGP_WAVES_PER_EU 2
#endif
#if DGP_WAVES_PER_EU > 0
__attribute__((amdgpu_waves_per_eu(DGP_WAVES_PER_EU, DGP_WAVES_PER_EU)))
#endif
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgs g) {
  using Cfg = GemmCfg<TA, TB, BM, BN, BK, WR, WC, VA, VB>;
  constexpr int FM = Cfg::FM, FN = Cfg::FN;
  __shared__ __attribute__((aligned(16))) double smem[2 * (Cfg::AS_SZ + Cfg::BS_SZ)];

  // ---- tile coordinates (32-bit index arithmetic: a 64-bit divide is hundreds of instructions per workgroup) ----
  const unsigned tiles_n = (unsigned)((g.N + BN - 1) / BN), tiles_m = (unsigned)((g.M + BM - 1) / BM);
  const unsigned ntile = g.active_tiles > 0 ? (unsigned)g.active_tiles : tiles_n * tiles_m;
  const bool grouped = g.splits > 1 && g.splits % 8 == 0;
  // PERSISTENT over tiles: the host may launch fewer workgroups (gridDim.x) than tiles; each workgroup then walks
  // tiles lin = blockIdx.x, + gridDim.x, ...  Launching a workgroup of this kernel costs ~10 us of slot time
  // (measured: early-exit workgroups were as expensive as running ones), which dominated the short-K products.
  // Every wave of the workgroup takes the same trip count, so the barriers inside stay uniform.
  const unsigned nlin = grouped ? gridDim.x : ntile;
#ifdef DGP_CLOCK_STAMPS   // tuning only: shader-clock cycles and 100 MHz ticks spent by this workgroup
  const long long stamp_c0 = clock64(), stamp_r0 = wall_clock64();
#endif
  for (unsigned lin = blockIdx.x; lin < nlin; lin += gridDim.x) {
  unsigned bid, bz, sp;
  if (grouped) {
    // Reduction over K split into chunks.  All (tile, batch) members that stream the SAME chunk of rows are
    // placed on ONE XCD, back to back, so that the chunk is fetched from HBM once and re-read from that XCD's
    // L2 (blocks are dealt round-robin over the 8 XCDs in launch order, x fastest: placement is a speed
    // matter only).  Without this the Gram kernels had a 0 % L2 hit rate and were HBM-bound.
    const unsigned members = ntile * (unsigned)g.batch;
    const unsigned L = blockIdx.y * gridDim.x + lin;
    const unsigned xcd = L & 7u, j = L >> 3;
    const unsigned grp = j / members, mem = j - grp * members;
    sp = grp * 8u + xcd;
    bz = mem / ntile;
    bid = mem - bz * ntile;
  } else {
    // one XCD's L2 sees a contiguous run of tiles
    bid = lin;
    const unsigned q = ntile / 8u, r = ntile % 8u, xcd = bid % 8u, idx = bid / 8u;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for any ntile
    const unsigned z = blockIdx.y;
    // batch fastest: the batches of one K-split read the same rows of the streamed operands back to back
    bz = z % (unsigned)g.batch;
    sp = z / (unsigned)g.batch;
  }
  unsigned tm_u, tn_u;
  if (g.active_tiles > 0) {
    // compact grid: bid enumerates only the tiles with (m0 + BM - 1) >= (n0 mod triblk); find the bid-th one
    unsigned cnt = 0;
    tm_u = tn_u = 0;
    for (unsigned t = 0; t < tiles_n * tiles_m; ++t) {
      const unsigned a = t / tiles_n, b = t - a * tiles_n;
      bool act = ((unsigned)g.tri_row0 + a * BM + BM - 1) >= ((b * BN) % (unsigned)g.triblk);
      if (g.tri_part != 0) {
        const bool half = ((unsigned)g.tri_row0 + a * BM + BM / 2) <= ((b * BN) % (unsigned)g.triblk);
        act = act && (half == (g.tri_part == 2));
      }
      if (act) { if (cnt == bid) { tm_u = a; tn_u = b; } ++cnt; }
    }
  } else {
    tm_u = bid / tiles_n;
    tn_u = bid - tm_u * tiles_n;   // column tiles fastest: neighbours share A rows
  }
  const long tm = tm_u, tn = tn_u;
  const long m0 = tm * BM, n0 = tn * BN;

  const long nloc0 = (g.tri != TRI_NONE && g.triblk > 0) ? (long)((unsigned)n0 % (unsigned)g.triblk) : n0;
  const bool tri_ok = (g.tri != TRI_NONE) && (g.triblk % BN == 0 || g.N <= g.triblk);
  if (g.tri == TRI_OUT_LOWER && tri_ok && (g.tri_row0 + m0 + BM - 1) < nloc0) continue;   // tile strictly above the diagonal

  const double* __restrict__ A = g.A + (long)bz * g.sA;
  const double* __restrict__ B = g.B + (long)bz * g.sB;
  double* __restrict__ C = g.C + (long)bz * g.sC;

  // ---- K ranges: `nkb` blocks of length kblen, inside each block k in [klo, khi) ----
  long kblen = g.K, nkb = 1, klo = 0, khi = g.K;
  if ((g.tri == TRI_B_UPPER || g.tri == TRI_B_LOWER) && tri_ok) {
    kblen = (g.K > g.triblk) ? g.triblk : g.K;
    nkb = (long)((unsigned)g.K / (unsigned)kblen);
    if (g.tri == TRI_B_UPPER) { khi = nloc0 + BN; if (khi > kblen) khi = kblen; }
    else { klo = (nloc0 / BK) * BK; khi = kblen; }
  } else if (g.splits > 1) {
    klo = (long)sp * g.ksplit;
    khi = klo + g.ksplit; if (khi > g.K) khi = g.K;
    if (klo >= khi) continue;
  }
  const long ktiles_per_blk = (khi - klo + BK - 1) / BK;
  const long ktiles = ktiles_per_blk * nkb;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it scalar
  const int wr = wave / WC, wc = wave % WC;
  const int li = lane & 15, lk = lane >> 4;

  d4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};

  double ra[Cfg::A_PASS][VA], rb[Cfg::B_PASS][VB];

  // k-tile cursor advanced incrementally: no integer divisions in the loop (a 64-bit divide is hundreds of
  // instructions on the GPU and would sit between the MFMA blocks).  kb = triangular block, kk = tile in block,
  // ablk / aoff = block and offset of A's wrapped K (ascale_mode 1).
  long cur_kb = 0, cur_kk = 0, cur_ablk = 0, cur_aoff = 0;
  auto gload = [&]() {                                     // loads the tile at the cursor, then advances it
    const long k0 = cur_kb * kblen + klo + cur_kk * BK, kend = cur_kb * kblen + khi;
    if constexpr (TA) {
      if (g.ascale_mode == 2)
        tile_load<Cfg::A_ROWS, Cfg::A_COLS, VA, Cfg::A_TPR, Cfg::A_RPP, Cfg::A_PASS>(ra, A, g.lda, k0, m0, kend, g.M, tid,
                                                                                      g.ascale, g.as_ld, (long)bz);
      else
        tile_load<Cfg::A_ROWS, Cfg::A_COLS, VA, Cfg::A_TPR, Cfg::A_RPP, Cfg::A_PASS>(ra, A, g.lda, k0, m0, kend, g.M, tid);
    } else {
      if (g.ascale_mode == 1 && g.a_wrap) {                // (a_wrap 2 = 1 here: only the summation order differs)
                                                           // BK divides a_kblk: a k-tile never straddles blocks
        tile_load<Cfg::A_ROWS, Cfg::A_COLS, VA, Cfg::A_TPR, Cfg::A_RPP, Cfg::A_PASS>(ra, A, g.lda, m0, cur_aoff, g.M, g.a_kblk, tid,
                                                                                      g.ascale, g.as_ld, cur_ablk);
      } else if (g.ascale_mode == 1) {                     // all K columns stored; scale block = k0 / a_kblk
        const long sblk = (nkb > 1 || g.tri == TRI_B_UPPER || g.tri == TRI_B_LOWER) ? cur_kb : cur_ablk;
        tile_load<Cfg::A_ROWS, Cfg::A_COLS, VA, Cfg::A_TPR, Cfg::A_RPP, Cfg::A_PASS>(ra, A, g.lda, m0, k0, g.M, kend, tid,
                                                                                      g.ascale, g.as_ld, sblk);
      } else {
        tile_load<Cfg::A_ROWS, Cfg::A_COLS, VA, Cfg::A_TPR, Cfg::A_RPP, Cfg::A_PASS>(ra, A, g.lda, m0, k0, g.M, kend, tid);
      }
    }
    if constexpr (TB)
      tile_load<Cfg::B_ROWS, Cfg::B_COLS, VB, Cfg::B_TPR, Cfg::B_RPP, Cfg::B_PASS>(rb, B, g.ldb, n0, k0, g.N, kend, tid);
    else
      tile_load<Cfg::B_ROWS, Cfg::B_COLS, VB, Cfg::B_TPR, Cfg::B_RPP, Cfg::B_PASS>(rb, B, g.ldb, k0, n0, kend, g.N, tid);
    cur_kk += 1;
    if (cur_kk == ktiles_per_blk) { cur_kk = 0; cur_kb += 1; }
    cur_aoff += BK;
    if (cur_aoff >= g.a_kblk) { cur_aoff = 0; cur_ablk += 1; }
  };
  auto sstore = [&](int buf) {
    double* as = smem + buf * (Cfg::AS_SZ + Cfg::BS_SZ);
    double* bs = as + Cfg::AS_SZ;
    tile_store<Cfg::A_ROWS, Cfg::A_COLS, VA, Cfg::A_TPR, Cfg::A_RPP, Cfg::A_PASS, Cfg::LDA_S>(ra, as, tid);
    tile_store<Cfg::B_ROWS, Cfg::B_COLS, VB, Cfg::B_TPR, Cfg::B_RPP, Cfg::B_PASS, Cfg::LDB_S>(rb, bs, tid);
  };

  // One k-step feeds 8 k values to 2 x 32 MFMAs: lane (li, lk) supplies k = 8 s + 2 lk + q in half-step q (any
  // assignment of k to the instruction's four k slots is legal as long as A and B agree), so that its two values
  // are adjacent in a k-contiguous image.  v_mfma_f64_4x4x4_4b multiplies, for each of its four blocks b, rows
  // 4b..4b+3 of the A fragment with the columns supplied by lanes 4b..4b+3: lane c supplies column 4 (c & 3) + e of
  // the 16-column block to the e-th MFMA, so MFMA e yields columns {e, 4+e, 8+e, 12+e} and a lane ends up with the
  // four consecutive columns 4 (c & 3) .. 4 (c & 3) + 3 of its row.
  //
  // Pipeline: the unit of work is a half-step (32 MFMAs).  The LDS reads of unit u+1 are issued under the MFMAs
  // of unit u (B fragments per half-step, A fragments per step because one b128 read brings both halves), also
  // ACROSS the k-tile barrier: the barrier sits before the LAST half-step, whose MFMAs cover the first reads of
  // the next k-tile.  The next tile's LDS stores are issued under the half-step that ends step NS-2, long before
  // the barrier that publishes them.  sched_group_barrier pins the interleaving (one LDS instruction, then a few
  // MFMAs): issued back to back, the LDS instructions fill the LDS queue and the wave, which issues in order, sits
  // in front of its own MFMAs.  Measured on the dense K = D*Mp product (tools/gemm_bench): 75.7 TFLOP/s with all
  // LDS/global traffic removed, 55 with everything issued at the point of use, 63 with this schedule.
  constexpr int NS = BK / 8;
  static_assert(NS % 2 == 0, "fragment ping-pong needs an even number of k-steps per k-tile");
  double fa[2][2][FM], fb[2][FN][4];
  auto fragA = [&](int buf, int s8, double (&xa)[2][FM]) {
    const double* as = smem + buf * (Cfg::AS_SZ + Cfg::BS_SZ);
    const int kk = s8 * 8 + lk * 2;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = (i * WR + wr) * 16 + li;
#ifdef DGP_ABLATE_NOLDS
      xa[0][i] = (double)(row + kk) * g.alpha; xa[1][i] = (double)(row - kk) * g.alpha;
#else
      if constexpr (TA) {
        xa[0][i] = as[kk * Cfg::LDA_S + row];
        xa[1][i] = as[(kk + 1) * Cfg::LDA_S + row];
      } else {
        const d2_t v = *reinterpret_cast<const d2_t*>(as + row * Cfg::LDA_S + kk);
        xa[0][i] = v[0]; xa[1][i] = v[1];
      }
#endif
    }
  };
  auto fragB = [&](int buf, int s8, int q, double (&xb)[FN][4]) {
    const double* bs = smem + buf * (Cfg::AS_SZ + Cfg::BS_SZ) + Cfg::AS_SZ;
    const int kk = s8 * 8 + lk * 2 + q;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int cb = (j * WC + wc) * 16;
#ifdef DGP_ABLATE_NOLDS
#pragma unroll
      for (int e = 0; e < 4; ++e) xb[j][e] = (double)(cb + e - kk) * g.alpha;
#else
      if constexpr (TB) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xb[j][e] = bs[(cb + 4 * (li & 3) + e) * Cfg::LDB_S + kk];
      } else {
        const double* pq = bs + kk * Cfg::LDB_S + cb + (li & 3) * 4;
        const d2_t v0 = *reinterpret_cast<const d2_t*>(pq), v1 = *reinterpret_cast<const d2_t*>(pq + 2);
        xb[j][0] = v0[0]; xb[j][1] = v0[1]; xb[j][2] = v1[0]; xb[j][3] = v1[1];
      }
#endif
    }
  };
  // SK: 0 = all column blocks; 1 / 2 = the wave's first / last column block is skipped (k-tiles of a triangular
  // operand's diagonal block where those 16-column blocks of B are structurally zero, see `skip_at` below)
  auto mma = [&](auto sk_c, const double (&xa)[FM], const double (&xb)[FN][4]) {
    constexpr int SK = decltype(sk_c)::value;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if constexpr ((SK == 1 && FN == 2) || (SK == 2 && FN == 2)) { if (j == (SK == 1 ? 0 : FN - 1)) continue; }
#pragma unroll
      for (int i = (SK == 3) ? FM / 2 : 0; i < FM; ++i)      // SK 3: the tile's first BM/2 rows are not needed
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[i][j][e] = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[i], xb[j][e], acc[i][j][e], 0, 0, 0);
    }
  };
  constexpr int N_MFMA_FULL = FM * FN * 4;                          // per half-step
  constexpr int N_RA = TA ? 2 * FM : FM, N_RB = TB ? 4 * FN : 2 * FN;   // LDS read instructions of fragA / fragB
  constexpr int N_WRITE = Cfg::A_PASS + Cfg::B_PASS;
  auto hint = [&](auto sk_c, auto nw_c, auto nr_c) {   // nw LDS writes first, then nr LDS reads, spread over the unit's MFMAs
    constexpr int nw = decltype(nw_c)::value, nr = decltype(nr_c)::value;
    constexpr int N_MFMA = (decltype(sk_c)::value == 3 || (decltype(sk_c)::value != 0 && FN == 2)) ? N_MFMA_FULL / 2 : N_MFMA_FULL;
    constexpr int per = N_MFMA / (nw + nr) > 0 ? N_MFMA / (nw + nr) : 1;
#pragma unroll
    for (int n = 0; n < nw; ++n) {
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // one DS write
      __builtin_amdgcn_sched_group_barrier(0x008, per, 0);   // `per` MFMAs
    }
#pragma unroll
    for (int n = 0; n < nr; ++n) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one DS read
      __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
    }
    if constexpr (N_MFMA - per * (nw + nr) > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA - per * (nw + nr), 0);
  };
  using std::integral_constant;
  // One k-tile.  `stage_next` stores the prefetched next tile into the other LDS buffer: unconditionally (after
  // the last k-tile it rewrites stale registers into a buffer nobody reads before the next prologue), so that the
  // stores sit in the same basic block as the MFMAs.  That buffer was last read before the previous barrier
  // (every wave drains its LDS reads before it arrives there).  `fetch_next2` starts the global loads of a later
  // tile into the registers the store has freed.
  auto ktile = [&](auto sk, int cur, auto&& stage_next, auto&& fetch_next2) {
#pragma unroll
    for (int s8 = 0; s8 < NS; ++s8) {
      // half-step (s8, 0): B fragments of (s8, 1) fly
      fragB(cur, s8, 1, fb[1]);
      mma(sk, fa[s8 & 1][0], fb[0]);
      hint(sk, integral_constant<int, 0>{}, integral_constant<int, N_RB>{});
      __builtin_amdgcn_sched_barrier(0);
      if (s8 < NS - 1) {
        // half-step (s8, 1): fragments of (s8 + 1, 0) fly; the last such unit also carries the LDS stores
        if (s8 == NS - 2) stage_next();
        fragA(cur, s8 + 1, fa[(s8 + 1) & 1]);
        fragB(cur, s8 + 1, 0, fb[0]);
        mma(sk, fa[s8 & 1][1], fb[1]);
        if (s8 == NS - 2) hint(sk, integral_constant<int, N_WRITE>{}, integral_constant<int, N_RA + N_RB>{});
        else hint(sk, integral_constant<int, 0>{}, integral_constant<int, N_RA + N_RB>{});
        __builtin_amdgcn_sched_barrier(0);
      } else {
#ifndef DGP_ABLATE_BARRIER
        __syncthreads();
#endif
        fetch_next2();                       // issues the next global loads if a tile is left (uniform branch)
        __builtin_amdgcn_sched_barrier(0);
        // last half-step: the first fragments of the next k-tile fly (harmless after the last k-tile: valid LDS)
        fragA(cur ^ 1, 0, fa[0]);
        fragB(cur ^ 1, 0, 0, fb[0]);
        mma(sk, fa[s8 & 1][1], fb[1]);
        hint(sk, integral_constant<int, 0>{}, integral_constant<int, N_RA + N_RB>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  constexpr integral_constant<int, 0> SK0{};
  auto first_frags = [&]() { fragA(0, 0, fa[0]); fragB(0, 0, 0, fb[0]); };

  if constexpr (!FAST) {
    // Generic path (edge tiles, unaligned operands): predicated loads.  The triangular structure is
    // exploited at tile level only (k-range per tile, tiles above the diagonal skipped): in-loop predication
    // of the MFMAs made hipcc move the accumulators between VGPRs and AGPRs around every MFMA (2.3x slower).
    if (ktiles > 0) {
      gload();
      sstore(0);
    }
    __syncthreads();
    if (ktiles > 1) gload();
    if (ktiles > 0) first_frags();
    for (long kt = 0; kt < ktiles; ++kt) {
      const int cur = (int)(kt & 1);
      ktile(SK0, cur, [&]() { sstore(cur ^ 1); }, [&]() { if (kt + 2 < ktiles) gload(); });
    }
  } else {
    // ---- interior fast path: running pointers, unconditional 16-byte loads, no predicates ----
    static_assert(!FAST || (VA == 2 && VB == 2), "fast path is vectorised");
    const int a_tr = tid / Cfg::A_TPR, a_tc = (tid % Cfg::A_TPR) * 2;
    const int b_tr = tid / Cfg::B_TPR, b_tc = (tid % Cfg::B_TPR) * 2;
    const bool wrap = (!TA) && SCALED && g.a_wrap == 1;        // ascale_mode 1: K wraps around a_kblk physical columns
    // a_wrap == 2: the same product with the block index INNERMOST (k = kk_tile, then d = 0..nblk-1): the A tile of
    // a k-chunk is fetched nblk times back to back (cache hits) instead of once per pass over the whole row block,
    // which at 16 row blocks x 256 KB per XCD overflowed the 4 MB L2 (the dC launch fetched 9x its operand from HBM)
    const bool wrap2 = (!TA) && SCALED && g.a_wrap == 2;
    const long nblk = wrap2 ? g.K / g.a_kblk : 1;
    long d_idx = 0;
    const bool blk_scale = (!TA) && SCALED && g.a_wrap == 0;       // ... or A holds all K columns: the scale follows the k-blocks
    // Addressing: one wave-uniform base pointer per operand (scalar registers, advanced per k-tile) plus constant
    // 32-bit per-lane byte offsets, so that the loads use the scalar-base form and no 64-bit vector adds.
    const bool ablk = !TA && g.a_blocked;
    const char* ua = reinterpret_cast<const char*>(TA ? A + klo * g.lda + m0
                                                      : ablk ? A + m0 * g.lda + klo * BM
                                                             : A + m0 * g.lda + ((wrap || wrap2) ? 0 : klo));
    const char* ub = reinterpret_cast<const char*>(TB ? B + n0 * g.ldb + klo : B + klo * g.ldb + n0);
    unsigned offa[Cfg::A_PASS], offb[Cfg::B_PASS], offs[Cfg::A_PASS];
#pragma unroll
    for (int p = 0; p < Cfg::A_PASS; ++p)
      offa[p] = (unsigned)(((long)(p * Cfg::A_RPP + a_tr) * (ablk ? (long)BK : g.lda) + a_tc) * 8);
#pragma unroll
    for (int p = 0; p < Cfg::B_PASS; ++p) offb[p] = (unsigned)(((long)(p * Cfg::B_RPP + b_tr) * g.ldb + b_tc) * 8);
    const long a_kstride = (TA ? g.lda : (ablk ? (long)BM : 1)) * 8, b_kstride = (TB ? 1 : g.ldb) * 8;   // bytes per unit of k
    // scale source: mode 1 -> ascale[(m0 + row)*as_ld + d], constant within a d-block; mode 2 -> ascale[k*as_ld + batch]
    const char* us = nullptr;
    long s_kstride = 0;
    if constexpr (SCALED) {
      // transposed A: a tile row is one k (point) and is loaded by a single wave (A_TPR == 64), so its scale is
      // wave-uniform: address it through the scalar wave index and hipcc emits scalar (s_load) fetches
      const int a_tr_s = (TA && Cfg::A_TPR == 64) ? wave : a_tr;
      us = reinterpret_cast<const char*>(TA ? g.ascale + klo * g.as_ld + bz : g.ascale + m0 * g.as_ld);
#pragma unroll
      for (int p = 0; p < Cfg::A_PASS; ++p) offs[p] = (unsigned)(((long)(p * Cfg::A_RPP + a_tr_s) * g.as_ld) * 8);
      s_kstride = TA ? g.as_ld * 8 : 0;
    }
    const long jump = kblen - (ktiles_per_blk - 1) * BK;      // k advance when crossing a triangular block boundary
    const long tiles_per_wrap = wrap ? g.a_kblk / BK : 0;
    long kk_in_blk = 0, kk_in_wrap = 0;
    // Two register sets: the global loads of k-tile u go to set u & 1 and are issued TWO k-tiles before the tile is
    // stored to LDS (with one tile of lead the stores waited on HBM latency: removing the global->LDS staging
    // from the loop was worth 61 -> 72 TFLOP/s on the dense K = D*Mp product).
    d2_t fra[2][Cfg::A_PASS], frb[2][Cfg::B_PASS];
    double fsc[2][Cfg::A_PASS];
    auto fload = [&](int set) {
#pragma unroll
      for (int p = 0; p < Cfg::A_PASS; ++p) fra[set][p] = *reinterpret_cast<const d2_t*>(ua + offa[p]);
#pragma unroll
      for (int p = 0; p < Cfg::B_PASS; ++p) frb[set][p] = *reinterpret_cast<const d2_t*>(ub + offb[p]);
      if constexpr (SCALED) {
#pragma unroll
        for (int p = 0; p < Cfg::A_PASS; ++p) fsc[set][p] = *reinterpret_cast<const double*>(us + offs[p]);
      }
      // advance the cursors (uniform, scalar)
      long ka = BK, kbs = BK;
      kk_in_blk += 1;
      if (kk_in_blk == ktiles_per_blk) { kk_in_blk = 0; ka = jump; kbs = jump; }
      if (blk_scale) {             // scale blocks = triangular blocks when a hint is given, else every a_kblk columns
        kk_in_wrap += 1;
        if (nkb > 1 ? kk_in_blk == 0 : kk_in_wrap == g.a_kblk / BK) { kk_in_wrap = 0; if constexpr (SCALED) us += 8; }
      }
      if (wrap) {
        kk_in_wrap += 1;
        if (kk_in_wrap == tiles_per_wrap) { kk_in_wrap = 0; ka = BK - g.a_kblk; if constexpr (SCALED) us += 8; }
      }
      if (wrap2) {
        d_idx += 1;
        ka = 0; kbs = g.a_kblk;                              // same A columns, next block of B
        long sadv = 8;
        if (d_idx == nblk) { d_idx = 0; ka = BK; kbs = BK - (nblk - 1) * g.a_kblk; sadv = -(nblk - 1) * 8; }
        if constexpr (SCALED) us += sadv;
      }
      ua += ka * a_kstride;
      ub += kbs * b_kstride;
      if constexpr (SCALED) us += BK * s_kstride;
    };
    auto fstore = [&](int set, int buf) {
      double* as = smem + buf * (Cfg::AS_SZ + Cfg::BS_SZ);
      double* bs = as + Cfg::AS_SZ;
#pragma unroll
      for (int p = 0; p < Cfg::A_PASS; ++p) {
        d2_t v = fra[set][p];
        if constexpr (SCALED) { v[0] *= fsc[set][p]; v[1] *= fsc[set][p]; }
        *reinterpret_cast<d2_t*>(as + (p * Cfg::A_RPP + a_tr) * Cfg::LDA_S + a_tc) = v;
      }
#pragma unroll
      for (int p = 0; p < Cfg::B_PASS; ++p)
        *reinterpret_cast<d2_t*>(bs + (p * Cfg::B_RPP + b_tr) * Cfg::LDB_S + b_tc) = frb[set][p];
    };
    if (ktiles > 0) {
      fload(0);
      fstore(0, 0);
    }
    __syncthreads();
    if (ktiles > 1) fload(1);
    if (ktiles > 2) fload(0);
    if (ktiles > 0) first_frags();
    // k-tile kt reads LDS buffer kt & 1, stores tile kt+1 (register set (kt+1) & 1) and then refills that set
    // with tile kt+3; two tiles per trip keep the set indices compile-time.
    // Triangular B with 64-column tiles: inside the diagonal 64 x 64 block of B, the k-tiles (16 rows of B each) meet
    // 16-column blocks that are structurally zero.  Block cb of the tile belongs to wave column wc as j = cb / WC
    // (cb = j * WC + wc), so in the LAST two k-tiles of an upper-triangular range (rows >= 32 of the diagonal block)
    // both waves' j = 0 blocks (columns < 32) are zero, and in the FIRST two of a lower-triangular range both waves'
    // j = 1 blocks are: those k-tiles issue half of the MFMAs (the other two diagonal k-tiles keep one wave busy
    // with both of its blocks, so nothing is gained there).  10 % of the MFMA work of an M = 256 triangular product.
    // Handled as separate phases before / after the main loop (single triangular block only), not as a branch
    // inside it: a three-way dispatch per k-tile made hipcc spill ~400 VGPRs in every instantiation.
    const int tri_sk = (FN != 2 || WC != 2 || !tri_ok || !g.tri_half || khi - klo < 4 * BK || (ktiles_per_blk & 1)) ? 0
                       : (g.tri == TRI_B_UPPER ? 1 : (g.tri == TRI_B_LOWER ? 2 : 0));
    auto pair = [&](auto sk, long kt) {
#ifndef DGP_ABLATE_GLOBAL
      ktile(sk, 0, [&]() { fstore(1, 1); }, [&]() { if (kt + 3 < ktiles) fload(1); });
      if (kt + 1 < ktiles) ktile(sk, 1, [&]() { fstore(0, 0); }, [&]() { if (kt + 4 < ktiles) fload(0); });
#else
      ktile(sk, 0, [&]() {}, [&]() {});
      if (kt + 1 < ktiles) ktile(sk, 1, [&]() {}, [&]() {});
#endif
    };
    if constexpr (ROWSEL == 1) {
      // Lower-triangular OUTPUT, 128 x 64 tiles: the tiles whose first 64 rows lie strictly above the diagonal need
      // only the row blocks i >= FM/2 of both wave rows (row block = i * WR + wr): their whole K loop runs with half of
      // the MFMAs.  Two of the six active tiles of a 256 x 256 block are of this kind (Gram and Q products); they
      // are launched as their own grid (tri_part 2) with this instantiation, because a second copy of the loop
      // inside the general kernel made hipcc spill.
      for (long kt = 0; kt < ktiles; kt += 2) pair(integral_constant<int, 3>{}, kt);
    } else {
      // one pass per triangular block (a single one for the K = Mp products; D of them when K spans the d-blocks)
      const long nblk_run = tri_sk ? nkb : 1, per_run = tri_sk ? ktiles_per_blk : ktiles;
      for (long kb = 0; kb < nblk_run; ++kb) {
        long kt = kb * per_run;
        const long kt_last = kt + per_run;
        if (tri_sk == 2) { pair(integral_constant<int, 2>{}, kt); kt += 2; }
        const long kt_end = tri_sk == 1 ? kt_last - 2 : kt_last;
        for (; kt < kt_end; kt += 2) pair(SK0, kt);
        if (tri_sk == 1) pair(integral_constant<int, 1>{}, kt_last - 2);
      }
    }
  }

  // ---- epilogue: acc[i][j][e] of lane l is C[rowblk*16 + 4*((l&15)>>2) + (l>>4)][colblk*16 + 4*(l&3) + e]
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] *= g.alpha;
  if (g.epi != 0) {
    // per-row sums of squares over this wave's columns: a lane holds 4 values of its row per fragment, the
    // 4 lanes l&3 = 0..3 of a quad hold the rest of that row's 16 columns
    double* __restrict__ rs = g.rowsq + ((long)tn * WC + wc) * g.rowsq_ld;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      double q = 0.0;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) q += acc[i][j][r] * acc[i][j][r];
      q += __shfl_xor(q, 1);
      q += __shfl_xor(q, 2);
      const long row = m0 + (long)(i * WR + wr) * 16 + 4 * (li >> 2) + lk;
      if ((li & 3) == 0 && row < g.M) rs[row] = q;
    }
    if (g.epi == 1) continue;
  }
  if (g.rank > 0) {
    // rank-R update: a lane owns FM rows and 4*FN columns, so per q it needs FM + 4*FN factor values for its
    // FM*FN*4 outputs (fetching both factors per output element cost 2 ms per dC launch at config 2)
    for (int q = 0; q < g.rank; ++q) {
      double rfq[FM], cfq[FN][4];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const long row = m0 + (long)(i * WR + wr) * 16 + 4 * (li >> 2) + lk;
        rfq[i] = row < g.M ? g.rowf[row * g.rank + q] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long col = n0 + (long)(j * WC + wc) * 16 + 4 * (li & 3) + r;
          cfq[j][r] = col < g.N ? g.colf[col * g.rank + q] : 0.0;
        }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] += rfq[i] * cfq[j][r];
    }
  }
  const bool atomic = g.splits > 1;
  if constexpr (FAST && !SCALED) {   // (the scaled kernels sit at the register limit: the extra path made them spill)
    if (!atomic && !g.beta && g.eadd == nullptr) {
      // interior tile, 16-byte aligned outputs (host-checked): a lane's four values of a fragment are four
      // consecutive columns -> two 16-byte stores instead of four scattered 8-byte ones
      // row-major: row * ldc + col.  Blocked: panel base + 16-column block * (BM * 16) + row-in-tile * 16 + col-in-block
      const long o00 = g.c_blocked ? m0 * g.ldc + (n0 / 16 + wc) * (long)(BM * 16) + ((long)wr * 16 + 4 * (li >> 2) + lk) * 16 + 4 * (li & 3)
                                   : (m0 + (long)wr * 16 + 4 * (li >> 2) + lk) * g.ldc + n0 + (long)wc * 16 + 4 * (li & 3);
      const long istep = g.c_blocked ? (long)WR * 16 * 16 : (long)WR * 16 * g.ldc;
      const long jstep = g.c_blocked ? (long)WC * BM * 16 : (long)WC * 16;
#pragma unroll
      for (int i = (ROWSEL == 1 ? FM / 2 : 0); i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const long o = o00 + i * istep + j * jstep;
          d2_t v0 = {acc[i][j][0], acc[i][j][1]}, v1 = {acc[i][j][2], acc[i][j][3]};
          if (g.C2 != nullptr) {
            *reinterpret_cast<d2_t*>(g.C2 + o) = v0 * *reinterpret_cast<const d2_t*>(g.emul + o);
            *reinterpret_cast<d2_t*>(g.C2 + o + 2) = v1 * *reinterpret_cast<const d2_t*>(g.emul + o + 2);
          }
          if (!g.c2_only) {
            *reinterpret_cast<d2_t*>(C + o) = v0;
            *reinterpret_cast<d2_t*>(C + o + 2) = v1;
          }
        }
      continue;
    }
  }
#pragma unroll
  for (int i = (ROWSEL == 1 ? FM / 2 : 0); i < FM; ++i) {    // ROWSEL 1: rows above the diagonal are neither computed nor stored
    double esc = 0.0;
    if (g.eadd != nullptr) {
      const long row = m0 + (long)(i * WR + wr) * 16 + 4 * (li >> 2) + lk;
      if (row < g.M)
        for (int q = 0; q < g.eadd_nsc; ++q) esc += g.ascale[row * g.as_ld + q];
      esc *= g.alpha;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const long row = m0 + (long)(i * WR + wr) * 16 + 4 * (li >> 2) + lk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long col = n0 + (long)(j * WC + wc) * 16 + 4 * (li & 3) + r;
        if (row < g.M && col < g.N) {
          double* p = C + row * g.ldc + col;
          double v = acc[i][j][r];
          if (g.eadd != nullptr) v -= esc * g.eadd[row * g.ldc + col];
          if (g.c2_only) {}
          else if (atomic) unsafeAtomicAdd(p, v);
          else if (g.beta) *p += v;
          else *p = v;
          if (g.C2 != nullptr) g.C2[row * g.ldc + col] = v * g.emul[row * g.ldc + col];
        }
      }
    }
  }
  }   // persistent tile loop
#ifdef DGP_CLOCK_STAMPS
  if (threadIdx.x == 0 && g.C2 == nullptr && g.emul != nullptr) {
    long long* st = reinterpret_cast<long long*>(const_cast<double*>(g.emul));
    st[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = clock64() - stamp_c0;
    st[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = wall_clock64() - stamp_r0;
  }
#endif
}

// Host-side dispatcher (defined in gemm_f64.hip)
enum GemmOp : int { GEMM_NN = 0, GEMM_NT = 1, GEMM_TN = 2 };
hipError_t gemm_f64(hipStream_t st, GemmOp op, const GemmArgs& args);
// which kernel family gemm_f64 selects for this call: 0 the 128 x 64 engine, 1 wide-tile, 2 tall-tile (T), 3 tall-tile (dC),
// 4 weighted Gram, 5 one-workgroup small product (tests assert that a size reaches the kernel it is meant to check)
int gemm_engine_of(GemmOp op, const GemmArgs& args);
// CUs left free by the persistent kernels while a collective may run beside them (gemm_f64.hip)
int gemm_device_cus(const GemmArgs& a);                 // a.cu_count, else the current device's CU count
int gemm_persistent_grid(const GemmArgs& a, int cus);   // workgroups of a one-per-CU kernel: cus less a.reserve_cus
// one-workgroup products of the small models' chains (gemm_small.hip)
bool gemm_small_ok(GemmOp op, const GemmArgs& a);
hipError_t gemm_small(hipStream_t st, GemmOp op, const GemmArgs& a);
// tall-tile kernel for the forward T product (gemm_tall.h / gemm_tall.hip)
bool gemm_tall_ok(const GemmArgs& a);
hipError_t gemm_tall(hipStream_t st, const GemmArgs& a);
// tall-tile kernel for the backward dC product (gemm_tallu.h / gemm_tallu.hip)
bool gemm_tallu_ok(const GemmArgs& a);
hipError_t gemm_tallu(hipStream_t st, const GemmArgs& a);
// gemm_dcpanel.h / .hip: the same product from a row-major T (128-row panels, T read once)
bool dc_panel_shape_ok(long M, long Mp, long D);     // would the dC product of this shape run there?  (then T is written row-major)
bool dc_panel_ok(const GemmArgs& a);
hipError_t dc_panel(hipStream_t st, const GemmArgs& a);
// weighted Gram products over the points (gemm_gram.h / gemm_gram.hip)
// latency-oriented products of the chains' Mp x Mp matrices (gemm_mid.hip)
bool gemm_mid_ok(GemmOp op, const GemmArgs& a);
hipError_t gemm_mid(hipStream_t st, GemmOp op, const GemmArgs& a);
bool gemm_gram_ok(const GemmArgs& a);
long gemm_gram_ws_bytes(int cu_count);      // cu_count <= 0: the current device
hipError_t gemm_gram(hipStream_t st, const GemmArgs& a);

}  // namespace dgp
