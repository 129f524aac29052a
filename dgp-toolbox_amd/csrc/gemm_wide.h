// Wide-tile fp64 MFMA kernel for the row-panel products of the SVGP-layer path whose B operand is an Mp x Mp
// (block-)triangular matrix:
//   forward   Ct  = Kt * Linv^T            (layers.py:245-247, tf.linalg.triangular_solve)      B upper  (k <= n)
//             Tt  = Ct * Wcat  (+ |t_d|^2) (layers.py:254-263, SK @ A in triangular form)       B lower  (k >= n)
//   backward  dCt = [2 vbar .* Tt] * WTcat - (sum 2 vbar) Ct + mbar u^T                         B upper, K = D*Mp
//             dKt = dCt * Linv  (+ g = dK .* k)                                                 B lower
// Why a second kernel next to gemm_f64.h's 128 x 64 engine: those products have K = Mp (256) per triangular block.
// With 64-column tiles every output tile refills a cold pipeline after 4..16 k-tiles, the column tiles of a row panel
// have unequal k-ranges (1 : 4), the A panel is re-read once per column tile, and the triangular skip works at 64-column
// granularity (executed / algorithmic MFMA work 1.18).  Here ONE workgroup owns a 128-row panel across a whole
// 256-column block:
//   * tile 128 x 256, 8 waves as 4 (rows) x 2 (columns), wave tile 32 x 128; wave column wc owns the 16-column blocks
//     cb = 2 j + wc (j = 0..7), so that inside the diagonal 256 x 256 block of B the k-tile t (16 rows of B) needs
//     j >= t/2 (upper) or j <= t/2 (lower) for BOTH wave columns: eight compile-time variants of the k-tile body,
//     identical instruction streams in all waves, executed / algorithmic = 144/136 = 1.06;
//   * operands staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into a ring of three
//     48 KB stages, requested three k-tiles ahead and retired by a counted vmcnt before a raw s_barrier; the stream of
//     k-tiles runs on ACROSS output tiles (persistent workgroups, one per CU), so the pipeline is filled once per
//     launch, not once per tile;
//   * LDS images are lane-linear per DMA piece, XOR-swizzled on the source address (A: 16-byte chunk ^ ((row >> 1) & 7),
//     B: chunk ^ ((krow >> 1) & 1)) so that all fragment reads are conflict-free ds_read_b128;
//   * fragment reads run two units (one 16-column block of one k-substep = 8 MFMAs) ahead of their MFMAs, also across
//     the k-tile barrier (the barrier sits two units before the end of a k-tile).
// MFMA instruction and lane maps: see gemm_f64.h (v_mfma_f64_4x4x4_4b_f64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gemm_f64.h"

namespace dgp {

constexpr int WBM = 128, WBN = 256, WBK = 16;
constexpr int W_A_BYTES = WBM * WBK * 8;          // 16 KB: [128 rows][8 chunks of 16 B]
constexpr int W_B_BYTES = WBK * WBN * 8;          // 32 KB: [16 k rows][128 chunks of 16 B]
constexpr int W_STAGE = W_A_BYTES + W_B_BYTES;    // 48 KB
constexpr int W_NSTAGE = 3;
constexpr int W_MAXD = 16;                        // scale blocks per row kept in LDS (SCALED kernels)
constexpr int W_GLDS = 6;                         // LDS-DMA instructions per wave and k-tile (2 for A, 4 for B)

#define W_LDS3(p) ((__attribute__((address_space(3))) void*)(p))
#define W_GLB1(p) ((const __attribute__((address_space(1))) void*)(p))

template <int B, int E, class F>
__device__ __forceinline__ void w_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    w_static_for<B + 1, E>(f);
  }
}

// position in a workgroup's stream of k-tiles (all members wave-uniform)
struct WCur {
  unsigned lin;        // position in the workgroup's list of output tiles (blockIdx.x + r * gridDim.x)
  unsigned tm, tn;     // output tile
  int b, t;            // k block, k-tile inside the block
  int t_lo, t_hi;      // k-tile range inside a block
  int t_d0;            // first k-tile of the diagonal 256 x 256 block
  int valid;
};

// DIR: 0 dense B, 1 B upper (k mod triblk <= n mod triblk), 2 B lower.  SCALED: A[m][k] *= ascale[m*as_ld + k / kblen].
// EM: epilogue mode.  0 = generic (every GemmArgs epilogue feature, the whole tile written after its last k-tile);
//   1 = alpha 1, row sums of squares + C (row-major or blocked); 2 = alpha 1, C and C2 = C .* emul; 3 = alpha 1, row
//   sums of squares only.  Modes 1-3 write one block column at a time, interleaved with the k-tile stream (the products
//   with one k block per tile: an epilogue every 16 k-tiles), through buffer stores with per-lane offsets computed once.
//   4 = mode 1 of the forward product Ct = Kt LinvT (DIR 1, N = K = triblk = 256) that ALSO produces the layer's mean
//   before the mean function, mean0 = Ct u (layers.py:249, whitened), as Kt alpha with alpha = LinvT u [256 x D <= 8]
//   (prepared with the layer's matrices): alpha sits in the 16 KB of LDS the ring leaves free, every k-substep gets two more
//   MFMAs (the A fragments of the wave's two row blocks times alpha's k rows; wave column wc owns outputs 4 wc .. 4 wc + 3):
//   128 MFMAs per tile next to 2304, instead of a launch of its own that read Ct once more (2 GB per 10^6 points).
// Host-guaranteed (gemm_wide_ok): A not transposed, B not transposed, M % 128 == 0, N % 256 == 0, K % 16 == 0,
// triblk % 256 == 0, K % min(K, triblk) == 0, batch == splits == 1, beta == 0, 16-byte aligned operands.
template <int DIR, bool SCALED, int EM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_wide_kernel(GemmArgs g) {
  constexpr int LDS_BYTES = W_NSTAGE * W_STAGE + ((SCALED || EM == 4) ? WBM * W_MAXD * 8 : 0);
  static_assert(EM != 4 || (DIR == 1 && !SCALED), "mode 4 is the forward product Ct = Kt LinvT");
  constexpr bool E_RS = EM == 1 || EM == 3 || EM == 4;      // row sums of squares
  constexpr bool E_ST = EM == 1 || EM == 2 || EM == 4;      // C is stored
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lk = lane >> 4;

  const unsigned tiles_n = (unsigned)(g.N / WBN), tiles_m = (unsigned)(g.M / WBM), ntile = tiles_n * tiles_m;
  const int kblen = (DIR != 0 && g.K > g.triblk) ? (int)g.triblk : (int)g.K;
  const int nkb = (int)(g.K / kblen), ktb = kblen / WBK;
  const unsigned Q = DIR != 0 ? (unsigned)(g.triblk / WBN) : 1u;
  if (blockIdx.x >= ntile) return;

  auto setup = [&](WCur& c) __attribute__((always_inline)) {
    unsigned bid = c.lin;
    const unsigned q = ntile / 8u, r = ntile % 8u, xcd = bid % 8u, idx = bid / 8u;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // one XCD's L2 sees a contiguous run of tiles
    c.tm = bid / tiles_n;
    c.tn = bid - c.tm * tiles_n;
    const int nq = DIR != 0 ? (int)(c.tn % Q) : 0;
    c.t_d0 = DIR != 0 ? 16 * nq : (1 << 28);
    c.t_lo = DIR == 2 ? 16 * nq : 0;
    c.t_hi = DIR == 1 ? 16 * nq + 16 : ktb;
    c.b = 0;
    c.t = c.t_lo;
  };
  auto advance = [&](WCur& c) __attribute__((always_inline)) {          // past the end the cursor stays on the last k-tile (valid = 0)
    if (c.t + 1 < c.t_hi) { c.t += 1; return; }
    if (c.b + 1 < nkb) { c.b += 1; c.t = c.t_lo; return; }
    if (c.lin + gridDim.x < ntile) { c.lin += gridDim.x; setup(c); return; }
    c.valid = 0;
  };
  // variant of a k-tile = (J0, J1), the range of the wave's column blocks j that B's structure leaves non-zero:
  //   DIR 1: J0 = v, J1 = 7;  DIR 2: J0 = 0, J1 = v;  dense k-tiles are v = 0 (DIR 1) / v = 7 (DIR 2)
  auto variant = [&](const WCur& c) __attribute__((always_inline)) -> int {
    if constexpr (DIR == 0) return 0;
    if constexpr (DIR == 1) return c.t < c.t_d0 ? 0 : (c.t - c.t_d0) >> 1;
    return c.t >= c.t_d0 + 16 ? 7 : (c.t - c.t_d0) >> 1;
  };

  // ---- LDS-DMA sources (buffer_load_dwordx4 ... lds: wave-uniform resource + scalar offset + one 32-bit lane offset;
  //      the LDS side is lane-linear).  The MUBUF form on purpose: with the FLAT form (global_load_lds) hipcc treats a
  //      DMA in flight as a pending FLAT access, turns every LDS wait into lgkmcnt(0) and every use of a reloaded
  //      register into vmcnt(0), and keeps six 64-bit lane addresses alive (spills inside the k-tile loop).
  // A piece P (8 rows x 128 B) of wave w: P = 2 w + p; lane -> row = 8 P + lane / 8, physical chunk = lane % 8, logical
  //   chunk = physical ^ ((row >> 1) & 7) = (lane & 7) ^ (4 p + lane / 16)
  // B piece P (half a k row, 1 KB) of wave w: P = 4 w + p; krow = P / 2 = 2 w + p / 2, physical chunk = 64 (P % 2) + lane,
  //   logical = physical ^ ((krow >> 1) & 1) = physical ^ (w & 1)
  const long ldaA = g.a_blocked ? (long)WBK : g.lda;
  unsigned offA[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int c = (lane & 7) ^ (4 * p + (lane >> 4));
    offA[p] = (unsigned)(((long)(lane >> 3) * ldaA + 2 * c) * 8);
  }
  const unsigned offB = (unsigned)((lane ^ (wave & 1)) * 16);
  const unsigned stepA = (unsigned)(8 * ldaA * 8), stepB = (unsigned)(g.ldb * 8);     // bytes per A piece / per k row of B
  auto issue = [&](const WCur& c, int stage) __attribute__((always_inline)) {
#ifdef W_ABL_NOISSUE
    return;
#endif
    const long kglob = (long)c.b * kblen + (long)c.t * WBK;
    const double* ua = (g.a_blocked ? g.A + (long)c.tm * WBM * g.lda + kglob * WBM : g.A + (long)c.tm * WBM * g.lda + kglob) +
                       (long)(wave * 2) * 8 * ldaA;
    const double* ub = g.B + (kglob + wave * 2) * g.ldb + (long)c.tn * WBN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(ua), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(ub), 0, 0x7ffffff0, 0x00020000);
    char* base = smem + stage * W_STAGE;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, W_LDS3(base + (wave * 2 + p) * 1024), 16, offA[p], p * stepA, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, W_LDS3(base + W_A_BYTES + (wave * 4 + p) * 1024), 16, offB,
                                               (p >> 1) * stepB + (p & 1) * 1024, 0, 0);
  };

  // ---- fragment addresses (bytes from smem, stage included; advanced by the stage step at every k-tile barrier) ----
  //   A: row = (2 wr + i) 16 + li, logical chunk s8 * 4 + lk  ->  row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
  //   B: krow = 8 s8 + 2 lk + q, columns (2 j + wc) 16 + 4 (li & 3) + e: logical chunks c0, c0 + 1 with
  //      c0 = (2 j + wc) 8 + 2 (li & 3); (krow >> 1) & 1 = lk & 1 swaps the two 16-byte halves
  unsigned bA[2][2], bB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wr * 2 + i) * 16 + li;
#pragma unroll
    for (int s8 = 0; s8 < 2; ++s8) bA[i][s8] = (unsigned)(row * 128 + (((s8 * 4 + lk) ^ ((row >> 1) & 7)) << 4));
  }
  {
    const int sw = lk & 1;
    const unsigned common = (unsigned)(W_A_BYTES + (2 * lk) * 2048 + wc * 128 + (li & 3) * 32);
    bB[0] = common + sw * 16;
    bB[1] = common + (1 - sw) * 16;
  }

  double acc[2][8][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
  double fa[2][2][2];       // [s8][q][i]
  double fb[4][4];          // ring of four units: [slot][e]
  double sc[2] = {1.0, 1.0};
  // mode 4: alpha's k row of the next k-substep (ONE value, read a substep ahead of its MFMAs: the kernel has no registers
  // to spare), the mean accumulators: macc[i] of lane (li, lk) is mean0[m0 + (2 wr + i) 16 + 4 (li >> 2) + lk][4 wc + (li & 3)]
  double fm = 0.0, macc[2] = {0.0, 0.0};
  auto read_fm = [&](auto tc, auto sc_) __attribute__((always_inline)) {       // alpha[16 t + 8 s8 + 2 lk + q][4 wc + (li & 3)]
    constexpr int t = decltype(tc)::value, sub = decltype(sc_)::value;
    int l2 = lane;
    asm volatile("" : "+v"(l2));          // (the address is re-derived at every use instead of living in a register)
    const unsigned amean = (unsigned)(W_NSTAGE * W_STAGE + ((2 * (l2 >> 4)) * 8 + 4 * wc + (l2 & 3)) * 8);
    fm = *reinterpret_cast<const double*>(smem + amean + (unsigned)((16 * t + 8 * (sub >> 1) + (sub & 1)) * 64));
  };

  auto readA = [&](int s8) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#ifdef W_ABL_NOREAD
      const d2_t v = {(double)bA[i][s8], 1.0};
#else
      const d2_t v = *reinterpret_cast<const d2_t*>(smem + bA[i][s8]);
#endif
      fa[s8][0][i] = v[0];
      fa[s8][1][i] = v[1];
    }
  };
  auto readB = [&](int slot, unsigned off) __attribute__((always_inline)) {   // off: (8 s8 + q) * 2048 + j * 256
#ifdef W_ABL_NOREAD
    const d2_t v0 = {(double)(bB[0] + off), 2.0}, v1 = {(double)(bB[1] + off), 3.0};
#else
    const d2_t v0 = *reinterpret_cast<const d2_t*>(smem + bB[0] + off);
    const d2_t v1 = *reinterpret_cast<const d2_t*>(smem + bB[1] + off);
#endif
    fb[slot][0] = v0[0]; fb[slot][1] = v0[1]; fb[slot][2] = v1[0]; fb[slot][3] = v1[1];
  };
  // offsets of the first two units of a k-tile of variant v
  auto first_units = [&](int v, unsigned& o0, unsigned& o1) __attribute__((always_inline)) {
    const int j0 = DIR == 1 ? v : 0, nact = DIR == 0 ? 8 : (DIR == 1 ? 8 - v : v + 1);
    o0 = (unsigned)(j0 * 256);
    o1 = nact == 1 ? (unsigned)(2048 + j0 * 256) : (unsigned)((j0 + 1) * 256);
  };

  // SCALED: the scales of the 128 rows of an output tile, all k blocks, are copied to LDS when the tile starts
  double* sreg = reinterpret_cast<double*>(smem + W_NSTAGE * W_STAGE);
  const int nsc = SCALED ? nkb : 0;
  auto stage_scales = [&](const WCur& c) __attribute__((always_inline)) {
    if constexpr (SCALED) {
      for (int x = tid; x < WBM * nsc; x += 512) {
        const int row = x / nsc, d = x - row * nsc;
        sreg[row * W_MAXD + d] = g.ascale[((long)c.tm * WBM + row) * g.as_ld + d];
      }
    }
  };
  auto load_scales = [&](int b) __attribute__((always_inline)) {
    if constexpr (SCALED) {
#pragma unroll
      for (int i = 0; i < 2; ++i) sc[i] = sreg[((wr * 2 + i) * 16 + li) * W_MAXD + b];
    }
  };

  WCur cons, load;
  cons.lin = blockIdx.x;
  cons.valid = 1;
  setup(cons);
  load = cons;
  int stage = 0;
  // vmcnt counts loads and stores in issue order.  A k-tile's DMA wait must let the youngest W_GLDS requests (the k-tile
  // two ahead) stay in flight, plus every store issued after the requests it waits for: st_a = stores that the NEXT wait
  // has to allow for, st_b = those the wait after it has to (a lower bound is safe: it only waits longer).
  int st_a = 0, st_b = 0;

  if constexpr (EM == 4) {        // alpha [256][D] -> LDS [256][8], zero-padded (published by the prologue's barrier)
    double* al = reinterpret_cast<double*>(smem + W_NSTAGE * W_STAGE);
    for (int x = tid; x < 256 * 8; x += 512) {
      const int d = x & 7;
      al[x] = d < g.mean_d ? g.mean_alpha[(x >> 3) * g.mean_d + d] : 0.0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  // ---- prologue: three k-tiles in flight, the first one landed and published, its first fragments requested ----
  issue(load, 0); advance(load);
  issue(load, 1); advance(load);
  issue(load, 2); advance(load);
  stage_scales(cons);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    unsigned o0, o1;
    first_units(variant(cons), o0, o1);
    readA(0);
    readB(0, o0);
    readB(1, o1);
  }
  if constexpr (EM == 4) {
    read_fm(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  }

  using std::integral_constant;
  // ---- epilogue of ONE block column j (16 columns per wave column) of the output tile (tm, tn):
  // acc[i][j][e] of lane (li, lk) is C[m0 + (2 wr + i) 16 + 4 (li >> 2) + lk][n0 + (2 j + wc) 16 + 4 (li & 3) + e].
  // The block columns of a tile are written one at a time, interleaved with the k-tile stream (see the main loop): a
  // whole-tile epilogue is a burst of 32 stores per lane that every workgroup issues at the same moment, and on gfx9
  // an older store holds back the vmcnt of the younger DMA requests, so the stream stalled until the burst had drained
  // (T product: 11.7 ms against 9.3 ms with the stores removed).
  double rsum[2] = {0.0, 0.0};    // row sums of squares of the block columns finished so far
  double esc[2] = {0.0, 0.0};     // alpha * sum_q ascale[row][q] of the tile being finished (eadd term, generic mode)
  // modes 1-3: byte offset of the lane's piece of a block column inside the tile: row pitch = 16 doubles (blocked C)
  // or ldc; the block column (2 j + wc) adds a wave-uniform offset
  unsigned vo[2] = {0u, 0u};
  const long cpitch = g.c_blocked ? 16L : g.ldc;
  if constexpr (EM != 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) vo[i] = (unsigned)((((wr * 2 + i) * 16 + 4 * (li >> 2) + lk) * cpitch + 4 * (li & 3)) * 8);
  }
  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  d2_t em[2][2];                  // mode 2: the emul values of the block column about to be written
  // base of block column (2 j + wc) of tile (tm, tn).  The block offset goes into the resource base, NOT into the
  // instruction's SGPR offset: hipcc's hazard recognizer assumes that a 16-byte buffer store with an SGPR offset needs
  // no wait state before its data registers are overwritten, reused the first data register for the next store's
  // address one instruction later, and on gfx950 that store then wrote the address into the low dword (0.01 % of the
  // values).  With a zero soffset the compiler inserts the wait state itself.
  auto tile_base = [&](const double* C, unsigned tm, unsigned tn, int j) __attribute__((always_inline)) {
    const long m0 = (long)tm * WBM, n0 = (long)tn * WBN;
    return C + m0 * g.ldc + (g.c_blocked ? ((n0 >> 4) + 2 * j + wc) * (long)(WBM * 16) : n0 + (2 * j + wc) * 16);
  };
  auto epi_fetch = [&](auto jc, unsigned tm, unsigned tn) __attribute__((always_inline)) {   // mode 2: request emul
    if constexpr (EM == 2) {
      constexpr int j = decltype(jc)::value;
      const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(tile_base(g.emul, tm, tn, j)), 0, 0x7ffffff0, 0x00020000);
      constexpr unsigned so = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        em[i][0] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(re, vo[i], so, 0));
        em[i][1] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(re, vo[i] + 16, so, 0));
      }
    }
  };
  auto epi_fast = [&](auto jc, unsigned tm, unsigned tn, bool last_block, bool after_dma) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    if constexpr (E_RS) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) rsum[i] += acc[i][j][e] * acc[i][j][e];
      if (last_block) {
        int li2 = lane & 15, lk2 = lane >> 4;
        asm volatile("" : "+v"(li2), "+v"(lk2));
        double* __restrict__ rs = g.rowsq + ((long)tn * 2 + wc) * g.rowsq_ld + (long)tm * WBM;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          double t = rsum[i];
          t += __shfl_xor(t, 1);
          t += __shfl_xor(t, 2);
          if ((li2 & 3) == 0) rs[(wr * 2 + i) * 16 + 4 * (li2 >> 2) + lk2] = t;
          rsum[i] = 0.0;
        }
      }
    }
    if constexpr (E_ST) {
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(tile_base(g.C, tm, tn, j)), 0, 0x7ffffff0, 0x00020000);
      constexpr unsigned so = 0;
      if constexpr (EM == 2) {
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(tile_base(g.C2, tm, tn, j)), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const d2_t g0 = (d2_t){acc[i][j][0], acc[i][j][1]} * em[i][0], g1 = (d2_t){acc[i][j][2], acc[i][j][3]} * em[i][1];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, g0), r2, vo[i], so, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, g1), r2, vo[i] + 16, so, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const d2_t v0 = {acc[i][j][0], acc[i][j][1]}, v1 = {acc[i][j][2], acc[i][j][3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v0), rc, vo[i], so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v1), rc, vo[i] + 16, so, 0);
      }
      constexpr int nst = EM == 2 ? 8 : 4;
      st_a += nst;
      if (after_dma) st_b += nst;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
  };

  // ---- generic epilogue of block column j (mode 0): every feature, plain addressing, runtime branches
  auto epi_block = [&](auto jc, unsigned tm, unsigned tn, bool last_block, bool after_dma) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const long m0 = (long)tm * WBM, n0 = (long)tn * WBN;
    // lane coordinates re-derived behind an opaque barrier: otherwise this arithmetic is hoisted out of the k-tile loop
    // and its values spill the accumulators inside the loop
    int li = lane & 15, lk = lane >> 4;
    asm volatile("" : "+v"(li), "+v"(lk));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] *= g.alpha;
    if (g.epi != 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) rsum[i] += acc[i][j][e] * acc[i][j][e];
      if (last_block) {
        double* __restrict__ rs = g.rowsq + ((long)tn * 2 + wc) * g.rowsq_ld;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          double t = rsum[i];
          t += __shfl_xor(t, 1);
          t += __shfl_xor(t, 2);
          if ((li & 3) == 0) rs[m0 + (wr * 2 + i) * 16 + 4 * (li >> 2) + lk] = t;
          rsum[i] = 0.0;
        }
      }
    }
#ifdef W_ABL_NOEPI
    if (g.alpha == 123.456) {
#else
    if (g.epi != 1) {
#endif
      const long colb = n0 + (2 * j + wc) * 16 + 4 * (li & 3);
      // rank-R update C += rowf[row][r] colf[col][r]: all factor loads of the block column are issued before the first
      // use (a runtime loop over r was 8 dependent L2 round trips per block column: 50 us per tile)
      auto rank_update = [&](auto rc_) __attribute__((always_inline)) {
        constexpr int R = decltype(rc_)::value;
        double cf[4][R], rf[2][R];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < R; ++r) cf[e][r] = g.colf[(colb + e) * R + r];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < R; ++r) rf[i][r] = g.rowf[(m0 + (wr * 2 + i) * 16 + 4 * (li >> 2) + lk) * R + r];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] += rf[i][r] * cf[e][r];
      };
      if (g.rank == 8) rank_update(integral_constant<int, 8>{});
      else if (g.rank == 1) rank_update(integral_constant<int, 1>{});
      else if (g.rank > 0) {
        for (int r = 0; r < g.rank; ++r) {
          double cf[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) cf[e] = g.colf[(colb + e) * g.rank + r];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const double rf = g.rowf[(m0 + (wr * 2 + i) * 16 + 4 * (li >> 2) + lk) * g.rank + r];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] += rf * cf[e];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long rit = (wr * 2 + i) * 16 + 4 * (li >> 2) + lk, row = m0 + rit;
        const long o = g.c_blocked ? m0 * g.ldc + (colb >> 4) * (long)(WBM * 16) + rit * 16 + 4 * (li & 3) : row * g.ldc + colb;
        d2_t v0 = {acc[i][j][0], acc[i][j][1]}, v1 = {acc[i][j][2], acc[i][j][3]};
        if (g.eadd != nullptr) {
          v0 -= esc[i] * *reinterpret_cast<const d2_t*>(g.eadd + o);
          v1 -= esc[i] * *reinterpret_cast<const d2_t*>(g.eadd + o + 2);
        }
        if (g.C2 != nullptr) {
          *reinterpret_cast<d2_t*>(g.C2 + o) = v0 * *reinterpret_cast<const d2_t*>(g.emul + o);
          *reinterpret_cast<d2_t*>(g.C2 + o + 2) = v1 * *reinterpret_cast<const d2_t*>(g.emul + o + 2);
        }
        *reinterpret_cast<d2_t*>(g.C + o) = v0;
        *reinterpret_cast<d2_t*>(g.C + o + 2) = v1;
      }
      const int nst = g.C2 != nullptr ? 8 : 4;
      st_a += nst;
      if (after_dma) st_b += nst;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
  };
  // eadd term: the row factor alpha * sum_q ascale[row][q], from the tile's scales in LDS (SCALED kernels)
  auto load_esc = [&]() __attribute__((always_inline)) {
    if (g.eadd != nullptr) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rit = (wr * 2 + i) * 16 + 4 * ((lane & 15) >> 2) + (lane >> 4);
        double t = 0.0;
        if constexpr (SCALED) {
          for (int q = 0; q < g.eadd_nsc; ++q) t += sreg[rit * W_MAXD + q];
        } else {
          for (int q = 0; q < g.eadd_nsc; ++q) t += g.ascale[((long)cons.tm * WBM + rit) * g.as_ld + q];
        }
        esc[i] = t * g.alpha;
      }
    }
  };

  // One k-tile of variant (J0, J1) from the current stage.  Two units before its end: every LDS read of the stage has
  // been issued -> wait for them and for the DMA of the next k-tile, barrier, request the next k-tile's first
  // fragments, run `pre_dma` (block-column epilogues whose stores should be OLDER than the next DMA requests), and
  // refill the stage just released with the k-tile three ahead.
  // `tkc` (mode 4): the k-tile's index inside the 256-deep k block, -1 = no mean work (k-tiles mode 4 never reaches)
  auto ktile = [&](auto j0c, auto j1c, unsigned nb0, unsigned nb1, int first_of_block, int blk, auto&& pre_dma, auto fence_c, auto tkc) __attribute__((always_inline)) {
    constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value, TK = decltype(tkc)::value;
    constexpr bool FENCE = decltype(fence_c)::value;   // the last pair carries a block-column epilogue: keep its stores ahead of the DMA requests
    constexpr int NACT = J1 - J0 + 1, U = 4 * NACT;
    if constexpr (SCALED) {
      if (first_of_block) load_scales(blk);
    }
    // Units (one 16-column block of one k-substep, 8 MFMAs) are processed in PAIRS: the LDS reads of the next pair are
    // issued right after the first MFMA of the current pair, one read per MFMA.
    w_static_for<0, U / 2>([&](auto pc) __attribute__((always_inline)) {
      constexpr int p = decltype(pc)::value, u0 = 2 * p, u1 = u0 + 1;
      constexpr bool lastp = (p == U / 2 - 1);
      auto mma1 = [&](auto uc, auto ic, auto ec) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, i = decltype(ic)::value, e = decltype(ec)::value;
        constexpr int sub = u / NACT, j = J0 + u % NACT, s8 = sub >> 1, q = sub & 1;
        acc[i][j][e] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[s8][q][i], fb[u & 3][e], acc[i][j][e], 0, 0, 0);
      };
      if constexpr (lastp) {
        __builtin_amdgcn_sched_barrier(0);
        if (st_a >= 32) asm volatile("s_waitcnt vmcnt(38) lgkmcnt(0)" ::: "memory");
        else if (st_a >= 8) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
        else if (st_a >= 4) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
#ifndef W_ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        st_a = st_b;
        st_b = 0;
        const int dstep = stage == W_NSTAGE - 1 ? -(W_NSTAGE - 1) * W_STAGE : W_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) { bA[i][0] += dstep; bA[i][1] += dstep; }
        bB[0] += dstep; bB[1] += dstep;
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SCALED && u0 == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { fa[0][0][i] *= sc[i]; fa[0][1][i] *= sc[i]; }
      }
      if constexpr (SCALED && u0 == 2 * NACT) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { fa[1][0][i] *= sc[i]; fa[1][1][i] *= sc[i]; }
      }
      mma1(integral_constant<int, u0>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{});
      constexpr int n_ds = lastp ? 6 : (((u0 + 2 < U) ? 4 : 0) + (p == 0 ? 2 : 0));
      if constexpr (lastp) {
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
        readA(0);
        readB(0, nb0);
        readB(1, nb1);
        pre_dma();
        issue(load, stage);
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
      } else {
        {
          constexpr int u2 = u0 + 2, sub2 = u2 / NACT, j2 = J0 + u2 % NACT;
          readB(u2 & 3, (unsigned)((8 * (sub2 >> 1) + (sub2 & 1)) * 2048 + j2 * 256));
        }
        {
          constexpr int u3 = u0 + 3, sub3 = u3 / NACT, j3 = J0 + u3 % NACT;
          readB(u3 & 3, (unsigned)((8 * (sub3 >> 1) + (sub3 & 1)) * 2048 + j3 * 256));
        }
        if constexpr (p == 0) readA(1);
      }
      w_static_for<1, 8>([&](auto xc) __attribute__((always_inline)) {
        constexpr int x = decltype(xc)::value;
        mma1(integral_constant<int, u0>{}, integral_constant<int, x / 4>{}, integral_constant<int, x % 4>{});
      });
      w_static_for<0, 8>([&](auto xc) __attribute__((always_inline)) {
        constexpr int x = decltype(xc)::value;
        mma1(integral_constant<int, u1>{}, integral_constant<int, x / 4>{}, integral_constant<int, x % 4>{});
      });
      // mode 4: the unit that starts k-substep `sub` carries the substep's two mean MFMAs and the read of alpha's row for
      // the next substep (this k-tile's, or the next one's: the tile after a tile starts at k-tile 0 again)
      constexpr int nmean = (EM == 4 && TK >= 0) ? ((u0 % NACT == 0) ? 1 : 0) + ((u1 % NACT == 0) ? 1 : 0) : 0;
      auto mean_work = [&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        if constexpr (EM == 4 && TK >= 0 && u % NACT == 0) {
          constexpr int sub = u / NACT, s8 = sub >> 1, q = sub & 1;
          // (tied accumulators, as in gemm_gram.h: with the builtin the allocator renamed around these two and spilled 17 registers
          //  into the k-tile stream - reloads that wait on the DMA queue; tied: 3, all outside it.  The operands come from LDS
          //  reads, the next use of macc[i] is a k-substep away, the tile's stores wait out the last one: see there.)
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(macc[0]) : "v"(fa[s8][q][0]), "v"(fm));
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(macc[1]) : "v"(fa[s8][q][1]), "v"(fm));
          read_fm(integral_constant<int, (sub < 3 ? TK : ((TK + 1) & 15))>{}, integral_constant<int, ((sub + 1) & 3)>{});
        }
      };
      mean_work(integral_constant<int, u0>{});
      mean_work(integral_constant<int, u1>{});
      if constexpr (!lastp) {
        // first MFMA (it carries the wait), then one LDS read per MFMA, then the rest
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        w_static_for<0, n_ds>([&](auto) __attribute__((always_inline)) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, 15 - n_ds, 0);
        if constexpr (nmean > 0) {
          __builtin_amdgcn_sched_group_barrier(0x100, nmean, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * nmean, 0);
        }
      } else if constexpr (!FENCE) {
        // the k-tile's last pair: first MFMA, then the next k-tile's first fragment reads and the DMA requests of the
        // k-tile three ahead, ONE per MFMA, so that their ~80 scalar / address instructions issue under the remaining
        // MFMAs instead of in front of them (both waves of a SIMD are at this point together: nothing else would feed
        // the matrix pipe meanwhile)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        w_static_for<0, 6>([&](auto) __attribute__((always_inline)) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        });
        w_static_for<0, 6>([&](auto) __attribute__((always_inline)) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        if constexpr (nmean > 0) {
          __builtin_amdgcn_sched_group_barrier(0x100, nmean, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * nmean, 0);
        }
      }
    });
    advance(load);
    stage = stage == W_NSTAGE - 1 ? 0 : stage + 1;
  };
  auto nothing = [&]() __attribute__((always_inline)) {};

  // The stream of k-tiles as straight-line code: per output tile and k block, the dense k-tiles (a loop) and the
  // sixteen k-tiles of the diagonal 256 x 256 block, each with its compile-time column range.  (A switch over the eight
  // variants inside one loop made hipcc give every variant its own accumulator registers and spill 300-600 of them.)
  //
  // Where the block columns of a tile are written (epi_block):
  //   DIR 1 (B upper: k-tile t of the diagonal block touches j >= t/2): in the LAST k block, column j is final after
  //     k-tile 2j+1 and is written inside k-tile 2j+2 (j <= 6); column 7 right after the tile's last k-tile.
  //   DIR 2 (B lower: k-tile t touches j <= t/2), one k block: the tile's column 0 is written right after its last
  //     k-tile; columns j >= 1 stay in their registers and are written inside k-tile 2j-1 of the workgroup's NEXT
  //     tile, which does not touch j before k-tile 2j (`pend`); the last tile of a workgroup flushes them at once.
  //   otherwise (dense B; B lower with several k blocks): the whole tile after its last k-tile.
  constexpr integral_constant<int, 0> I0{};
  constexpr integral_constant<int, 7> I7{};
  // offsets of the first two units of: a dense k-tile / k-tile 0 of a diagonal block
  const unsigned dn0 = 0, dn1 = 256, dg0 = 0, dg1 = DIR == 2 ? 2048u : 256u;
  bool pend = false;              // DIR 2: columns 1..7 of the previous tile (ptm, ptn) are still in their registers
  unsigned ptm = 0, ptn = 0;
  for (;;) {
    const int nq = DIR != 0 ? (int)(cons.tn % Q) : 0;
    WCur ntile_cur = cons;                       // the workgroup's next output tile, if any
    const bool has_next = cons.lin + gridDim.x < ntile;
    if (has_next) { ntile_cur.lin = cons.lin + gridDim.x; setup(ntile_cur); }
    for (int blk = 0; blk < nkb; ++blk) {
      int fob = 1;
      const bool lastblk = blk == nkb - 1;
      if constexpr (DIR == 1) {
        const int nd = 16 * nq;
        for (int t = 0; t < nd; ++t) { ktile(I0, I7, dn0, dn1, fob, blk, nothing, std::false_type{}, integral_constant<int, -1>{}); fob = 0; }
      }
      if constexpr (DIR != 0) {
        w_static_for<0, 16>([&](auto tc) __attribute__((always_inline)) {
          constexpr int t = decltype(tc)::value, v = t >> 1, vn = (t + 1) >> 1;
          // first units of the k-tile after this one: the next diagonal k-tile, or what follows the diagonal block
          unsigned nb0, nb1;
          if constexpr (t < 15) {
            if constexpr (DIR == 1) { nb0 = vn * 256; nb1 = (8 - vn) == 1 ? 2048 + vn * 256 : (vn + 1) * 256; }
            else { nb0 = 0; nb1 = (vn + 1) == 1 ? 2048 : 256; }
          } else {
            if constexpr (DIR == 1) { nb0 = 0; nb1 = 256; }                                // dense or diagonal k-tile 0: same
            else { const bool dense_next = 16 * nq + 16 < ktb; nb0 = 0; nb1 = dense_next ? dn1 : dg1; }
          }
          if constexpr (SCALED && DIR == 1 && t == 15) {
            // the next output tile's scales: written before this k-tile's barrier, read after it (this tile's blocks
            // have their scales in registers since they started)
            if (lastblk && has_next) stage_scales(ntile_cur);
          }
          if constexpr (EM == 0 && DIR == 1 && t == 0) {
            if (lastblk) load_esc();              // (SCALED: the tile's scales are still in LDS here, not after k-tile 15)
          }
          // modes 1-3: which block column is written inside this k-tile (before its DMA requests)
          constexpr int jw = (DIR == 1 && t >= 2 && t % 2 == 0) ? t / 2 - 1 : ((DIR == 2 && t % 2 == 1 && t <= 13) ? (t + 1) / 2 : -1);
          auto pre = [&]() __attribute__((always_inline)) {
            if constexpr (EM != 0 && jw >= 0) {
              if (DIR == 1 ? lastblk : pend) {
                epi_fast(integral_constant<int, (jw >= 0 ? jw : 0)>{}, DIR == 1 ? cons.tm : ptm, DIR == 1 ? cons.tn : ptn, DIR == 2 && t == 13, false);
                // mode 2: the emul values of the NEXT column to be written (two k-tiles from here) are requested now,
                // in front of this k-tile's DMA requests: when they are used, everything older in the in-order vmcnt
                // queue is a DMA request that the barrier wait needs anyway.  (Requested next to their use they made
                // every such k-tile wait for the DMA requests issued just before: 2.7 ms against 2.1 on the old engine.)
                if constexpr (EM == 2 && DIR == 2 && jw < 7) epi_fetch(integral_constant<int, (jw >= 0 && jw < 7 ? jw + 1 : 0)>{}, ptm, ptn);
              }
            }
          };
          // (measured: the interleaved last pair gains 1-2 % on Ct / dCt and loses 5 % on the T product, whose deferred
          //  block-column stores share the vmcnt queue with the DMA requests: that kernel keeps the fenced form throughout)
          constexpr bool has_epi = EM != 0 && (jw >= 0 || DIR == 2);
          if constexpr (DIR == 1) ktile(integral_constant<int, v>{}, I7, nb0, nb1, fob, blk, pre, integral_constant<bool, has_epi>{}, integral_constant<int, (EM == 4 ? t : -1)>{});
          else ktile(I0, integral_constant<int, v>{}, nb0, nb1, fob, blk, pre, integral_constant<bool, has_epi>{}, integral_constant<int, -1>{});
          fob = 0;
        });
        if constexpr (DIR == 2) pend = false;
      }
      if constexpr (DIR != 1) {
        const int t0 = DIR == 2 ? 16 * nq + 16 : 0;
        for (int t = t0; t < ktb; ++t) {
          const bool more = t + 1 < ktb;
          ktile(I0, I7, more ? dn0 : dg0, more ? dn1 : (DIR == 2 ? dg1 : dn1), fob, blk, nothing, std::false_type{}, integral_constant<int, -1>{});
          fob = 0;
        }
      }
    }
    // ---- after the tile's last k-tile (these stores are younger than the DMA requests of that k-tile)
    if constexpr (EM != 0 && DIR == 1) {
      epi_fetch(I7, cons.tm, cons.tn);
      epi_fast(I7, cons.tm, cons.tn, true, true);
      if constexpr (EM == 4) {
        // the tile's mean0 rows: every lane issues its two 8-byte stores (lanes of outputs d >= D aim past the buffer's
        // records and are dropped), so that the store count the vmcnt scheme relies on does not depend on D
        int li2 = lane & 15, lk2 = lane >> 4;
        asm volatile("" : "+v"(li2), "+v"(lk2));
        const int d = 4 * wc + (li2 & 3);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
            g.mean_out + (long)cons.tm * WBM * g.mean_d, 0, WBM * g.mean_d * 8, 0x00020000);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // MFMA results -> stores: the wait states hipcc cannot see (asm MFMAs)
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = (wr * 2 + i) * 16 + 4 * (li2 >> 2) + lk2;
          const unsigned vo_m = d < g.mean_d ? (unsigned)((row * g.mean_d + d) * 8) : 0x7ffffff0u;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, macc[i]), rm, vo_m, 0, 0);
          macc[i] = 0.0;
        }
        st_a += 4; st_b += 4;       // (these two and the two row-sum stores of the last block column)
      }
    } else if constexpr (EM != 0 && DIR == 2) {
      if (has_next) {
        epi_fetch(I0, cons.tm, cons.tn);
        epi_fast(I0, cons.tm, cons.tn, false, true);
        if constexpr (EM == 2) epi_fetch(integral_constant<int, 1>{}, cons.tm, cons.tn);     // for the write inside k-tile 1
        pend = true; ptm = cons.tm; ptn = cons.tn;
      } else {
        w_static_for<0, 8>([&](auto jc) __attribute__((always_inline)) {
          epi_fetch(jc, cons.tm, cons.tn);
          epi_fast(jc, cons.tm, cons.tn, decltype(jc)::value == 7, true);
        });
      }
    } else {
      if constexpr (DIR != 1) load_esc();
      w_static_for<0, 8>([&](auto jc) __attribute__((always_inline)) {
        epi_block(jc, cons.tm, cons.tn, decltype(jc)::value == 7, true);
      });
    }
    if (!has_next) break;
    cons = ntile_cur;
  }
  // the DMA requests issued past the end of the stream land in LDS: retire them before the workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// Host side (gemm_wide.hip): true when `a` (an NN product) can run on the wide kernel
bool gemm_wide_ok(const GemmArgs& a);
bool gemm_wide_mean_ok(const GemmArgs& a);      // `a` runs as mode 4: the launch also writes mean_out = A mean_alpha
hipError_t gemm_wide(hipStream_t st, const GemmArgs& a);

}  // namespace dgp
