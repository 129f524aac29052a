// Weighted Gram kernel: G_d += sum_p s[p][d] c_p c_p^T for d = 0..D-1, lower triangle, Mp = 256.
//   reference: what tf.GradientTape derives for d ELBO / d q_sqrt through  reduce_sum(A_tiled * (SK @ A_tiled))
//   (layers.py:254-263): G_d = sum_p vbar_pd c_p c_p^T, W-bar_d = 2 G_d W_d  (SURVEY App. C step 2).
// Why a kernel of its own next to the 128 x 64 engine's TN product (gemm_f64.h): that one computes the lower triangle at
// 128 x 64 tile granularity (62.5 % of the dense product for a need of 50.2 %), stages the SAME k-tile of Ct twice (as A
// and as B) and splits K into many chunks whose partial tiles are added with atomics.  Here:
//   * one workgroup (8 waves) owns the whole 256 x 256 lower triangle of ONE d at a time and a contiguous range of the
//     points.  The 136 lower 16 x 16 blocks are dealt 17 to a wave (53.1 % of dense, the same MFMA count in every wave)
//     by a compile-time map (GramMap): the six off-diagonal 64 x 64 super-blocks go to six waves, each with one block
//     borrowed from a diagonal super-block that shares a fragment with it; the two other waves own two (reduced)
//     diagonal super-blocks each.  A wave therefore reads 4..8 B fragments and 4..8 A fragments per 68 MFMAs.  The maps
//     differ per wave, so each wave runs its own specialisation (template W) of the k-loop, selected once per launch by
//     a switch on the wave index; the hot loop of a wave is one k-tile (272 MFMAs, < 4 KB of code);
//   * one k-tile of Ct (16 points x 256 columns, 32 KB) is staged ONCE by LDS-DMA and serves as both operands: B fragments
//     as in gemm_wide.h, A fragments (c_p^T: row = column of Ct) by 8-byte reads of the same image; the weights s[p][d] of
//     the k-tile's points travel with it (16 x D doubles) and scale the A fragments;
//   * a ring of four stages filled by LDS-DMA with counted vmcnt + raw s_barrier (gemm_wide.h's scheme), but ONE stage more
//     than the requests in flight need: the barrier of k-tile i (publish k-tile i + 1, request k-tile i + 3 into the
//     stage of k-tile i - 1) sits in the MIDDLE of k-tile i, needs no LDS read to be complete, and is surrounded by MFMA
//     work whose fragments are already on their way - no drain and no burst of first reads behind it;
//   * no atomics: the D * P / 16 k-tiles, d-major, are cut into one contiguous range per workgroup (at most two d each);
//     a workgroup stores its one or two partial triangles (278 KB each, coalesced) and gram_reduce_kernel adds them to
//     G_d in a fixed order - 71 MB of traffic per launch and a reproducible sum.  (Measured: 256 workgroups adding
//     their triangles to the same G_d with fp64 atomics cost 0.23 ms per d, 1.9 of 9.3 ms at D = 8, and 2 GB of atomic
//     traffic per launch for the 128 x 64 engine's 4096-point chunks.)
//   * DU (weighted form): the launch also produces du_d = sum_p mbar[p][d] c_p (the adjoint of mean = c^T u_d, SURVEY App. C
//     step 1: until round 3 a launch of its own that read Ct once more, 2 GB per 10^6 points).  Waves 0 and 1 own the
//     sixteen row blocks between them (their two groups of four): they multiply the RAW A fragments they read anyway by
//     mbar[k][d] - which travels with the k-tile like the weights, requested by wave 1 - on the VALU: 32 FMAs per wave and
//     k-tile beside 272 MFMAs.  The 256 sums go out behind the segment's triangle (a 137th block of the slot).
// Lane maps of v_mfma_f64_4x4x4_4b_f64: gemm_f64.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gemm_wide.h"

namespace dgp {

// The MFMAs are issued through inline asm with the accumulator TIED (dst = srcC).  hipcc's builtin leaves dst free: with
// 136 accumulator values, 272 MFMAs per loop body and every other register in use it renames accumulators, reaches
// the 256-VGPR limit and spills in one or another wave's loop depending on details of the source (a spill reload waits
// vmcnt(0), i.e. drains the DMA queue).  Tied: 199-203 VGPRs, no spill, in every specialisation.  The compiler does not
// see an MFMA in the asm: no hazard handling (the accumulators are only read again after a barrier; dependent MFMAs
// are >= 64 instructions apart) and no MFMA class for sched_group_barrier (units are fenced with sched_barrier).
#ifndef GR_ASM_MFMA
#define GR_ASM_MFMA 1
#endif
#ifndef GR_AHEAD
#define GR_AHEAD 2      // B fragments are read this many units ahead of their MFMAs (ring of four slots: 2 or 3; 3 measured: no change)
#endif

constexpr int GR_STAGE_B = 16 * 256 * 8;        // 32 KB: [16 k][128 chunks of 16 B], chunk ^ ((k >> 1) & 1)
constexpr int GR_STAGE_S = 2048;                // 16 points x up to 8 weights (1 KB) + spare
constexpr int GR_STAGE = GR_STAGE_B + GR_STAGE_S;
constexpr int GR_NSTAGE = 4;

struct GramArgs {
  const double* A;      // two-source form: [P][256] point-major, the ROW operand (G = A^T C); nullptr = C
  const double* C;      // [P][256] point-major (Ct)
  const double* s;      // [P][D] weights (vbar), or nullptr = 1
  double* G;            // [D][256][256]: the lower triangles are ADDED to it (by gram_reduce_kernel)
  double* ws;           // partial triangles: [2 * gridDim.x slots][137 blocks][64 lanes][4]  (GR_SLOT_BYTES each; block 136: du)
  long P;               // points, a multiple of 16
  int D;                // 1..8
  const double* mb;     // DU: [P][D] (mbar), and
  double* du;           // [256][D]: du[m][d] += sum_p mb[p][d] C[p][m]
};
constexpr long GR_SLOT_DOUBLES = 137L * 256;
constexpr long GR_SLOT_BYTES = GR_SLOT_DOUBLES * 8;
// Work split, XCD-grouped for ANY grid size G >= 8 D.  Workgroups are dealt round-robin to the 8 XCDs (b % 8, observed
// placement; speed only): XCD x holds n_x = ceil((G - x) / 8) of them and gets a share n_x / G of the points (k-tiles
// [KT c_x / G, KT c_{x+1} / G), c_x = workgroups on the XCDs before x).  Inside an XCD its KTx * D k-tiles, d-major, are cut
// into n_x equal contiguous ranges: workgroup idx = b / 8 owns [Fx idx / n_x, Fx (idx + 1) / n_x) - at most TWO d (D <= n_x),
// i.e. at most two segments, each with its own partial triangle (slot 2 b + segment).  The D workgroups that stream the
// same points therefore sit on ONE XCD and walk them at the same time (idx, idx + n_x / D, ...: one HBM fetch, D - 1 L2
// hits) whatever G is: the first version cut the global d-major list into G ranges, which kept those readers on one XCD
// only when G was a multiple of 8 D (with 248 of 256 workgroups - eight CUs left to a collective - the launch fetched 62
// instead of 13 GB per three iterations).
struct GramRange { long kt0, ktn; long f_lo, f_hi; };      // the XCD's k-tiles [kt0, kt0 + ktn); the workgroup's local f range
__host__ __device__ inline GramRange gram_range(long KT, int D, unsigned b, unsigned G) {
  const unsigned x = b % 8u, idx = b / 8u;
  const unsigned q = G / 8u, r = G % 8u;                  // XCD y holds q + (y < r) workgroups
  const unsigned n_x = q + (x < r ? 1u : 0u);
  const unsigned c_x = x * q + (x < r ? x : r);
  GramRange g;
  g.kt0 = (KT * (long)c_x) / (long)G;
  g.ktn = (KT * (long)(c_x + n_x)) / (long)G - g.kt0;
  const long Fx = g.ktn * D;
  g.f_lo = (Fx * (long)idx) / (long)n_x;
  g.f_hi = (Fx * (long)(idx + 1)) / (long)n_x;
  return g;
}


// A wave's 17 blocks, in one or two GROUPS: group q has row blocks rows[q][0..nr[q]) and column blocks
// cols[q][0..nc[q]); bit ri of mask[q][ci] says whether block (rows[q][ri], cols[q][ci]) belongs to the wave.  The A
// fragments (one per row block) are loaded group by group, one group ahead of their use.
struct GramMap {
  int ng;
  int nr[2], nc[2];
  int rows[2][5], cols[2][5];
  unsigned mask[2][5];
};

constexpr GramMap gram_map(int w) {
  GramMap m{};
  if (w < 2) {
    // two diagonal 64 x 64 super-blocks, (2w, 2w) and (2w+1, 2w+1), lower parts, minus the blocks lent to other waves
    m.ng = 2;
    for (int q = 0; q < 2; ++q) {
      m.nr[q] = 4; m.nc[q] = 4;
      for (int i = 0; i < 4; ++i) { m.rows[q][i] = 8 * w + 4 * q + i; m.cols[q][i] = 8 * w + 4 * q + i; m.mask[q][i] = 0xfu & ~((1u << i) - 1); }
    }
    if (w == 0) { m.mask[0][0] &= ~8u; m.mask[0][1] &= ~8u; m.mask[0][2] &= ~8u; }     // (3,0) (3,1) (3,2)
    else { m.mask[0][0] &= ~8u; m.mask[1][0] &= ~8u; m.mask[1][1] &= ~8u; }            // (11,8) (15,12) (15,13)
    return m;
  }
  // off-diagonal super-block (I, J) + one borrowed block
  const int SI[6] = {1, 2, 3, 2, 3, 3}, SJ[6] = {0, 0, 0, 1, 1, 2};
  const int I = SI[w - 2], J = SJ[w - 2];
  m.ng = 1;
  for (int i = 0; i < 4; ++i) { m.rows[0][i] = 4 * I + i; m.cols[0][i] = 4 * J + i; m.mask[0][i] = 0xf; }
  if (w <= 4) {            // borrowed (3, w - 2): a fifth row, in a column the wave already reads: a group of its own
    m.ng = 2;              // (first, so that the k-tile does not end with it), which keeps the fragment set at four rows
    m.nr[1] = 4; m.nc[1] = 4;
    for (int i = 0; i < 4; ++i) { m.rows[1][i] = m.rows[0][i]; m.cols[1][i] = m.cols[0][i]; m.mask[1][i] = 0xf; }
    m.nr[0] = 1; m.nc[0] = 1;
    m.rows[0][0] = 3; m.cols[0][0] = w - 2; m.mask[0][0] = 1u;
  } else {                 // borrowed (11, 8), (15, 12), (15, 13): a fifth column (read FIRST: the short unit is not the
    m.nr[0] = 4; m.nc[0] = 5;     // one the k-tile ends with), in the wave's last row
    for (int i = 4; i > 0; --i) { m.cols[0][i] = m.cols[0][i - 1]; m.mask[0][i] = m.mask[0][i - 1]; }
    m.cols[0][0] = w == 5 ? 8 : (w == 6 ? 12 : 13);
    m.mask[0][0] = 1u << 3;
  }
  return m;
}

constexpr int gr_popc(unsigned x) { int n = 0; while (x) { n += x & 1; x >>= 1; } return n; }
constexpr int gram_units(const GramMap& m) { return m.nc[0] + (m.ng > 1 ? m.nc[1] : 0); }       // per k-substep
constexpr int gram_blk(const GramMap& m, int q, int ci, int ri) {      // index of a block among the wave's 17
  int n = 0;
  for (int qq = 0; qq < m.ng; ++qq)
    for (int c = 0; c < m.nc[qq]; ++c) {
      if (qq == q && c == ci) return n + gr_popc(m.mask[qq][c] & ((1u << ri) - 1));
      n += gr_popc(m.mask[qq][c]);
    }
  return n;
}
constexpr int gram_nblk(const GramMap& m) { return gram_blk(m, m.ng, 0, 0); }

template <int W> struct GramMapOf { static constexpr GramMap m = gram_map(W); };

template <int W, bool SC, int KH, bool DU>
__device__ __forceinline__ void gram_wave(const GramArgs& g, char* smem, int wave, int lane) {
  using GM = GramMapOf<W>;
  constexpr int NG = GM::m.ng, NS = gram_units(GM::m);
  constexpr int NRM = GM::m.nr[0] > GM::m.nr[NG - 1] ? GM::m.nr[0] : GM::m.nr[NG - 1];
  constexpr int NB = gram_nblk(GM::m);
  static_assert(NB == 17, "every wave owns 17 blocks");
  constexpr int U = 4 * NS;                       // units per k-tile: (substep, group, column index)
  // KH = 1: one source (A = C), a stage = 16 points.  KH = 2: two sources (rows from g.A, columns from g.C), a stage =
  // 8 points of both (two 16 KB images): the loop body still covers 16 points, as two halves with a barrier each.
  constexpr int SUBH = 4 / KH;                    // k-substeps per stage
  constexpr int IMG_B = KH == 2 ? 16384 : 0;      // the column operand's image inside a stage (the row operand's is at 0)
  static_assert(!(SC && KH == 2), "the weighted form has one source");
  static_assert(!DU || SC, "du rides on the weighted form");
  constexpr bool DUW = DU && W < 2;               // this wave accumulates du for its eight row blocks
  static_assert(!DUW || (NG == 2 && GM::m.nr[0] == 4 && GM::m.nr[1] == 4), "waves 0 / 1: two groups of four row blocks");
  static_assert(U % 4 == 0 && GM::m.nc[NG - 1] >= 2, "slot ring / barrier placement");
  const int li = lane & 15, lk = lane >> 4;
  const int sw = lk & 1;
  // (all fragment addresses include the current stage: they are advanced at every k-tile barrier)
  // B fragment of column block c, substep (s8, q): k = 8 s8 + 2 lk + q, 32 contiguous bytes at columns 16 c + 4 (li & 3),
  // the two 16-byte halves swapped when (k >> 1) & 1 = lk & 1
  unsigned bB0 = (unsigned)(IMG_B + (2 * lk) * 2048 + (li & 3) * 32 + sw * 16), bB1 = (unsigned)(IMG_B + (2 * lk) * 2048 + (li & 3) * 32 + (1 - sw) * 16);
  // A fragment of row block r: element (k, m = 16 r + li) of the same image: chunk m / 2 = 8 r + (li >> 1), swizzled
  // (bit 0 only), + (m & 1) * 8: ONE per-lane address, the row block is a compile-time offset of 128 r bytes
  unsigned bA = (unsigned)((2 * lk) * 2048 + (((li >> 1) ^ sw) << 4) + (li & 1) * 8);

  const long KT = g.P / 16;
  const GramRange gr = gram_range(KT, g.D, blockIdx.x, gridDim.x);
  const long f_lo = gr.f_lo, f_hi = gr.f_hi;
  long kt_lo = 0;             // current segment: k-tiles [kt_lo, kt_lo + nkt) of output d
  int nkt = 0;

  constexpr bool extra = (SC && W == 0) || (DU && W == 1);      // this wave also requests the weights' (wave 0) / mbar's (wave 1) piece
  // DMA: the B image = 32 pieces of 1 KB (half a k row each): wave w issues pieces 4 w .. 4 w + 3 (k rows 2 w, 2 w + 1);
  // the weights (16 x D doubles, contiguous, D <= 8) = 1 piece, requested by wave 0
  // KH = 2: each image = 16 pieces (8 k rows): wave w requests row w of both
  const unsigned offB = (unsigned)((lane ^ (KH == 2 ? (wave >> 1) & 1 : wave & 1)) * 16);
  const double* srcA = KH == 2 ? g.A : g.C;
  auto issue = [&](long kt, int h, int stage) __attribute__((always_inline)) {
    char* base = smem + stage * GR_STAGE;
    if constexpr (KH == 1) {
      const double* ub = g.C + (kt * 16 + wave * 2) * 256;
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(ub), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
      for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, W_LDS3(base + (wave * 4 + p) * 1024), 16, offB, (p >> 1) * 2048 + (p & 1) * 1024, 0, 0);
    } else {
      const long row = kt * 16 + h * 8 + wave;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(srcA + row * 256), 0, 0x7ffffff0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.C + row * 256), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, W_LDS3(base + (wave * 2 + p) * 1024), 16, offB, p * 1024, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, W_LDS3(base + IMG_B + (wave * 2 + p) * 1024), 16, offB, p * 1024, 0, 0);
      }
    }
    if constexpr (extra) {
      {
        // rows 16 kt .. 16 kt + 15 of s: 16 D doubles from s + 16 kt D; reads past the end of s return zeros (num_records)
        const long off = kt * 16 * g.D * 8, left = g.P * g.D * 8 - off;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(W == 0 ? g.s : g.mb) + off), 0, (int)(left > 0x7ffffff0 ? 0x7ffffff0 : (left > 0 ? left : 0)), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, W_LDS3(base + GR_STAGE_B + (W == 0 ? 0 : 1024)), 16, (unsigned)(lane * 16), 0, 0, 0);
      }
    }
  };
  // wait until at most one / two of this wave's k-tile requests are outstanding (4 or 5 DMA instructions each)
  auto wait_tiles1 = [&]() __attribute__((always_inline)) {
    if (extra) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  };
  auto wait_tiles2 = [&]() __attribute__((always_inline)) {
    if (extra) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };
  auto kt_of = [&](int i) { return kt_lo + (i < nkt ? i : nkt - 1); };   // requests past the range re-read its last k-tile
  // stage x (counted from the segment's start) -> its 16-point k-tile and half
  auto issue_x = [&](int x, int stage) __attribute__((always_inline)) { issue(kt_of(x / KH), x % KH, stage); };

  double acc[NB][4];
  double fb[4][4];              // ring of four units: [slot][e]
  double fr[NRM], fa[NRM];      // A fragments of a group: raw (read one phase ahead) and scaled
  double fw = 1.0;
  double fm = 0.0, dacc[2][4];  // DUW: mbar[k][d] of the phase, the du partial sums of the wave's row blocks (k = 2 lk + q mod 8 only)

  auto readB = [&](int slot, unsigned off) __attribute__((always_inline)) {
    const d2_t v0 = *reinterpret_cast<const d2_t*>(smem + bB0 + off);
    const d2_t v1 = *reinterpret_cast<const d2_t*>(smem + bB1 + off);
    fb[slot][0] = v0[0]; fb[slot][1] = v0[1]; fb[slot][2] = v1[0]; fb[slot][3] = v1[1];
  };
  // unit u of a k-tile -> substep, group, column index
  auto u_sub = [](int u) constexpr { return u / NS; };
  auto u_grp = [](int u) constexpr { return (u % NS) < GM::m.nc[0] ? 0 : 1; };
  auto u_ci = [](int u) constexpr { return (u % NS) < GM::m.nc[0] ? (u % NS) : (u % NS) - GM::m.nc[0]; };
  auto uoff = [&](int u) constexpr {
    const int sub = u_sub(u) % SUBH;
    return (unsigned)((8 * (sub >> 1) + (sub & 1)) * 2048 + GM::m.cols[u_grp(u)][u_ci(u)] * 128);
  };

  int seg = 0;
  for (long f = f_lo; f < f_hi; f += nkt, ++seg) {
    const int d = (int)(f / gr.ktn);
    const long koff = f - (long)d * gr.ktn;             // position inside the XCD's share of the points
    kt_lo = gr.kt0 + koff;
    nkt = (int)((gr.ktn - koff) < (f_hi - f) ? (gr.ktn - koff) : (f_hi - f));
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[b][e] = 0.0;
    if constexpr (DUW) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) dacc[q][ri] = 0.0;
    }
    unsigned wbase = (unsigned)(GR_STAGE_B + (2 * lk * g.D + d) * 8);
    const unsigned wstep = (unsigned)(g.D * 8);
    // raw A fragments (and the weight) of phase ph = substep * NG + group
    auto rawA = [&](auto phc) __attribute__((always_inline)) {
      constexpr int ph = decltype(phc)::value, sub = (ph / NG) % SUBH, q = ph % NG;
      constexpr unsigned ko = (unsigned)((8 * (sub >> 1) + (sub & 1)) * 2048);
#pragma unroll
      for (int ri = 0; ri < GM::m.nr[q]; ++ri) fr[ri] = *reinterpret_cast<const double*>(smem + bA + (ko + GM::m.rows[q][ri] * 128));
      if constexpr (SC) fw = *reinterpret_cast<const double*>(smem + wbase + (unsigned)(8 * (sub >> 1) + (sub & 1)) * wstep);
      if constexpr (DUW) fm = *reinterpret_cast<const double*>(smem + wbase + 1024u + (unsigned)(8 * (sub >> 1) + (sub & 1)) * wstep);
    };

    // ---- prologue: three k-tiles requested, the first one landed and published
    issue_x(0, 0);
    issue_x(1, 1);
    issue_x(2, 2);
    wait_tiles2();
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    rawA(std::integral_constant<int, 0>{});
    readB(0, uoff(0));
    readB(1, uoff(1));
#if GR_AHEAD == 3
    readB(2, uoff(2));
#endif

    for (int i = 0; i < nkt; ++i) {
      // address step out of the stage of half 0 / half 1 of this iteration
      const int dstep0 = stage == GR_NSTAGE - 1 ? -(GR_NSTAGE - 1) * GR_STAGE : GR_STAGE;
      const int dstep1 = ((stage + 1) & (GR_NSTAGE - 1)) == GR_NSTAGE - 1 ? -(GR_NSTAGE - 1) * GR_STAGE : GR_STAGE;
      // unit u = (substep, group, column index): its B fragment sits in slot u & 3 and was read two units earlier
      w_static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, sub = u_sub(u), q = u_grp(u), ci = u_ci(u), ph = sub * NG + q;
        constexpr bool ph_start = ci == 0;
        if constexpr (u % (SUBH * NS) == (SUBH / 2) * NS) {
          constexpr int h = u / (SUBH * NS);
          // (first in its unit: with one group per wave this unit also starts the stage's last phase, which reads the
          //  NEXT stage's first A fragments - only published by this barrier)
          // mid k-tile: this wave's pieces of the NEXT k-tile have landed (the one after it may stay in flight); the
          // barrier publishes the next k-tile and certifies that every wave has left the PREVIOUS k-tile, whose stage
          // takes the request for the k-tile three ahead.  (No LDS read has to be complete here: the current stage is
          // not overwritten before the next barrier.)
          __builtin_amdgcn_sched_barrier(0);
          wait_tiles1();
#ifndef GR_ABL_NOBAR
          __builtin_amdgcn_s_barrier();
#endif
          __builtin_amdgcn_sched_barrier(0);
#ifndef GR_ABL_NODMA
          issue_x(KH * i + h + 3, (stage + h + 3) & (GR_NSTAGE - 1));
#endif
        }
        if constexpr (ph_start) {
#pragma unroll
          for (int ri = 0; ri < GM::m.nr[q]; ++ri) fa[ri] = SC ? fr[ri] * fw : fr[ri];
          if constexpr (DUW) {
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) dacc[q][ri] += fr[ri] * fm;
          }
          // raw fragments of the next phase; the k-tile's last phase reads those of the NEXT k-tile's first one (its
          // stage was published by this k-tile's barrier)
          if constexpr ((ph + 1) % (SUBH * NG) != 0) {
            rawA(std::integral_constant<int, ph + 1>{});
          } else {
            { const int ds = (ph / (SUBH * NG)) == 0 ? dstep0 : dstep1; bA += ds; wbase += ds; }
            rawA(std::integral_constant<int, (ph + 1) % (4 * NG)>{});
          }
          // (fence: left to itself the scheduler sinks these reads to the last LDS slots before their first use, which
          //  for the k-tile's last phase is the top of the next iteration)
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (u % (SUBH * NS) == SUBH * NS - GR_AHEAD) {
          // from here on the B fragment reads belong to the next stage
          { const int ds = (u / (SUBH * NS)) == 0 ? dstep0 : dstep1; bB0 += ds; bB1 += ds; }
        }
        readB((u + GR_AHEAD) & 3, uoff((u + GR_AHEAD) % U));
        w_static_for<0, GM::m.nr[q]>([&](auto rc) __attribute__((always_inline)) {
          constexpr int ri = decltype(rc)::value;
          if constexpr ((GM::m.mask[q][ci] >> ri) & 1u) {
            constexpr int b = gram_blk(GM::m, q, ci, ri);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#if GR_ASM_MFMA
              double c_ = acc[b][e];
              const double a_ = fa[ri], b_ = fb[u & 3][e];
              asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c_) : "v"(a_), "v"(b_));
              acc[b][e] = c_;
#else
              acc[b][e] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[ri], fb[u & 3][e], acc[b][e], 0, 0, 0);
#endif
            }
          }
        });
#if GR_ASM_MFMA
        __builtin_amdgcn_sched_barrier(0);      // (units stay in program order: fragment reads, then the unit's MFMAs)
#else
        {
          // issue order inside the unit: its first MFMA (it carries the wait for the fragments), then one LDS read per MFMA
          constexpr int nm = 4 * gr_popc(GM::m.mask[q][ci]);
          constexpr int n_ds = 2;
          constexpr int n_vm = (u % (SUBH * NS) == (SUBH / 2) * NS) ? 4 : 0;                 // (the fifth request of waves 0 / 1 floats)
          constexpr int n_il = n_ds + n_vm < nm - 1 ? n_ds + n_vm : nm - 1;
          constexpr int n_il_vm = n_vm < n_il ? n_vm : n_il, n_il_ds = n_il - n_il_vm;
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          w_static_for<0, n_il_vm>([&](auto) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          });
          w_static_for<0, n_il_ds>([&](auto) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          });
          if constexpr (n_vm > n_il_vm) __builtin_amdgcn_sched_group_barrier(0x020, n_vm - n_il_vm, 0);
          if constexpr (n_ds > n_il_ds) __builtin_amdgcn_sched_group_barrier(0x100, n_ds - n_il_ds, 0);
          if constexpr (nm - 1 - n_il > 0) __builtin_amdgcn_sched_group_barrier(0x008, nm - 1 - n_il, 0);
        }
#endif
      });
      __builtin_amdgcn_sched_barrier(0);
      stage = (stage + KH) & (GR_NSTAGE - 1);
    }
    // ---- flush: this segment's partial lower triangle, block by block (canonical block id r (r + 1) / 2 + c), each
    //      lane's four values contiguous: acc[b][e] of lane (li, lk) is G[16 r + 4 (li >> 2) + lk][16 c + 4 (li & 3) + e]
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // (the requests past the range)
    __builtin_amdgcn_s_barrier();
#if GR_ASM_MFMA
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");               // MFMA results -> stores: the wait states hipcc cannot see
#endif
    {
      const int back = stage * GR_STAGE;       // the next segment starts in stage 0 again
      bA -= back; bB0 -= back; bB1 -= back;
    }
    double* __restrict__ slot = g.ws + ((long)blockIdx.x * 2 + seg) * GR_SLOT_DOUBLES + lane * 4;
    w_static_for<0, NS>([&](auto xc) __attribute__((always_inline)) {
      constexpr int x = decltype(xc)::value, q = u_grp(x), ci = u_ci(x);
      w_static_for<0, GM::m.nr[q]>([&](auto rc) __attribute__((always_inline)) {
        constexpr int ri = decltype(rc)::value;
        if constexpr ((GM::m.mask[q][ci] >> ri) & 1u) {
          constexpr int b = gram_blk(GM::m, q, ci, ri);
          constexpr int r = GM::m.rows[q][ri], c = GM::m.cols[q][ci];
          const d4_t v = {acc[b][0], acc[b][1], acc[b][2], acc[b][3]};
          *reinterpret_cast<d4_t*>(slot + (r * (r + 1) / 2 + c) * 256) = v;
        }
      });
    });
    if constexpr (DUW) {
      // du partial of column m = 16 r + li: the four lk groups hold the k = 2 lk + q (mod 8) shares
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          double v = dacc[q][ri];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          if (lk == 0) slot[136 * 256 - lane * 4 + 16 * GM::m.rows[q][ri] + li] = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

template <bool SC, int KH, bool DU = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_gram_kernel(GramArgs g) {
  __shared__ __attribute__((aligned(1024))) char smem[GR_NSTAGE * GR_STAGE];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  switch (wave) {
    case 0: gram_wave<0, SC, KH, DU>(g, smem, wave, lane); break;
    case 1: gram_wave<1, SC, KH, DU>(g, smem, wave, lane); break;
    case 2: gram_wave<2, SC, KH, DU>(g, smem, wave, lane); break;
    case 3: gram_wave<3, SC, KH, DU>(g, smem, wave, lane); break;
    case 4: gram_wave<4, SC, KH, DU>(g, smem, wave, lane); break;
    case 5: gram_wave<5, SC, KH, DU>(g, smem, wave, lane); break;
    case 6: gram_wave<6, SC, KH, DU>(g, smem, wave, lane); break;
    default: gram_wave<7, SC, KH, DU>(g, smem, wave, lane); break;
  }
}

// host side: gemm_gram_ok / gemm_gram / gemm_gram_ws_bytes (gemm_gram.hip, declared in gemm_f64.h)

}  // namespace dgp
