// Tall-tile kernel for the forward T product  T_d = Ct * W_d  (+ |t_d|^2)  with W_d lower triangular, Mp = 256
//   (layers.py:254-263, SK @ A in triangular form; the product gemm_wide.h runs as <DIR 2, EM 1>).
// Why: per k-tile the wide-tile kernel costs its MFMA cycles + ~1050 (NOTES.md par. 10).  Its 128 x 256 tile walks a
// triangular 256 x 256 block in 16 k-tiles of 1..8 column-block PAIRS (144 of 256 block-steps executed for 136 needed).
// Here a tile is 256 rows x 128 columns: eight waves stacked over the rows, every wave across all eight 16-column blocks
// of the tile, so that a k-tile t needs exactly the blocks cb <= t: the left half of a triangular block is 16 k-tiles of
// 1, 2, .., 8, 8, .., 8 blocks, the right half 8 k-tiles of 1..8 - 24 k-tiles with 4352 MFMAs per wave for 256 rows
// (per 128 rows: 12 k-tiles instead of 16, 136 block-steps instead of 144).  Same LDS stage (A image 32 KB + B image
// 16 KB), same six DMA requests per wave and k-tile, same ring of three stages and end-of-k-tile barrier as
// gemm_wide.h; MFMAs with tied accumulators as in gemm_gram.h.
// The 24 k-tiles of a (row tile, d) pair are straight-line code (compile-time block ranges).  Block column j of a
// finished 128-column tile stays in its registers and is written (with its part of the row sums of squares) inside
// k-tile j-1 of the NEXT tile, which does not touch block j before its k-tile j; block column 0 right after the tile's
// last k-tile (the scheme of gemm_wide.h's lower form).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gemm_wide.h"

namespace dgp {

constexpr int TL_A = 256 * 16 * 8;      // 32 KB: [256 rows][8 chunks of 16 B], chunk ^ ((row >> 1) & 7)
constexpr int TL_B = 16 * 128 * 8;      // 16 KB: [16 k][64 chunks of 16 B], chunk ^ ((k >> 1) & 1)
constexpr int TL_STAGE = TL_A + TL_B;   // 48 KB
constexpr int TL_NSTAGE = 3;

struct TallArgs {
  const double* A; long lda;     // Ct [M][lda]
  const double* B; long ldb;     // Wcat [256][ldb], ldb = D * 256: D lower-triangular 256 x 256 blocks side by side
  double* C; long ldc;           // T, blocked: [M / 128][D * 16][128][16] (ldc = D * 256), or nullptr (row sums only)
  double* rowsq; long rowsq_ld;  // planes [2 D][rowsq_ld]: plane 2 d + h = sum of squares over columns 128 h .. 128 h + 127 of block d
  long M;                        // rows, a multiple of 128 (the last tile may be half empty)
  int D;
  int rowmajor;                  // T as a plain [M][ldc] array (read back by gemm_dcpanel.h) instead of the blocked layout
#ifdef TL_TIMING
  unsigned* ts;                  // (diagnostic build, tools/tall_bench only) [workgroup][TL_TS_PER_WG]: per position of every pair, shader
                                 // clocks from the previous barrier to this position's wait, and inside the wait + barrier
#endif
};
#ifdef TL_TIMING
constexpr int TL_TS_PER_WG = 160 * 48;
#endif

// position p = 0..23 inside a (row tile, d) pair: p < 16: left half (h = 0), k-tile t = p, blocks 0..min(p, 7);
// p >= 16: right half (h = 1), k-tile t = p - 8, blocks 0..p - 16
constexpr int tl_h(int p) { return p < 16 ? 0 : 1; }
constexpr int tl_t(int p) { return p < 16 ? p : p - 8; }
constexpr int tl_nact(int p) { return p < 16 ? (p < 7 ? p : 7) + 1 : p - 15; }
// Stores of the deferred block columns, per position: 4 in front of the position's DMA request (positions 0..6: the
// previous right half's columns 1..7; 16..22: this pair's left half's), 4 behind it at the end of positions 15 and 23
// (column 0 of the half just finished).  vmcnt counts loads and stores in issue order: the wait of position p (for
// the requests of position p + 1, issued at p - 2) must allow for everything younger - the requests of p + 2 (6) and
// the stores issued since.  All compile-time: the first pair issues the same (zero) stores as any other.
constexpr int tl_pre(int p) { return (p <= 6 || (p >= 16 && p <= 22)) ? 4 : 0; }
constexpr int tl_post(int p) { return (p == 15 || p == 23) ? 4 : 0; }
#ifndef TL_STORE_SPREAD
#define TL_STORE_SPREAD 0
#endif
#if TL_STORE_SPREAD
// Spread form (NOT the default - measured slower): the four stores of a deferred column issued one by one BETWEEN the MFMA units of
// its position, in front of the position's wait.  Why it was tried: the clock records of the diagnostic build (TL_TIMING) show ~800
// clocks per storing position in which all eight waves queue their stores behind the barrier and none issues an MFMA.  What it
// did: 170 k instead of 177 k clocks per pair - and 10.18 instead of 9.69 ms, because with the stores the kernel runs at the
// package power limit (1355 W) and the spread form is clocked at 2.10 instead of 2.28 GHz there (profiles/r4_power.txt, NOTES.md
// 13.6).  Program order of a position p:
// S(p) .. wait(p) .. barrier .. request(p + 3) .. [column 0's stores P(p) behind positions 15 and 23]; younger than the requests
// wait(p) is for (issued at p - 2): P(p - 2), S(p - 1), the requests of p - 1, P(p - 1), S(p).
constexpr int tl_allow(int p) {
  return 6 + tl_post((p + 22) % 24) + tl_pre((p + 23) % 24) + tl_post((p + 23) % 24) + tl_pre(p);
}
#else
constexpr int tl_allow(int p) { return 6 + tl_post((p + 22) % 24) + tl_pre((p + 23) % 24) + tl_post((p + 23) % 24); }
#endif

// NT: the T stores carry the nontemporal hint (cache policy nt): 16 GB per 10^6 rows at D = 8 that nothing re-reads before the
// backward pass would otherwise stream through the same 4 MB L2 that holds the Ct panel (read by the D pairs of a row tile)
// and the W_d tiles
template <bool STORE, bool NT = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_tall_kernel(TallArgs g) {
  __shared__ __attribute__((aligned(1024))) char smem[TL_NSTAGE * TL_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4, sw = lk & 1;
  const unsigned tiles_m = (unsigned)((g.M + 255) / 256), ntile = tiles_m * (unsigned)g.D;
  if (blockIdx.x >= ntile) return;

  // (row tile, d) of the workgroup's r-th pair: one XCD's L2 sees a contiguous run of pairs (gemm_wide.h)
  auto locate = [&](unsigned lin, unsigned& tm, unsigned& d) __attribute__((always_inline)) {
    const unsigned q = ntile / 8u, r = ntile % 8u, xcd = lin % 8u, idx = lin / 8u;
    const unsigned bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tm = bid / (unsigned)g.D;
    d = bid - tm * (unsigned)g.D;
  };

  // ---- DMA.  A piece P = 4 w + p4 (8 rows x 128 B): lane -> row 8 P + lane / 8, physical chunk lane % 8, logical chunk =
  // physical ^ ((row >> 1) & 7) = (lane & 7) ^ ((4 p4 + lane / 16) & 7).  B piece P = 2 w + p2 (one k row, 1 KB): lane ->
  // physical chunk lane, logical = lane ^ ((k >> 1) & 1) = lane ^ (w & 1).
  unsigned offA[4];
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) {
    const int c = (lane & 7) ^ ((4 * p4 + (lane >> 4)) & 7);
    offA[p4] = (unsigned)(((long)(lane >> 3) * g.lda + 2 * c) * 8);
  }
  const unsigned offB = (unsigned)((lane ^ (wave & 1)) * 16);
  const unsigned stepA = (unsigned)(8 * g.lda * 8), stepB = (unsigned)(g.ldb * 8);
  const long a_bytes = g.M * g.lda * 8;
  // Per pair: this wave's A rows and B rows / columns as 64-bit bases and the bytes of A left behind the base (rows past
  // M - the second half of an odd last tile - are beyond the resource's range and return zeros); per position only
  // compile-time offsets are added.
  struct Src { const char* a; const char* b; long left; };
  auto src_of = [&](unsigned tm, unsigned d) __attribute__((always_inline)) {
    Src r;
    const long aoff = ((long)tm * 256 + wave * 32) * g.lda * 8;
    r.a = reinterpret_cast<const char*>(g.A) + aoff;
    r.left = a_bytes - aoff;
    r.b = reinterpret_cast<const char*>(g.B) + ((long)wave * 2 * g.ldb + (long)d * 256) * 8;
    return r;
  };
  auto issue = [&](auto lpc, const Src& sc, int stage) __attribute__((always_inline)) {
    constexpr int lp = decltype(lpc)::value, h = tl_h(lp), t = tl_t(lp);
    const long left = sc.left - 16 * t * 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(left > 0 ? sc.a + 16 * t * 8 : sc.a), 0, (int)(left > 0x7ffffff0 ? 0x7ffffff0 : (left > 0 ? left : 0)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sc.b + ((long)16 * t * g.ldb + 128 * h) * 8), 0, 0x7ffffff0,
                                                                        0x00020000);
    char* base = smem + stage * TL_STAGE;
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, W_LDS3(base + (wave * 4 + p4) * 1024), 16, offA[p4], p4 * stepA, 0, 0);
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, W_LDS3(base + TL_A + (wave * 2 + p2) * 1024), 16, offB, p2 * stepB, 0, 0);
  };

  // ---- fragment addresses (stage included; advanced at every k-tile barrier)
  unsigned bA[2][2], bB0, bB1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 32 + i * 16 + li;
#pragma unroll
    for (int s8 = 0; s8 < 2; ++s8) bA[i][s8] = (unsigned)(row * 128 + (((s8 * 4 + lk) ^ ((row >> 1) & 7)) << 4));
  }
  bB0 = (unsigned)(TL_A + (2 * lk) * 1024 + (li & 3) * 32 + sw * 16);
  bB1 = (unsigned)(TL_A + (2 * lk) * 1024 + (li & 3) * 32 + (1 - sw) * 16);

  double acc[2][8][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#ifdef TL_BUILTIN_MFMA
        acc[i][j][e] = 0.0;
#else
        double z_;
        asm volatile("v_mov_b64 %0, 0" : "=v"(z_));      // (explicit: see flush)
        acc[i][j][e] = z_;
#endif
      }
  double fa[2][2][2];       // [s8][q][i]
  double fb[4][4];          // ring of four units: [slot][e]
  auto readA = [&](int s8) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const d2_t v = *reinterpret_cast<const d2_t*>(smem + bA[i][s8]);
      fa[s8][0][i] = v[0];
      fa[s8][1][i] = v[1];
    }
  };
  auto readB = [&](int slot, unsigned off) __attribute__((always_inline)) {   // off: (8 s8 + q) * 1024 + j * 128
    const d2_t v0 = *reinterpret_cast<const d2_t*>(smem + bB0 + off);
    const d2_t v1 = *reinterpret_cast<const d2_t*>(smem + bB1 + off);
    fb[slot][0] = v0[0]; fb[slot][1] = v0[1]; fb[slot][2] = v1[0]; fb[slot][3] = v1[1];
  };
  auto uoff = [](int p, int u) constexpr {
    const int n = tl_nact(p), sub = u / n, j = u % n;
    return (unsigned)((8 * (sub >> 1) + (sub & 1)) * 1024 + j * 128);
  };

  // ---- epilogue of block column j of the 128-column tile (tm, d, h).  acc[i][j][e] of lane (li, lk) is
  //      T[256 tm + 32 w + 16 i + 4 (li >> 2) + lk][256 d + 128 h + 16 j + 4 (li & 3) + e]
  double rsum[2] = {0.0, 0.0};
  unsigned vo[2];
  if (g.rowmajor) {
#pragma unroll
    for (int i = 0; i < 2; ++i) vo[i] = (unsigned)(((long)(i * 16 + 4 * (li >> 2) + lk) * g.ldc + 4 * (li & 3)) * 8);
  } else {
#pragma unroll
  for (int i = 0; i < 2; ++i) vo[i] = (unsigned)((((wave & 3) * 32 + i * 16 + 4 * (li >> 2) + lk) * 16 + 4 * (li & 3)) * 8);
  }
  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  // PART -1: the whole epilogue of the column at once.  PART 0..3 (spread form): store PART alone (row block PART / 2, 16-byte half
  // PART % 2); the sums of squares (and, for the half's last column, the row sums' own stores) ride on part 0, the accumulators
  // restart from zero behind part 3.
  auto flush = [&](auto jc, unsigned tm, unsigned d, int h, bool last_block, auto partc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value, PART = decltype(partc)::value;
    // rows of a half-empty last tile are neither stored nor summed: their stores go through a resource of zero records
    const int nrec = ((long)tm * 256 + wave * 32 < g.M) ? 0x7ffffff0 : 0;
    if constexpr (PART <= 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) rsum[i] += acc[i][j][e] * acc[i][j][e];
#ifndef TL_DBG_NORS
      if (last_block) {
        int li2 = lane & 15, lk2 = lane >> 4;
        asm volatile("" : "+v"(li2), "+v"(lk2));
        double* rs = g.rowsq + ((long)d * 2 + h) * g.rowsq_ld + (long)tm * 256 + wave * 32;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rs, 0, nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          double t = rsum[i];
          t += __shfl_xor(t, 1);
          t += __shfl_xor(t, 2);
          // (the four lanes of a quad hold the same sum and write it to the same address: no divergence)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, t), rr, (unsigned)((i * 16 + 4 * (li2 >> 2) + lk2) * 8), 0, 0);
          rsum[i] = 0.0;
        }
      }
#endif
    }
    if constexpr (STORE) {
      // blocked T: panel (2 tm + w / 4) of 128 rows, 16-column block 16 d + 8 h + j; the block offset sits in the
      // resource base, the instruction's scalar offset stays 0 (gemm_wide.h: store / data-register hazard).  The
      // stores are ALWAYS issued (the vmcnt bookkeeping is compile-time): a wave whose rows lie past M gets a resource of
      // zero records, which drops them
      const double* cb = g.rowmajor ? g.C + ((long)tm * 256 + wave * 32) * g.ldc + (long)d * 256 + 128 * h + 16 * j
                                    : g.C + ((long)tm * 2 + (wave >> 2)) * 128 * g.ldc + ((long)d * 16 + 8 * h + j) * 2048;
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(cb), 0, nrec, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const d2_t v0 = {acc[i][j][0], acc[i][j][1]}, v1 = {acc[i][j][2], acc[i][j][3]};
        if (PART < 0 || PART == 2 * i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v0), rc, vo[i], 0, NT ? 2 : 0);
        if (PART < 0 || PART == 2 * i + 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v1), rc, vo[i] + 16, 0, NT ? 2 : 0);
      }
    }
    // The accumulators of the column restart from zero.  Written HERE by explicit moves: the MFMAs are inline asm (hipcc
    // sees no MFMA), and a plain `acc = 0.0` is materialised as a v_mov right in front of the first MFMA that uses it -
    // without the wait states a VALU write needs before an MFMA reads it as its accumulator (the MFMA then added to the
    // register's stale content: a B fragment; found as errors of one 4-k partial sum in acc[1][j][1] of some columns).
    if constexpr (PART < 0 || PART == 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#ifdef TL_BUILTIN_MFMA
          acc[i][j][e] = 0.0;
#else
          double z_;
          asm volatile("v_mov_b64 %0, 0" : "=v"(z_));
          acc[i][j][e] = z_;
#endif
        }
#ifndef TL_BUILTIN_MFMA
      asm volatile("s_nop 7" ::: "memory");
#endif
    }
  };
  constexpr std::integral_constant<int, -1> whole{};

  // ---- prologue: three k-tiles requested, the first one landed and published, its first fragments requested
  struct Pair { unsigned lin, tm, d; };
  Pair cons, nxt;                // the pair being computed; the workgroup's next pair (past the end: the same once more)
  cons.lin = blockIdx.x; locate(cons.lin, cons.tm, cons.d);
  auto next_of = [&](const Pair& c) __attribute__((always_inline)) {
    Pair n = c;
    if (c.lin + gridDim.x < ntile) { n.lin = c.lin + gridDim.x; locate(n.lin, n.tm, n.d); }
    return n;
  };
  nxt = next_of(cons);
  Src scur = src_of(cons.tm, cons.d), snxt = src_of(nxt.tm, nxt.d);
  int stage = 0;
  unsigned ptm = cons.tm, pd = cons.d;
  issue(std::integral_constant<int, 0>{}, scur, 0);
  issue(std::integral_constant<int, 1>{}, scur, 1);
  issue(std::integral_constant<int, 2>{}, scur, 2);
  // the stores a previous pair would have issued behind its last request: zeros into this pair's own right half, which
  // the pair overwrites later (the wave's stores to one address complete in order)
  flush(std::integral_constant<int, 0>{}, ptm, pd, 1, false, whole);
  if constexpr (STORE) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  readA(0);
  readB(0, uoff(0, 0));
  readB(1, uoff(0, 1));

#ifdef TL_TIMING
  unsigned ts_off = (unsigned)((blockIdx.x * 8 + wave) * TL_TS_PER_WG * 4);      // (every wave writes its own record: no branch)
  unsigned long t_b;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_b));
#endif
  // (ptm, pd): the pair whose right half's block columns 1..7 are still in their registers (the first time: zeros,
  // written to this pair's own right half)
  for (;;) {
    w_static_for<0, 24>([&](auto pc) __attribute__((always_inline)) {
      constexpr int p = decltype(pc)::value, n = tl_nact(p), U = 4 * n, pn = (p + 1) % 24;
      w_static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, sub = u / n, j = u % n, s8 = sub >> 1, q = sub & 1;
        if constexpr (u == U - 2) {
          // every LDS read of this stage has been issued: wait for them and for this wave's pieces of the next k-tile
          // (the one after it may stay in flight, and the stores issued behind it), barrier, then the next k-tile's
          // first A fragments, the block column due in this k-tile, and the request for the k-tile three ahead into the
          // stage just released
          __builtin_amdgcn_sched_barrier(0);
#ifdef TL_TIMING
          unsigned long t_a;
          // (the wait INSIDE the statement: the compiler takes an asm output as written at once and may spill or reuse the pair
          //  while the scalar cache is still about to write it - found as a memory fault through a clobbered descriptor)
          asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_a));
#endif
          constexpr int allow = STORE ? tl_allow(p) : 6;
          static_assert(allow == 6 || allow == 10 || allow == 14 || allow == 18, "vmcnt classes");
          if constexpr (allow == 18) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
          else if constexpr (allow == 14) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
          else if constexpr (allow == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
#ifdef TL_TIMING
          {
            // raw low words: [2 i] = clock at the wait of position i, [2 i + 1] = clock behind the barrier of position i - 1
            const unsigned lo_a = (unsigned)t_a, lo_b = (unsigned)t_b;
            const unsigned o1 = ts_off + 4;
            asm volatile("s_store_dword %0, %2, %3\n\ts_store_dword %1, %2, %4" :: "s"(lo_a), "s"(lo_b), "s"(g.ts), "s"(ts_off), "s"(o1) : "memory");
            ts_off += 8;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_b));
          }
#endif
          // (opaque: 24 positions are a whole number of trips round the ring, so the compiler knows every stage of the
          //  straight-line body, turns the fragment addresses into base + constants beyond the 16-bit offset field and
          //  keeps dozens of them in registers)
          int dstep = stage == TL_NSTAGE - 1 ? -(TL_NSTAGE - 1) * TL_STAGE : TL_STAGE;
          asm volatile("" : "+s"(dstep));
#pragma unroll
          for (int i = 0; i < 2; ++i) { bA[i][0] += dstep; bA[i][1] += dstep; }
          bB0 += dstep; bB1 += dstep;
          __builtin_amdgcn_sched_barrier(0);
          readA(0);
          // block column (p % 16) + 1 ... of the tile finished before this one: positions 0..6 (the previous pair's
          // right half, if any) and 16..22 (this pair's left half)
          // (with stores in the spread form the column has gone out part by part between the units above)
          if constexpr (!(STORE && TL_STORE_SPREAD)) {
            if constexpr (p <= 6) flush(std::integral_constant<int, p + 1>{}, ptm, pd, 1, p == 6, whole);
            if constexpr (p >= 16 && p <= 22) flush(std::integral_constant<int, p - 15>{}, cons.tm, cons.d, 0, p == 22, whole);
          }
          // the k-tile three ahead: position p + 3 of this pair, or of the workgroup's next pair
          if constexpr (p + 3 < 24) issue(std::integral_constant<int, p + 3>{}, scur, stage);
          else issue(std::integral_constant<int, p + 3 - 24>{}, snxt, stage);
          stage = stage == TL_NSTAGE - 1 ? 0 : stage + 1;
          __builtin_amdgcn_sched_barrier(0);
        }
        // B fragment two units ahead (the last two of a k-tile read the next k-tile's first two, behind the barrier)
        if constexpr (u + 2 < U) readB((u + 2) & 3, uoff(p, u + 2));
        else readB((u + 2) & 3, uoff(pn, u + 2 - U));
        if constexpr (u == 0) readA(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#ifdef TL_BUILTIN_MFMA
            acc[i][j][e] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[s8][q][i], fb[u & 3][e], acc[i][j][e], 0, 0, 0);
#else
            double c_ = acc[i][j][e];
            const double a_ = fa[s8][q][i], b_ = fb[u & 3][e];
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c_) : "v"(a_), "v"(b_));
            acc[i][j][e] = c_;
#endif
          }
        // spread form: store k of the deferred column behind unit k (U - 2) / 4 - all four in front of the position's wait
        if constexpr (STORE && TL_STORE_SPREAD && tl_pre(p) != 0) {
          w_static_for<0, 4>([&](auto kc) __attribute__((always_inline)) {
            if constexpr (u == decltype(kc)::value * (U - 2) / 4) {
              if constexpr (p <= 6) flush(std::integral_constant<int, p + 1>{}, ptm, pd, 1, p == 6, kc);
              else flush(std::integral_constant<int, p - 15>{}, cons.tm, cons.d, 0, p == 22, kc);
            }
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // a 128-column tile is complete: its block column 0 now (the next tile starts on it), the others inside the next
      // tile's k-tiles (above)
      if constexpr (p == 15) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // (MFMA results -> VALU / stores: wait states hipcc cannot see)
        flush(std::integral_constant<int, 0>{}, cons.tm, cons.d, 0, false, whole);
      }
      if constexpr (p == 23) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        flush(std::integral_constant<int, 0>{}, cons.tm, cons.d, 1, false, whole);
      }
    });
    ptm = cons.tm; pd = cons.d;
    if (cons.lin + gridDim.x >= ntile) break;
    cons = nxt;
    nxt = next_of(cons);
    scur = snxt;
    snxt = src_of(nxt.tm, nxt.d);
  }
  // the last pair's right half: block columns 1..7
  w_static_for<1, 8>([&](auto jc) __attribute__((always_inline)) {
    flush(jc, ptm, pd, 1, decltype(jc)::value == 7, whole);
  });
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef TL_TIMING
  asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

bool gemm_tall_ok(const GemmArgs& a);
hipError_t gemm_tall(hipStream_t st, const GemmArgs& a);

}  // namespace dgp
