// Tall-tile kernel for the backward product  dC = sum_d [2 vbar_d .* T_d] * W_d^T - (sum_d 2 vbar_d) c + mbar u^T
//   (the transpose of layers.py:254-263 in the hand-derived backward pass, DESIGN.md par. 2; the product gemm_wide.h runs
//   as <DIR 1, SCALED, EM 0>), Mp = 256: W_d^T upper triangular, K = D * 256, the A operand is the blocked T of the
//   forward pass with a row scale per k block.
// Same tile as gemm_tall.h (256 rows x 128 columns, eight waves stacked over the rows, 48 KB stages, six DMA requests
// per wave and k-tile, MFMAs with tied accumulators), mirrored: k-tile t of a triangular block needs the 16-column
// blocks cb >= t.  The LEFT half of the output (columns 0..127) takes 8 k-tiles per block d with 8, 7, .., 1 blocks, the
// RIGHT half 16 with 8 x 8, then 8, 7, .., 1: 136 block-steps per block d and 256 rows, the exact triangle.  A unit of
// work is (row tile, half): D * 16 or D * 8 k-tiles into the same 64 accumulator registers, then the whole epilogue at
// once (one in 64..128 k-tiles: no deferred stores).  T is read 1.5 times (the left half re-reads k < 128 of every
// block); at 9 ms of MFMA time per 10^6 rows the 24 GB are 3 ms of HBM time beside it.
// Row scales: a wave keeps alpha * s[row][d] of its own 32 rows in 2 KB of LDS (written by itself when a row tile
// starts), reads its two A-fragment rows' scale when a block d starts and multiplies the A fragments (8 v_mul_f64 per
// k-tile, inline asm like the MFMAs so that their order and the wait states between them are fixed here).
// Balance: the right halves cost 100, the left halves 36.  Workgroup b takes the right halves b, b + G, ..; the left
// halves are dealt so that the workgroups with one right half less get three left halves more (unit() below).
#pragma once
#include "gemm_tall.h"

namespace dgp {

struct TallUArgs {
  const double* A; long lda;       // blocked: [M / 128][D * 16][128][16], lda = D * 256
  const double* B; long ldb;       // [D * 256][ldb]: D upper-triangular 256 x 256 blocks stacked (zeros stored)
  double* C; long ldc;             // [M][ldc], row-major
  const double* s; long as_ld;     // row scales: s[row * as_ld + d] multiplies k block d of row `row`
  const double* eadd;              // [M][ldc]: C -= alpha * (sum_d s[row][d]) * eadd, or nullptr
  const double* rowf;              // [M][rank]   rank-r term C += rowf colf^T (r <= 8), or nullptr
  const double* colf;              // [256][rank]
  int rank;
  double alpha;
  long M;                          // rows, a multiple of 128
  int D;                           // 1..8
};

constexpr int TU_SCT = 8 * 32 * 8 * 8;    // 16 KB: [wave][32 rows][8 blocks d]
constexpr int tu_npos(int h) { return h ? 16 : 8; }
constexpr int tu_nact(int h, int t) { return h == 0 ? 8 - t : (t < 8 ? 8 : 16 - t); }

template <int R>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_tallu_kernel(TallUArgs g) {
  __shared__ __attribute__((aligned(1024))) char smem[TL_NSTAGE * TL_STAGE + TU_SCT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4, sw = lk & 1;
  const unsigned nb = (unsigned)((g.M + 255) / 256), G = gridDim.x, b = blockIdx.x;
  if (b >= nb) return;
  const int D = g.D;

  // ---- the workgroup's units, in order: its right halves, then its left halves
  const unsigned uq = nb / G, ur = nb % G, nbig = uq + (b < ur ? 1u : 0u);
  const unsigned uex = ur != 0 ? 3u * (G - ur) : 0u;
  const bool light = ur != 0 && b >= ur;
  auto unit = [&](unsigned i, unsigned& tm, int& h) __attribute__((always_inline)) -> bool {
    if (i < nbig) { tm = b + i * G; h = 1; return true; }
    const unsigned s = i - nbig;
    unsigned idx;
    if (light) idx = s < 3 ? s * (G - ur) + (b - ur) : uex + (s - 3) * G + b;
    else idx = uex + s * G + b;
    tm = idx; h = 0;
    return idx < nb;
  };
  struct Seg { unsigned i, tm; int h, d; };
  auto succ = [&](const Seg& c, Seg& n) __attribute__((always_inline)) -> bool {
    n = c;
    if (c.d + 1 < D) { n.d = c.d + 1; return true; }
    unsigned tm; int h;
    if (!unit(c.i + 1, tm, h)) return false;
    n.i = c.i + 1; n.tm = tm; n.h = h; n.d = 0;
    return true;
  };

  // ---- DMA.  A piece P = 4 w + p4 (8 rows x 128 B = 1 KB contiguous in the blocked operand): lane -> row 8 P + lane / 8,
  // physical chunk lane % 8, logical chunk = physical ^ ((row >> 1) & 7).  B piece P = 2 w + p2 (one k row, 1 KB): lane ->
  // physical chunk lane, logical = lane ^ (w & 1).  (gemm_tall.h)
  unsigned offA[4];
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) {
    const int c = (lane & 7) ^ ((4 * p4 + (lane >> 4)) & 7);
    offA[p4] = (unsigned)(((lane >> 3) * 16 + 2 * c) * 8);
  }
  const unsigned offB = (unsigned)((lane ^ (wave & 1)) * 16);
  const unsigned stepB = (unsigned)(g.ldb * 8);
  struct Src { const char* a; const char* b; int nrec; };
  auto src_of = [&](const Seg& sg) __attribute__((always_inline)) {
    Src r;
    const long panel = (long)sg.tm * 2 + (wave >> 2);
    const bool live = panel * 128 < g.M;
    r.a = reinterpret_cast<const char*>(g.A) + (live ? (panel * 128 * g.lda + (long)sg.d * 16 * 2048 + (wave & 3) * 32 * 16) * 8 : 0L);
    r.nrec = live ? 0x7ffffff0 : 0;
    r.b = reinterpret_cast<const char*>(g.B) + (((long)sg.d * 256 + wave * 2) * g.ldb + 128 * sg.h) * 8;
    return r;
  };
  auto issue = [&](auto tc, const Src& sc_, int stage) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sc_.a + (sc_.nrec ? t * 16384 : 0)), 0, sc_.nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sc_.b + (long)16 * t * g.ldb * 8), 0, 0x7ffffff0, 0x00020000);
    char* base = smem + stage * TL_STAGE;
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, W_LDS3(base + (wave * 4 + p4) * 1024), 16, offA[p4], p4 * 1024, 0, 0);
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
  auto zero_acc = [&]() __attribute__((always_inline)) {
    // explicit moves + wait states: the MFMAs are inline asm, hipcc would put a plain `= 0.0` right in front of them
    // (gemm_tall.h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double z_;
          asm volatile("v_mov_b64 %0, 0" : "=v"(z_));
          acc[i][j][e] = z_;
        }
    asm volatile("s_nop 7" ::: "memory");
  };
  zero_acc();
  double fa[2][2][2];       // [s8][q][i]
  double fb[4][4];          // ring of four units: [slot][e]
  double sc[2] = {0.0, 0.0};
  auto readA = [&](int s8) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const d2_t v = *reinterpret_cast<const d2_t*>(smem + bA[i][s8]);
      fa[s8][0][i] = v[0];
      fa[s8][1][i] = v[1];
    }
  };
  auto scaleA = [&](int s8) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double x_ = fa[s8][q][i];
        const double s_ = sc[i];
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x_) : "v"(s_));
        fa[s8][q][i] = x_;
      }
    asm volatile("s_nop 3" ::: "memory");                   // VALU result -> MFMA operand
  };
  auto readB = [&](int slot, unsigned off) __attribute__((always_inline)) {   // off: (8 s8 + q) * 1024 + j * 128
    const d2_t v0 = *reinterpret_cast<const d2_t*>(smem + bB0 + off);
    const d2_t v1 = *reinterpret_cast<const d2_t*>(smem + bB1 + off);
    fb[slot][0] = v0[0]; fb[slot][1] = v0[1]; fb[slot][2] = v1[0]; fb[slot][3] = v1[1];
  };
  auto uoff = [](int h, int t, int u) constexpr {
    const int n = tu_nact(h, t), sub = u / n, j = 8 - n + u % n;
    return (unsigned)((8 * (sub >> 1) + (sub & 1)) * 1024 + j * 128);
  };

  // ---- the wave's scale table: sct[(32 w + row) * 8 + d] = alpha * s[row][d] (zero past D and past M)
  double* sct = reinterpret_cast<double*>(smem + TL_NSTAGE * TL_STAGE) + wave * 256;
  auto load_table = [&](unsigned tm) __attribute__((always_inline)) {
    const long row = (long)tm * 256 + wave * 32 + (lane >> 1);
    const int d0 = (lane & 1) * 4;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      double v = 0.0;
      if (row < g.M && d0 + x < D) v = g.alpha * g.s[row * g.as_ld + d0 + x];
      sct[(lane >> 1) * 8 + d0 + x] = v;
    }
  };

  // ---- epilogue of a unit.  acc[i][j][e] of lane (li, lk) is
  //      C[256 tm + 32 w + 16 i + 4 (li >> 2) + lk][128 h + 16 j + 4 (li & 3) + e]
  // The rank-8 term C += rowf colf^T is two more k-steps of 4 on the matrix cores (as VALU code it was 8 x 64 FMAs per
  // lane behind 32 dependent round trips to L2: 1.2 ms of 11.5 per 10^6 rows): A fragment rowf[row li][4 ks + lk], B
  // fragment colf[col][4 ks + lk], the column factors of two block columns requested at a time.
  auto flush_all = [&](unsigned tm, int h) __attribute__((always_inline)) {
    const bool live = ((long)tm * 2 + (wave >> 2)) * 128 < g.M;
    if constexpr (R != 0) {
      static_assert(R == 8 || R == -1, "rank 8, or any rank 1..8 at run time (R = -1: factors past the rank read as zero)");
      if (live) {
        const int rk = R == 8 ? 8 : g.rank;
        double ra[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            ra[i][ks] = 0.0;
            if (R == 8 || 4 * ks + lk < rk) ra[i][ks] = g.rowf[((long)tm * 256 + wave * 32 + 16 * i + li) * rk + 4 * ks + lk];
          }
        // (opaque base: otherwise hipcc computes all 64 lane addresses once, in front of the loops, and keeps them in scratch)
        const double* cbp = g.colf + (128 * h + 4 * (li & 3)) * rk + lk;
        asm volatile("" : "+v"(cbp));
        w_static_for<0, 4>([&](auto jc2) __attribute__((always_inline)) {
          constexpr int j2 = decltype(jc2)::value;
          double cb[2][4][2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks) {
                cb[jj][e][ks] = 0.0;
                if (R == 8 || 4 * ks + lk < rk) cb[jj][e][ks] = cbp[(16 * (2 * j2 + jj) + e) * rk + 4 * ks];
              }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  double c_ = acc[i][2 * j2 + jj][e];
                  const double a_ = ra[i][ks], b_ = cb[jj][e][ks];
                  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c_) : "v"(a_), "v"(b_));
                  acc[i][2 * j2 + jj][e] = c_;
                }
        });
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // (MFMA results -> VALU: wait states hipcc cannot see)
    if (live) {
      double esc[2] = {0.0, 0.0};
      long row[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rit = 16 * i + 4 * (li >> 2) + lk;
        row[i] = (long)tm * 256 + wave * 32 + rit;
        if (g.eadd != nullptr) {
#pragma unroll
          for (int d = 0; d < 8; ++d) esc[i] += sct[rit * 8 + d];
        }
      }
      w_static_for<0, 8>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const long colb = 128 * h + 16 * j + 4 * (li & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const long o = row[i] * g.ldc + colb;
          d2_t v0 = {acc[i][j][0], acc[i][j][1]}, v1 = {acc[i][j][2], acc[i][j][3]};
          if (g.eadd != nullptr) {
            v0 -= esc[i] * *reinterpret_cast<const d2_t*>(g.eadd + o);
            v1 -= esc[i] * *reinterpret_cast<const d2_t*>(g.eadd + o + 2);
          }
          *reinterpret_cast<d2_t*>(g.C + o) = v0;
          *reinterpret_cast<d2_t*>(g.C + o + 2) = v1;
        }
      });
    }
    zero_acc();
  };

  // ---- one block d of a unit: tu_npos(H) k-tiles, straight-line
  int stage = 0;
  Src scur, snxt;
  auto body = [&](auto hc) __attribute__((always_inline)) {
    constexpr int H = decltype(hc)::value, NP = tu_npos(H);
    w_static_for<0, NP>([&](auto pc) __attribute__((always_inline)) {
      constexpr int p = decltype(pc)::value, n = tu_nact(H, p), U = 4 * n, j0 = 8 - n;
      w_static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, sub = u / n, j = j0 + u % n, s8 = sub >> 1, q = sub & 1;
        if constexpr (u == U - 2) {
          // every LDS read of this stage has been issued: wait for them and for this wave's pieces of the next k-tile
          // (the one after it may stay in flight), barrier, then the next k-tile's first A fragments and the request
          // for the k-tile three ahead into the stage just released
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          int dstep = stage == TL_NSTAGE - 1 ? -(TL_NSTAGE - 1) * TL_STAGE : TL_STAGE;
          asm volatile("" : "+s"(dstep));                   // (opaque: gemm_tall.h)
#pragma unroll
          for (int i = 0; i < 2; ++i) { bA[i][0] += dstep; bA[i][1] += dstep; }
          bB0 += dstep; bB1 += dstep;
          __builtin_amdgcn_sched_barrier(0);
          readA(0);
          if constexpr (p + 3 < NP) issue(std::integral_constant<int, p + 3>{}, scur, stage);
          else issue(std::integral_constant<int, p + 3 - NP>{}, snxt, stage);
          stage = stage == TL_NSTAGE - 1 ? 0 : stage + 1;
          __builtin_amdgcn_sched_barrier(0);
        }
        // B fragment two units ahead (the last two of a k-tile read the next k-tile's first two, behind the barrier;
        // the first k-tile of any block has all eight column blocks, whichever half follows)
        if constexpr (u + 2 < U) readB((u + 2) & 3, uoff(H, p, u + 2));
        else if constexpr (p + 1 < NP) readB((u + 2) & 3, uoff(H, p + 1, u + 2 - U));
        else readB((u + 2) & 3, uoff(1, 0, u + 2 - U));
        if constexpr (u == 0) { readA(1); scaleA(0); }
        if constexpr (u == 2 * n) scaleA(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            double c_ = acc[i][j][e];
            const double a_ = fa[s8][q][i], b_ = fb[u & 3][e];
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c_) : "v"(a_), "v"(b_));
            acc[i][j][e] = c_;
          }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- prologue: three k-tiles requested, the first one landed and published, its first fragments requested
  Seg cur, nxt;
  cur.i = 0; cur.d = 0;
  unit(0, cur.tm, cur.h);
  bool more = succ(cur, nxt);
  scur = src_of(cur);
  snxt = src_of(nxt);
  issue(std::integral_constant<int, 0>{}, scur, 0);
  issue(std::integral_constant<int, 1>{}, scur, 1);
  issue(std::integral_constant<int, 2>{}, scur, 2);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  readA(0);
  readB(0, uoff(1, 0, 0));
  readB(1, uoff(1, 0, 1));

  // Two loops, one per half, rather than one loop with a branch on the half: the accumulators and fragments then live in
  // the same registers all the way round each loop (with the branch hipcc moved ~50 accumulators through scratch at
  // every block d).  The workgroup's units are ordered right halves first, so each loop runs once.
  bool done = false;
  auto run_half = [&](auto hc) __attribute__((always_inline)) {
    constexpr int H = decltype(hc)::value;
    for (;;) {
      if (cur.d == 0) load_table(cur.tm);
#pragma unroll
      for (int i = 0; i < 2; ++i) sc[i] = sct[(16 * i + li) * 8 + cur.d];
      body(hc);
      if (cur.d == D - 1) flush_all(cur.tm, H);
      if (!more) { done = true; break; }
      cur = nxt;
      scur = snxt;
      more = succ(cur, nxt);
      snxt = src_of(nxt);
      if (cur.h != H) break;
    }
  };
  run_half(std::integral_constant<int, 1>{});
  if (!done) run_half(std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

bool gemm_tallu_ok(const GemmArgs& a);
hipError_t gemm_tallu(hipStream_t st, const GemmArgs& a);

}  // namespace dgp
