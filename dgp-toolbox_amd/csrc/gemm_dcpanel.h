// Row-panel kernel of the backward dC product, Mp = 256 (round 4):
//     dC = sum_d [alpha s_d .* T_d] W_d^T  -  (alpha sum_d s_d) Ct  +  mbar u^T        (what tf.GradientTape
//                                                                                      derives for layers.py:249-263)
// with W_d^T upper triangular (k <= n inside each 256 x 256 block), T row-major [rows][D * 256].
// Why beside gemm_tallu.h: that kernel's unit is (256-row tile, 128-column half), so the left halves read the first 128
// columns of every T_d a second time (T is read 1.5 times: 26 GB of the iteration's 94 GB), its A operand goes through LDS
// (4 of its 6 LDS-DMA requests per wave and k-tile, two thirds of its stage) and its units - a tile x half x ALL D blocks - are
// coarse: 488 row tiles of a rank's share of 8 GPUs leave most of the 256 CUs one unit behind.  The k-tile stream of
// gemm_gpanel.h - a wave owns 16 rows across all 256 columns, its A fragments come straight from global memory into registers,
// only B goes through LDS - reads every operand once and deals 128-row panels.  Measured (profiles/r4_dcpanel.txt): a tie at
// 10^6 rows, 10 % faster at 125 000 - 250 000 rows: the dispatcher uses this kernel between 4096 and 300 000 rows.
//
// Shape.  One persistent workgroup per CU (8 waves), 128-row panels dealt round-robin.  A panel is D segments of sixteen
// k-tiles (16 rows of W_d^T each, t = 0 .. 15: the k-tile meets the column blocks j >= t, the long k-tiles come first); the
// sixteen k-tiles of a segment are straight-line code, the stream runs on across segments and panels:
//   * A (T): lane (li, lk) loads row li, k = 8 s8 + 2 lk + q as two 16-byte loads per k-tile into a ring of four k-tiles
//     of registers, three k-tiles ahead; scaled by alpha s[row][d] when its k-tile starts.
//   * B (W_d^T): LDS-DMA into a ring of four 32 KB stages (two halves of 128 columns in gemm_tall.h's B image), three
//     k-tiles ahead; k-tiles t >= 8 only request the right half.  Sixteen k-tiles per segment and four stages: stage and
//     register slot of a k-tile are compile-time constants.
//   * One barrier per k-tile, in the MIDDLE of it (gemm_gram.h's arrangement): it publishes k-tile n + 1 (whose requests went
//     out two k-tiles earlier; every wave waits for its own pieces first, counted) and certifies that every wave has left
//     k-tile n - 1, whose stage and ring slot take the requests for k-tile n + 3 right behind it.  The fragment reads run
//     two units ahead across the k-tile boundary: no drain at the boundary.  (Measured against a first version with the barrier
//     at the top of the k-tile: the same time - the stream's ~1000 cycles per k-tile beyond its MFMAs are not the barrier;
//     fragments four units ahead: the same again.  NOTES.md 13.5.)
//   * Epilogue once per panel: the rank-D term as two more k-steps on the matrix cores (as gemm_tallu.h), then
//     dC = acc - esc * Ct, 32 buffer stores per lane; the next panel's first three k-tiles are already in flight.
// Rows past P read as zeros (buffer resources end at the operand's last row) and are not stored.
// MFMA instruction and lane maps: gemm_f64.h (v_mfma_f64_4x4x4_4b_f64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gemm_wide.h"

namespace dgp {

struct DcPanelArgs {
  const double* T; long ldt;       // [P][ldt], ldt = D * 256, row-major
  const double* B;                 // [D * 256][256]: D upper-triangular 256 x 256 blocks stacked (zeros stored)
  double* C;                       // [P][256]
  const double* s; long as_ld;     // row scales: s[row * as_ld + d] multiplies block d of row `row`
  const double* eadd;              // [P][256]: C -= alpha * (sum_d s[row][d]) * eadd, or nullptr
  const double* rowf;              // [P][rank]   rank-r term C += rowf colf^T (r <= 8), or nullptr
  const double* colf;              // [256][rank]
  int rank;
  double alpha;
  long P;
  int D;                           // 1..8
};

constexpr int DP_STAGE = 32768, DP_NSTAGE = 4;
// requests per wave for k-tile t: two k rows of the right half, and of the left half while the k-tile reaches it
constexpr int dp_bops(int t) { return t <= 7 ? 4 : 2; }
// what tile n issues (a few units in): B and A of the k-tile three ahead, and in tile 8 the next segment's row scale
constexpr int dp_group(int n) { return dp_bops((n + 3) % 16) + 2 + (n % 16 == 8 ? 1 : 0); }
// what may stay in flight at the barrier in the middle of tile n, which needs tile n + 1 (requested in tile n - 2): the group
// issued in tile n - 1
constexpr int dp_allow(int n) { return dp_group((n + 15) % 16); }
// units (column block x k-step of 4) of tile n, and the units of a segment in front of it (ring position of its first fragment)
constexpr int dp_units(int n) { return 4 * (16 - n); }
constexpr int dp_ubase(int n) { return n == 0 ? 0 : dp_ubase(n - 1) + dp_units(n - 1); }
static_assert(dp_ubase(15) + dp_units(15) == 544 && 544 % 4 == 0, "a segment is a whole number of turns of the fragment ring");

template <int R>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dc_panel_kernel(DcPanelArgs g) {
  __shared__ __attribute__((aligned(1024))) char smem[DP_NSTAGE * DP_STAGE + 256 * 8 * 8];
  // the rank term's column factors [256][8] (zero past the rank): its B fragments come from here, not from eight rounds of
  // dependent loads per panel (the first version's epilogue took ~20 us per panel: half a segment's time)
  double* sC = reinterpret_cast<double*>(smem + DP_NSTAGE * DP_STAGE);
  if constexpr (R != 0) {
    const int rkc = R == 8 ? 8 : g.rank;
    for (int x = threadIdx.x; x < 256 * 8; x += 512) sC[x] = ((x & 7) < rkc) ? g.colf[(x >> 3) * rkc + (x & 7)] : 0.0;
  }
  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4, sw = lk & 1;
  const long npanel = (g.P + 127) / 128;
  if ((long)blockIdx.x >= npanel) return;
  const int D = g.D;

  // ---- B by LDS-DMA: wave w requests k rows 2 w, 2 w + 1 of a half; lane -> physical chunk `lane`, logical chunk
  //      lane ^ ((k >> 1) & 1) = lane ^ (w & 1)   (gemm_tall.h's B image)
  const unsigned offB = (unsigned)((wave * 2 * 256) * 8 + ((lane ^ (wave & 1)) * 16));
  unsigned bs0[DP_NSTAGE], bs1[DP_NSTAGE];          // (opaque per-stage fragment bases: gemm_gpanel.h)
#pragma unroll
  for (int s4 = 0; s4 < DP_NSTAGE; ++s4) {
    bs0[s4] = (unsigned)(s4 * DP_STAGE + (2 * lk) * 1024 + (li & 3) * 32 + sw * 16);
    bs1[s4] = (unsigned)(s4 * DP_STAGE + (2 * lk) * 1024 + (li & 3) * 32 + (1 - sw) * 16);
    asm volatile("" : "+v"(bs0[s4]), "+v"(bs1[s4]));
  }
  const unsigned offA = (unsigned)((long)li * g.ldt * 8 + (2 * lk) * 8);      // + 128 t + 64 s8

  // ---- segments (panel, d) of the workgroup, and what a segment's requests need
  struct Seg { long pan; int d; };
  auto succ = [&](const Seg& c, Seg& n) __attribute__((always_inline)) -> bool {
    n = c;
    if (c.d + 1 < D) { n.d = c.d + 1; return true; }
    if (c.pan + gridDim.x >= npanel) return false;      // (past the end: the same segment once more - its requests are harmless)
    n.pan = c.pan + gridDim.x; n.d = 0;
    return true;
  };
  struct Src { __amdgpu_buffer_rsrc_t a, b; const double* sc; };
  const long t_bytes = g.P * g.ldt * 8;
  auto src_of = [&](const Seg& sg) __attribute__((always_inline)) {
    Src r;
    const long row0 = sg.pan * 128 + wave * 16;
    const long aoff = (row0 * g.ldt + (long)sg.d * 256) * 8;
    const long left = t_bytes - aoff;
    r.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.T) + (left > 0 ? aoff : 0)), 0,
                                            (int)(left > 0x7ffffff0L ? 0x7ffffff0L : (left > 0 ? left : 0)), 0x00020000);
    r.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.B + (long)sg.d * 65536), 0, 65536 * 8, 0x00020000);
    const long row = row0 + li;
    r.sc = (row < g.P) ? g.s + row * g.as_ld + sg.d : nullptr;
    return r;
  };
  auto issue_B = [&](const Src& sc, int t, int stage) __attribute__((always_inline)) {
    char* base = smem + stage * DP_STAGE;
    const unsigned so = (unsigned)(16 * t * 256 * 8);
    const __amdgpu_buffer_rsrc_t rb = sc.b;      // (a local copy: with the member of the referenced struct as the builtin's
                                                 //  argument this hipcc dropped the whole kernel from the HOST side - no stub, no
                                                 //  diagnostic - and the link failed)
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, W_LDS3(base + 16384 + (wave * 2 + p2) * 1024), 16, offB + p2 * 2048 + 1024, so, 0, 0);
    if (t <= 7) {
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, W_LDS3(base + (wave * 2 + p2) * 1024), 16, offB + p2 * 2048, so, 0, 0);
    }
  };
  double fa[4][2][2];       // [ring slot][s8][q]
  auto load_A = [&](const Src& sc, int t, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int s8 = 0; s8 < 2; ++s8) {
      const d2_t v = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(sc.a, offA + (unsigned)(128 * t + 64 * s8), 0, 0));
      fa[slot][s8][0] = v[0];
      fa[slot][s8][1] = v[1];
    }
  };

  double acc[16][4];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double z_;
        asm volatile("v_mov_b64 %0, 0" : "=v"(z_));       // (explicit moves + wait states: the MFMAs below are inline asm)
        acc[j][e] = z_;
      }
    asm volatile("s_nop 7" ::: "memory");
  };

  // ---- epilogue of a panel.  acc[j][e] of lane (li, lk) is C[128 pan + 16 w + 4 (li >> 2) + lk][16 j + 4 (li & 3) + e]
  auto flush = [&](long pan) __attribute__((always_inline)) {
    const long row0 = pan * 128 + wave * 16;
    const long row = row0 + 4 * (li >> 2) + lk;
    // rows past P are neither read nor stored: resources of zero records
    const long left = (g.P - row0) * 2048;
    const int nrec = (int)(left > 32768 ? 32768 : (left > 0 ? left : 0));            // (16 rows of 2 KB)
    const long boff = left > 0 ? row0 * 256 : 0;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(g.C + boff, 0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>((g.eadd ? g.eadd : g.C) + boff), 0, g.eadd ? nrec : 0, 0x00020000);
    const unsigned vo = (unsigned)((4 * (li >> 2) + lk) * 2048 + (4 * (li & 3)) * 8);
    // everything the epilogue reads from memory is requested up front: the rank term's row factors, the row's scales, and
    // the first four block columns of Ct (the others follow four blocks ahead of their use)
    double ra[2] = {0.0, 0.0};
    if constexpr (R != 0) {
      const int rk = R == 8 ? 8 : g.rank;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        if (row0 + li < g.P && (R == 8 || 4 * ks + lk < rk)) ra[ks] = g.rowf[(row0 + li) * rk + 4 * ks + lk];
    }
    double esc = 0.0;
    if (g.eadd != nullptr && row < g.P) {
      for (int d = 0; d < D; ++d) esc += g.alpha * g.s[row * g.as_ld + d];
    }
    d2_t ev[4][2];
    auto load_e = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      ev[j % 4][0] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(re, vo + 128 * j, 0, 0));
      ev[j % 4][1] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(re, vo + 128 * j + 16, 0, 0));
    };
    w_static_for<0, 4>([&](auto jc) __attribute__((always_inline)) { load_e(jc); });
    if constexpr (R != 0) {
      static_assert(R == 8 || R == -1, "rank 8, or any rank 1..8 at run time (R = -1: factors past the rank read as zero)");
      // C += rowf colf^T as two more k-steps of 4: A fragment rowf[row li][4 ks + lk], B fragment colf[col][4 ks + lk] from LDS
      const double* cbp = sC + (4 * (li & 3)) * 8 + lk;
      w_static_for<0, 16>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        double cb[4][2];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) cb[e][ks] = cbp[(16 * j + e) * 8 + 4 * ks];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            double c_ = acc[j][e];
            const double a_ = ra[ks], b_ = cb[e][ks];
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c_) : "v"(a_), "v"(b_));
            acc[j][e] = c_;
          }
      });
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // (MFMA results -> VALU: wait states hipcc cannot see)
    w_static_for<0, 16>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      d2_t v0 = {acc[j][0], acc[j][1]}, v1 = {acc[j][2], acc[j][3]};
      v0 -= esc * ev[j % 4][0];
      v1 -= esc * ev[j % 4][1];
      if constexpr (j + 4 < 16) load_e(std::integral_constant<int, j + 4>{});
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v0), rc, vo + 128 * j, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v1), rc, vo + 128 * j + 16, 0, 0);
    });
    zero_acc();
  };

  // ---- B fragments: a ring of four units, read two units ahead of their MFMAs, also across the k-tile boundary
  double fb[4][4];
  auto readB = [&](auto nc, auto uc) __attribute__((always_inline)) {      // unit u of tile n
    constexpr int n = decltype(nc)::value, u = decltype(uc)::value, NB = 16 - n, st = n % 4;
    constexpr int ks = u / NB, j = n + u % NB, s8 = ks >> 1, q = ks & 1, slot = (dp_ubase(n) + u) % 4;
    constexpr unsigned off = (unsigned)((j >> 3) * 16384 + (8 * s8 + q) * 1024 + (j & 7) * 128);
    const d2_t v0 = *reinterpret_cast<const d2_t*>(smem + bs0[st] + off);
    const d2_t v1 = *reinterpret_cast<const d2_t*>(smem + bs1[st] + off);
    fb[slot][0] = v0[0]; fb[slot][1] = v0[1]; fb[slot][2] = v1[0]; fb[slot][3] = v1[1];
  };

  // ---- prologue: the first three k-tiles requested, the first one landed and published, its first fragments requested
  Seg cur, nxt;
  cur.pan = blockIdx.x; cur.d = 0;
  bool more = succ(cur, nxt);
  Src scur = src_of(cur), snxt = src_of(nxt);
  double sc = scur.sc ? g.alpha * scur.sc[0] : 0.0, sc_next = 0.0;
  issue_B(scur, 0, 0); load_A(scur, 0, 0);
  issue_B(scur, 1, 1); load_A(scur, 1, 1);
  issue_B(scur, 2, 2); load_A(scur, 2, 2);
  zero_acc();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  readB(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  readB(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  bool after_flush = false;

  for (;;) {
    // ================= the sixteen k-tiles of segment `cur` (tile n = k-tile t uses stage / ring slot n % 4)
    w_static_for<0, 16>([&](auto nc) __attribute__((always_inline)) {
      constexpr int n = decltype(nc)::value, t = n, st = n % 4;
      // the row scale of the segment on this k-tile's A fragments
#pragma unroll
      for (int s8 = 0; s8 < 2; ++s8)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          double x_ = fa[st][s8][q];
          asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x_) : "v"(sc));
          fa[st][s8][q] = x_;
        }
      asm volatile("s_nop 3" ::: "memory");                   // VALU result -> MFMA operand
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NB = 16 - t, U = 4 * NB, MID = U / 2;
      w_static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, ks = u / NB, j = t + u % NB, s8 = ks >> 1, q = ks & 1;
        if constexpr (u == MID) {
          // this wave's pieces of tile n + 1 have landed (the group issued in tile n - 1 may stay in flight; behind an
          // epilogue its 32 stores are younger still), then the barrier: tile n + 1 is published, everybody has left tile
          // n - 1 - its stage and ring slot take the requests for tile n + 3 (of this segment, or the first ones of the next);
          // in tile 8 also the next segment's row scale
          __builtin_amdgcn_sched_barrier(0);
          constexpr int allow = dp_allow(n);
          static_assert(allow == 4 || allow == 5 || allow == 6, "vmcnt classes");
          if constexpr (n == 0) {
            static_assert(allow == 6, "the requests of tile 2, issued in tile 15");
            if (after_flush) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");       // + the epilogue's 32 stores
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          } else if constexpr (allow == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else if constexpr (allow == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (n + 3 < 16) { issue_B(scur, n + 3, (n + 3) % 4); load_A(scur, n + 3, (n + 3) % 4); }
          else { issue_B(snxt, n + 3 - 16, (n + 3) % 4); load_A(snxt, n + 3 - 16, (n + 3) % 4); }
          if constexpr (n == 8) {
            // (always ONE load: the counted waits rely on it)
            const double* p = snxt.sc ? snxt.sc : g.s;
            sc_next = *p;
            if (!snxt.sc) sc_next = 0.0;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // B fragment two units ahead: of this k-tile, or - behind the barrier above - the first ones of the next
        if constexpr (u + 2 < U) readB(nc, std::integral_constant<int, u + 2>{});
        else readB(std::integral_constant<int, (n + 1) % 16>{}, std::integral_constant<int, u + 2 - U>{});
        const double a = fa[st][s8][q];
        constexpr int fslot = (dp_ubase(n) + u) % 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double c_ = acc[j][e];
          const double b_ = fb[fslot][e];
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c_) : "v"(a), "v"(b_));
          acc[j][e] = c_;
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    after_flush = false;
    if (cur.d == D - 1) { flush(cur.pan); after_flush = true; }
    if (!more) break;
    cur = nxt;
    scur = snxt;
    sc = g.alpha * sc_next;
    more = succ(cur, nxt);
    snxt = src_of(nxt);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace dgp
