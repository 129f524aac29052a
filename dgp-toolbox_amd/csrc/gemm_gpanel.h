// Row-panel kernel of the backward pass through Kuf for the stationary kernels, Mp = 256 (round 4):
//     g  = (Cbar * Linv) .* E          (dK = dC Lu^-1, then g = dK .* e with e = k for the squared exponential;
//                                       what tf.GradientTape derives for layers.py:243-245, dgp.py:272-275)
//     R1 = g  [Z | 1]                  [rows x W]   -> x-gradient, second part of the lengthscale gradient
//     GX = g^T [X | 1]                 [256 x W]    -> Z-, lengthscale- and variance-gradient
// in ONE launch that never writes g.  Rounds 1-3 ran the product on the 128 x 64 engine (reads Cbar and E, writes g: 6 GB
// per 10^6 rows, HBM-bound at 1.8 ms) and the two contractions as a second pass over g (rg_contract_kernel, 0.48 ms): 8 GB
// of HBM traffic per 10^6 rows where 4 GB are input.
//
// Shape.  One persistent workgroup per CU (8 waves) walks a contiguous range of 128-row panels.  Wave w owns rows
// 16 w .. 16 w + 15 of the panel across all 256 columns: acc[16 blocks][4] = 128 registers, every wave runs the same
// straight-line code, and the lower-triangular Linv is exploited exactly (k-tile t of 16 rows of Linv meets the column
// blocks j <= t: 136 of 256 block-steps).  The k-tiles run from t = 15 down to 0: the panel starts with its long k-tiles
// (the operand prefetch of two k-tiles covers the memory latency from the first one on).
//   * A operand (Cbar): a wave's rows are private to it -> 16-byte loads straight into the MFMA fragment layout (lane
//     (li, lk): row li, k = 8 s8 + 2 lk + q), a ring of three k-tiles in registers; no LDS, no barrier for A.
//   * B operand (Linv): LDS-DMA into a ring of three 32 KB stages (two 16 KB halves of 128 columns in gemm_tall.h's
//     B image: [16 k][64 chunks of 16 B], chunk ^ ((k >> 1) & 1)); k-tiles t < 8 only request the left half.
//   * Epilogue, eight rounds per panel: wave r parks its 16 x 256 rows of dK in LDS (the ring's third stage, idle until the
//     next panel's third k-tile), all 512 threads multiply the image by E (8 elements each, fetched one round ahead: a
//     version that multiplied in the accumulators kept three block columns of E in flight during the k-tiles and spilled),
//     then ALL waves contract that image on the matrix cores as points.hip's
//     rg_contract_kernel does: R1 partial over the wave's 32 columns (Z1 image in LDS, partials of the 8 waves added in a
//     fixed order), GX of the wave's two column blocks accumulated in registers over the workgroup's whole range and
//     written once to a slab that rg_reduce_kernel adds in workgroup order.  No atomics: results are reproducible.
//   * Rows past P (the last panel) are read through buffer resources that end at row P: they contribute zeros.
// MFMA instruction and lane maps: gemm_f64.h (v_mfma_f64_4x4x4_4b_f64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gemm_wide.h"

namespace dgp {

struct GPanelArgs {
  const double* A;     // Cbar [P][256]
  const double* B;     // Linv [256][256], lower triangular (the zeros above the diagonal are stored)
  const double* E;     // [P][256]: Kuf^T for the squared exponential, the stored derivative factor otherwise
  const double* Z1;    // [256][W]
  const double* X1;    // [P][W]
  double* R1;          // [P][W]
  double* slab;        // [gridDim.x][256][W], or nullptr (no GX wanted)
  long P;              // rows
  long panels_per_wg;
};

constexpr int GP_STAGE = 32768;                      // one k-tile of Linv: 2 halves x [16 k][1 KB]
constexpr int GP_RING = 3 * GP_STAGE;
constexpr int GP_PITCH = 260;                        // doubles per row of the parked g image (as rg_contract_kernel's)
constexpr int GP_SG = 2 * GP_STAGE;                  // the g image lives in the third stage (+ 512 B of the pad behind the ring)
constexpr int GP_SZ = GP_RING + 1024;                // Z1 image [256][12]
constexpr int GP_SX = GP_SZ + 256 * 12 * 8;          // X1 rows of the round [16][12]
constexpr int GP_SP = GP_SX + 16 * 12 * 8;           // R1 partials, double buffered: [2][8 waves][64 lanes][3]
constexpr int GP_LDS = GP_SP + 2 * 8 * 64 * 3 * 8;   // 150 016 B
static_assert(16 * GP_PITCH * 8 <= GP_STAGE + 1024, "g image fits the third stage + pad");

// B requests of a wave for k-tile t: two k rows of the left half, and of the right half when the k-tile reaches it
constexpr int gp_bops(int t) { return t >= 8 ? 4 : 2; }

template <int W>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void g_panel_kernel(GPanelArgs g) {
  __shared__ __attribute__((aligned(1024))) char smem[GP_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4, sw = lk & 1;
  const long npanel = (g.P + 127) / 128;
  const long pan0 = (long)blockIdx.x * g.panels_per_wg;
  const long pan1 = pan0 + g.panels_per_wg < npanel ? pan0 + g.panels_per_wg : npanel;
  if (pan0 >= pan1) return;

  double* sZ = reinterpret_cast<double*>(smem + GP_SZ);
  double* sX = reinterpret_cast<double*>(smem + GP_SX);
  double* sP = reinterpret_cast<double*>(smem + GP_SP);
  double* sG = reinterpret_cast<double*>(smem + GP_SG);
  for (int x = tid; x < 256 * 12; x += 512) {
    const int m = x / 12, jj = x - m * 12, j = 3 * (jj >> 2) + (jj & 3);
    sZ[x] = ((jj & 3) < 3 && j < W) ? g.Z1[m * W + j] : 0.0;
  }
  if (tid < 16 * 12) sX[tid] = 0.0;

  // ---- B (Linv) by LDS-DMA: wave w requests k rows 2 w, 2 w + 1 of a half; lane -> physical chunk `lane`, logical chunk
  //      lane ^ ((k >> 1) & 1) = lane ^ (w & 1)   (gemm_tall.h's B image)
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.B), 0, 256 * 256 * 8, 0x00020000);
  const unsigned offB = (unsigned)((wave * 2 * 256) * 8 + ((lane ^ (wave & 1)) * 16));
  auto issue_B = [&](int t, int stage) __attribute__((always_inline)) {
    char* base = smem + stage * GP_STAGE;
    const unsigned so = (unsigned)(16 * t * 256 * 8);
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, W_LDS3(base + (wave * 2 + p2) * 1024), 16, offB + p2 * 2048, so, 0, 0);
    if (t >= 8) {
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, W_LDS3(base + 16384 + (wave * 2 + p2) * 1024), 16, offB + p2 * 2048 + 1024, so, 0, 0);
    }
  };
  // fragment addresses inside a half (stage and half added per read): k row 8 s8 + 2 lk + q sits at ((8 s8 + q) + 2 lk) KB
  // (one opaque base per stage: left to itself hipcc folds stage + offset into constants beyond the 16-bit offset field of
  //  ds_read and keeps dozens of precomputed addresses in registers - 250 spilled registers in the first build)
  unsigned bs0[3], bs1[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    bs0[s3] = (unsigned)(s3 * GP_STAGE + (2 * lk) * 1024 + (li & 3) * 32 + sw * 16);
    bs1[s3] = (unsigned)(s3 * GP_STAGE + (2 * lk) * 1024 + (li & 3) * 32 + (1 - sw) * 16);
    asm volatile("" : "+v"(bs0[s3]), "+v"(bs1[s3]));
  }

  // ---- A (Cbar) and E: per panel a resource that starts at the wave's first row and ends at row P
  struct RowSrc { __amdgpu_buffer_rsrc_t a; };
  auto rows_of = [&](long pan) __attribute__((always_inline)) {
    const long row0 = pan * 128 + wave * 16;
    const long left = (g.P - row0) * 2048;
    const int nrec = (int)(left > 0x7ffffff0L ? 0x7ffffff0L : (left > 0 ? left : 0));
    const long off = left > 0 ? row0 * 256 : 0;
    RowSrc r;
    r.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.A + off), 0, nrec, 0x00020000);
    return r;
  };
  const unsigned offA = (unsigned)(li * 2048 + (2 * lk) * 8);                               // + 16 t * 8 + 64 s8

  double fa[3][2][2];       // [ring slot][s8][q]
  auto load_A = [&](const RowSrc& rs, int t, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int s8 = 0; s8 < 2; ++s8) {
      const d2_t v = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(rs.a, offA + (unsigned)(128 * t + 64 * s8), 0, 0));
      fa[slot][s8][0] = v[0];
      fa[slot][s8][1] = v[1];
    }
  };
  // Workgroup barrier for the epilogue's LDS hand-overs.  NOT __syncthreads(): that one waits for vmcnt(0) as well, i.e. for the
  // next round's E / X1 rows (requested a round ahead precisely so that nobody waits for them) and for the next panel's
  // operand requests - a memory latency at each of the 24 barriers of a panel (the epilogue was 29 % of the kernel).
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  double acc[16][4];
  double gacc[2][3];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 3; ++e) gacc[b][e] = 0.0;

  // thread -> element of the X1 rows it stages per round and of the R1 rows it writes
  const int xp = tid / W, xj = tid - xp * W, xjj = 4 * (xj / 3) + xj % 3;
  const int r1_lane = 4 * (xp >> 2) + xj / 3 + 16 * (xp & 3), r1_e = xj % 3;

  RowSrc cur = rows_of(pan0);
  issue_B(15, 0); load_A(cur, 15, 0);
  issue_B(14, 1); load_A(cur, 14, 1);

  for (long pan = pan0; pan < pan1; ++pan) {
    const long prow0 = pan * 128;
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double z_;
        asm volatile("v_mov_b64 %0, 0" : "=v"(z_));       // (explicit moves + wait states: the MFMAs below are inline asm)
        acc[j][e] = z_;
      }
    asm volatile("s_nop 7" ::: "memory");

    // ================= the sixteen k-tiles, t = 15 .. 0 (tile n = 15 - t uses ring slot / stage n % 3)
    w_static_for<0, 16>([&](auto nc) __attribute__((always_inline)) {
      constexpr int n = decltype(nc)::value, t = 15 - n, st = n % 3;
      // B(n) and A(n) have landed (what may still be in flight: the requests issued at the top of tile n - 1, i.e. B / A of
      // tile n + 1; tile 0 follows an epilogue: everything has landed), then the barrier that
      // publishes B(n) and certifies that every wave has left tile n - 1, whose stage takes the requests for tile n + 2
      if constexpr (n == 0 || n == 15) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (gp_bops(t - 1) == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // units (k-step, block): B fragment two units ahead
      constexpr int NB = t + 1, U = 4 * NB;
      double fb[3][4];
      auto readB = [&](int slot, int u) __attribute__((always_inline)) {
        const int ks = u / NB, j = u - ks * NB, s8 = ks >> 1, q = ks & 1;
        const unsigned off = (unsigned)((j >> 3) * 16384 + (8 * s8 + q) * 1024 + (j & 7) * 128);
        const d2_t v0 = *reinterpret_cast<const d2_t*>(smem + bs0[st] + off);
        const d2_t v1 = *reinterpret_cast<const d2_t*>(smem + bs1[st] + off);
        fb[slot][0] = v0[0]; fb[slot][1] = v0[1]; fb[slot][2] = v1[0]; fb[slot][3] = v1[1];
      };
      readB(0, 0);
      if constexpr (U > 1) readB(1, 1);
      w_static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, ks = u / NB, j = u % NB, s8 = ks >> 1, q = ks & 1;
        // the requests for tile n + 2 go out a few units into the tile (its stage was released by the barrier above): issued
        // right behind the barrier, all eight waves spent their first ~500 cycles of every k-tile on them with the matrix
        // cores idle
        if constexpr (u == (U > 3 ? 3 : U - 1) && n + 2 <= 15) { issue_B(t - 2, (n + 2) % 3); load_A(cur, t - 2, (n + 2) % 3); }
        if constexpr (u + 2 < U) readB((u + 2) % 3, u + 2);
        const double a = fa[st][s8][q];
        // (tied accumulators: with the builtin hipcc renames the 64 accumulator pairs and spills ~200 registers; as in
        //  gemm_tall.h / gemm_gram.h the compiler then sees no MFMA - the VALU <-> MFMA wait states are placed by hand below)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double c_ = acc[j][e];
          const double b_ = fb[u % 3][e];
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c_) : "v"(a), "v"(b_));
          acc[j][e] = c_;
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // (MFMA results -> LDS writes: wait states hipcc cannot see)

    // ================= epilogue: everybody has left tile 15 -> stages 0 and 1 take the next panel's first two k-tiles
    __builtin_amdgcn_s_barrier();
    if (pan + 1 < pan1) {
      cur = rows_of(pan + 1);
      issue_B(15, 0); load_A(cur, 15, 0);
      issue_B(14, 1); load_A(cur, 14, 1);
    }
    // factor E of the round's 16 x 256 image, one round ahead: thread -> row tid / 32, the 16-byte chunks tid % 32 + 32 q4 of the
    // row (consecutive lanes on consecutive chunks: coalesced loads, conflict-free LDS accesses; eight consecutive columns per
    // thread put the lanes 64 bytes apart - four-way bank conflicts on every access of the multiply)
    const int erow = tid >> 5, ecol = (tid & 31) * 2;
    d2_t er[4];
    auto load_Er = [&](long row0) __attribute__((always_inline)) {
      if (row0 + erow < g.P) {
        const d2_t* src = reinterpret_cast<const d2_t*>(g.E + (row0 + erow) * 256 + ecol);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) er[q4] = src[32 * q4];
      } else {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) er[q4] = (d2_t){0.0, 0.0};
      }
    };
    load_Er(prow0);
    // (the round's rows of X1 are fetched one round ahead as well: fetched at the top of their own round they were waited for
    //  in front of the round's second barrier - about a memory latency per round, eight times per panel)
    double xv = (tid < 16 * W && prow0 + xp < g.P) ? g.X1[(prow0 + xp) * W + xj] : 0.0;
#ifdef GP_DBG_NO_ROUNDS      // (timing experiment: the k-tiles alone)
    for (int r = 0; r < 0; ++r) {
#else
    for (int r = 0; r < 8; ++r) {
#endif
      const long rrow0 = prow0 + 16 * r;
      lds_barrier();              // the previous round's reads of sG / sX are done, its partials are complete in sP[(r - 1) & 1]
      if (r > 0 && tid < 16 * W && rrow0 - 16 + xp < g.P) {
        const double* p = sP + ((r - 1) & 1) * (8 * 64 * 3) + r1_lane * 3 + r1_e;
        g.R1[(rrow0 - 16 + xp) * W + xj] = ((p[0 * 192] + p[1 * 192]) + (p[2 * 192] + p[3 * 192])) + ((p[4 * 192] + p[5 * 192]) + (p[6 * 192] + p[7 * 192]));
      }
      if (wave == r) {            // acc[j][e] of lane (li, lk) is dK[16 r + 4 (li >> 2) + lk][16 j + 4 (li & 3) + e]
        double* row = sG + (4 * (li >> 2) + lk) * GP_PITCH + 4 * (li & 3);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          *reinterpret_cast<d2_t*>(row + 16 * j) = (d2_t){acc[j][0], acc[j][1]};
          *reinterpret_cast<d2_t*>(row + 16 * j + 2) = (d2_t){acc[j][2], acc[j][3]};
        }
      }
      if (tid < 16 * W) {
        sX[xp * 12 + xjj] = xv;
        xv = (r + 1 < 8 && rrow0 + 16 + xp < g.P) ? g.X1[(rrow0 + 16 + xp) * W + xj] : 0.0;
      }
      lds_barrier();
      {                           // g = dK .* E in place, then the next round's factor
        d2_t* im = reinterpret_cast<d2_t*>(sG + erow * GP_PITCH + ecol);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) im[32 * q4] = im[32 * q4] * er[q4];
        if (r + 1 < 8) load_Er(rrow0 + 16);
      }
      lds_barrier();
      // ---- R1 partial of this wave: columns 32 wave .. 32 wave + 31 of g (k-steps of 4 columns)
      double racc[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int ss = 0; ss < 8; ++ss) {
        const int s = 8 * wave + ss;
        const double a = sG[li * GP_PITCH + 4 * s + lk];
        const double* zp = sZ + (4 * s + lk) * 12 + 4 * (li & 3);
        const d2_t z01 = *reinterpret_cast<const d2_t*>(zp);
        const double z2 = zp[2];
        racc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, z01[0], racc[0], 0, 0, 0);
        racc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, z01[1], racc[1], 0, 0, 0);
        racc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, z2, racc[2], 0, 0, 0);
      }
      {
        double* p = sP + (r & 1) * (8 * 64 * 3) + (wave * 64 + lane) * 3;
        p[0] = racc[0]; p[1] = racc[1]; p[2] = racc[2];
      }
      // ---- GX of this wave's two column blocks 2 wave, 2 wave + 1: k = the round's 16 rows
      if (g.slab != nullptr) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double* xq = sX + (4 * s + lk) * 12 + 4 * (li & 3);
          const d2_t x01 = *reinterpret_cast<const d2_t*>(xq);
          const double x2 = xq[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const double a = sG[(4 * s + lk) * GP_PITCH + 16 * (2 * wave + b) + li];
            gacc[b][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x01[0], gacc[b][0], 0, 0, 0);
            gacc[b][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x01[1], gacc[b][1], 0, 0, 0);
            gacc[b][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x2, gacc[b][2], 0, 0, 0);
          }
        }
      }
    }
    lds_barrier();                // the last round's partials
    if (tid < 16 * W && prow0 + 112 + xp < g.P) {
      const double* p = sP + (7 & 1) * (8 * 64 * 3) + r1_lane * 3 + r1_e;
      g.R1[(prow0 + 112 + xp) * W + xj] = ((p[0 * 192] + p[1 * 192]) + (p[2 * 192] + p[3 * 192])) + ((p[4 * 192] + p[5 * 192]) + (p[6 * 192] + p[7 * 192]));
    }
    // (the next panel's tile 0 starts with vmcnt(0) + barrier: the g image in the third stage is not touched before its
    //  tile 0 requests k-tile 13 into it, behind that barrier)
  }
  if (g.slab != nullptr) {
    // gacc[b][e] of lane (li, lk) is GX[16 (2 wave + b) + 4 (li >> 2) + lk][3 (li & 3) + e]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int m = 16 * (2 * wave + b) + 4 * (li >> 2) + lk, j = 3 * (li & 3) + e;
        if (j < W) g.slab[((long)blockIdx.x * 256 + m) * W + j] = gacc[b][e];
      }
  }
}

// host side: g_panel_ok / g_panel (gemm_gpanel.hip, declared in dgp_internal.h); the slab scratch is the one of points.hip's
// one-pass contraction (rbf_bwd_contract_ws_doubles: 512 slabs of 256 x 9)

}  // namespace dgp
