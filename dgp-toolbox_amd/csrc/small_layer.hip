// Fused per-point kernels of an SVGP layer with at most 64 inducing points (Mp = 64): the Bayesian-optimisation surrogates
// of SO_BO.py:248-258 (tens of inducing points, a few hundred to a few thousand points).  At that size the per-layer
// sequence Kuf -> Ct -> T -> mean0 and its adjoint ran as 4 + 8 launches of the 128 x 64 MFMA engine at 11-35 us each (a
// 1000-point product is a handful of tiles: the engine's pipeline never fills), i.e. 0.4 of the 0.67 ms of a config-1
// training iteration.  Here: one wave per point, lane = inducing index, the layer's 64 x 64 matrices in LDS, matrix-vector
// products as 64 broadcast-FMA steps (v_readlane + ds_read + v_fma); the reductions over the points as one launch.
//
// Reference arithmetic (whitened form, DESIGN.md par. 2):
//   forward  (layers.py:243-276):  k = K(Z, x), c = Lu^-1 k, t_d = W_d^T c, mean0_d = c . u_d, |c|^2, |t_d|^2
//   backward:      cb = sum_d mbar_d u_d + sum_d 2 vbar_d (W_d t_d - c),  kb = Lu^-T cb,  g = kb .* e,
//                                  R1 = g^T [Z | 1];   sums over points: G_d, Q', du, g^T [X | 1]
// Outputs have the layouts the unfused path uses (Kt, Ct, Cbar, Gt point-major [P][64]; T row-major [P][D * 64]; one
// row-norm plane), so finalize_layer, xbar_finish, fold and the small-matrix chains run unchanged behind them.
#include "dgp_internal.h"
#include "chain_bodies.h"

namespace dgp {

constexpr int SL_M = 64;           // Mp
constexpr int SL_MAXD = 3;         // outputs whose W_d fit the LDS beside Linv (32 KB each)

__device__ __forceinline__ double sl_bcast(double v, int lane) {       // lane: wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sl_wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int SL_Q = 4;            // points a wave works on at once: one LDS read of a matrix row serves SL_Q matrix-vector products,
                                   // SL_Q independent accumulation chains hide the FMA and LDS latencies (1 wave per SIMD)

// y_q[lane] = sum_j A[j][lane] * x_q[j], q < SL_Q.  A: LDS image [64][64] (row j contiguous over the lanes); x: the wave's
// LDS scratch xs[q][64] (written by the wave itself: a uniform address is a broadcast read)
__device__ __forceinline__ void sl_matvec4(const double* __restrict__ A, const double* __restrict__ xs, double (&y)[SL_Q]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < SL_Q; ++q) y[q] = 0.0;
#pragma unroll 8
  for (int j = 0; j < SL_M; ++j) {
    const double a = A[j * SL_M + lane];
#pragma unroll
    for (int q = 0; q < SL_Q; ++q) y[q] = fma(a, xs[q * SL_M + j], y[q]);
  }
}
__device__ __forceinline__ void sl_wave_fence() {        // the wave's own LDS writes are visible to its later reads
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void sl_wave_sum4(double (&v)[SL_Q]) {
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int q = 0; q < SL_Q; ++q) v[q] += __shfl_xor(v[q], o);
  }
}

// ---------------------------------------------------------------------------------------------------- forward
// grid: blocks of 4 waves; a wave takes the block's points in groups of SL_Q
__global__ __launch_bounds__(256) void small_layer_fwd_kernel(
    int kind, const double* __restrict__ Xin, long P, long x_row0, int Din, const double* __restrict__ Z, const double* __restrict__ var,
    const double* __restrict__ ls, int M, int D, const double* __restrict__ LinvT, const double* __restrict__ Wcat,
    const double* __restrict__ u, double* __restrict__ Kt, double* __restrict__ Et, double* __restrict__ Ct, double* __restrict__ Tt,
    double* __restrict__ cn, double* __restrict__ tn, long pstride, double* __restrict__ mean0, int pts_per_block) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sL = smem;                           // LinvT [64][64]:  c[m] = sum_j LinvT[j][m] k[j]
  double* sW = smem + SL_M * SL_M;             // W_d [m][n] (d-major):  t_d[n] = sum_m W_d[m][n] c[m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* xs = smem + (1 + D) * SL_M * SL_M + wave * SL_Q * SL_M;      // the wave's vectors
  for (int i = tid; i < SL_M * SL_M; i += 256) sL[i] = LinvT[i];
  for (int i = tid; i < D * SL_M * SL_M; i += 256) {
    const int d = i / (SL_M * SL_M), r = i % (SL_M * SL_M), m = r / SL_M, n = r % SL_M;
    sW[i] = Wcat[(long)m * D * SL_M + (long)d * SL_M + n];
  }
  __syncthreads();
  const double v0 = var[0];
  double ud[SL_MAXD];
#pragma unroll
  for (int d = 0; d < SL_MAXD; ++d) ud[d] = d < D ? u[(long)lane * D + d] : 0.0;
  const long p0 = (long)blockIdx.x * pts_per_block, p1 = min(P, p0 + pts_per_block);
  for (long pg = p0 + wave * SL_Q; pg < p1; pg += 4 * SL_Q) {
    double k[SL_Q], c[SL_Q], t[SL_Q], red[SL_Q];
#pragma unroll
    for (int q = 0; q < SL_Q; ++q) {
      const long p = min(pg + q, p1 - 1);            // (a short last group repeats its last point; only p < p1 is stored)
      const double* __restrict__ x = Xin + (x_row0 + p) * Din;
      double r2 = 0.0, e = 0.0;
      k[q] = 0.0;
      if (lane < M) {
        for (int j = 0; j < Din; ++j) { const double dd = (x[j] - Z[(long)lane * Din + j]) / ls[j]; r2 += dd * dd; }
        stationary_k(kind, v0, r2, k[q], e);
      }
      if (pg + q < p1) {
        Kt[p * SL_M + lane] = k[q];
        if (Et) Et[p * SL_M + lane] = e;
      }
      xs[q * SL_M + lane] = k[q];
    }
    sl_wave_fence();
    sl_matvec4(sL, xs, c);
    sl_wave_fence();
#pragma unroll
    for (int q = 0; q < SL_Q; ++q) {
      if (pg + q < p1) Ct[(pg + q) * SL_M + lane] = c[q];
      xs[q * SL_M + lane] = c[q];
      red[q] = c[q] * c[q];
    }
    sl_wave_sum4(red);
    if (lane < SL_Q && pg + lane < p1) cn[pg + lane] = lane == 0 ? red[0] : (lane == 1 ? red[1] : (lane == 2 ? red[2] : red[3]));
    sl_wave_fence();
    for (int d = 0; d < D; ++d) {
      sl_matvec4(sW + d * SL_M * SL_M, xs, t);
      const double udd = d == 0 ? ud[0] : (d == 1 ? ud[1] : ud[2]);
      double m0[SL_Q];
#pragma unroll
      for (int q = 0; q < SL_Q; ++q) {
        if (Tt && pg + q < p1) Tt[(pg + q) * (long)D * SL_M + (long)d * SL_M + lane] = t[q];
        red[q] = t[q] * t[q];
        m0[q] = c[q] * udd;
      }
      sl_wave_sum4(red);
      sl_wave_sum4(m0);
      if (lane < SL_Q && pg + lane < p1) {
        const double tv = lane == 0 ? red[0] : (lane == 1 ? red[1] : (lane == 2 ? red[2] : red[3]));
        const double mv = lane == 0 ? m0[0] : (lane == 1 ? m0[1] : (lane == 2 ? m0[2] : m0[3]));
        tn[(long)d * pstride + pg + lane] = tv;
        mean0[(pg + lane) * D + d] = mv;
      }
    }
    sl_wave_fence();                                  // (the next group overwrites xs)
  }
}

// ---------------------------------------------------------------------------------------------------- backward, per point
__global__ __launch_bounds__(256) void small_layer_bwd_kernel(
    const double* __restrict__ Xin, long P, long x_row0, int Din, int D, const double* __restrict__ Linv, const double* __restrict__ Scat,
    const double* __restrict__ u, const double* __restrict__ Z1, const double* __restrict__ Ct, const double* __restrict__ Tt,
    const double* __restrict__ Ek, const double* __restrict__ vbar, const double* __restrict__ mbar, double* __restrict__ Cbar,
    double* __restrict__ Gt, double* __restrict__ R1, double* __restrict__ X1, int pts_per_block) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sL = smem;                           // Linv [j][m]:     kb[m] = sum_j Linv[j][m] cb[j]
  double* sS = smem + SL_M * SL_M;             // W_d^T [n][m]:    (W_d t_d)[m] = sum_n W_d[m][n] t_d[n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* xs = smem + (1 + D) * SL_M * SL_M + wave * SL_Q * SL_M;
  const int w1 = Din + 1;
  for (int i = tid; i < SL_M * SL_M; i += 256) sL[i] = Linv[i];
  for (int i = tid; i < D * SL_M * SL_M; i += 256) sS[i] = Scat[i];
  __syncthreads();
  double ud[SL_MAXD];
#pragma unroll
  for (int d = 0; d < SL_MAXD; ++d) ud[d] = d < D ? u[(long)lane * D + d] : 0.0;
  const long p0 = (long)blockIdx.x * pts_per_block, p1 = min(P, p0 + pts_per_block);
  for (long pg = p0 + wave * SL_Q; pg < p1; pg += 4 * SL_Q) {
    double cb[SL_Q], vs[SL_Q], wt[SL_Q], g[SL_Q];
    long pq[SL_Q];
#pragma unroll
    for (int q = 0; q < SL_Q; ++q) { pq[q] = min(pg + q, p1 - 1); cb[q] = 0.0; vs[q] = 0.0; }
    for (int d = 0; d < D; ++d) {
      const double udd = d == 0 ? ud[0] : (d == 1 ? ud[1] : ud[2]);
#pragma unroll
      for (int q = 0; q < SL_Q; ++q) xs[q * SL_M + lane] = Tt[pq[q] * (long)D * SL_M + (long)d * SL_M + lane];
      sl_wave_fence();
      sl_matvec4(sS + d * SL_M * SL_M, xs, wt);
      sl_wave_fence();
#pragma unroll
      for (int q = 0; q < SL_Q; ++q) {
        const double vb = vbar[pq[q] * D + d], mb = mbar[pq[q] * D + d];
        cb[q] += mb * udd + 2.0 * vb * wt[q];
        vs[q] += vb;
      }
    }
#pragma unroll
    for (int q = 0; q < SL_Q; ++q) {
      cb[q] -= 2.0 * vs[q] * Ct[pq[q] * SL_M + lane];
      if (pg + q < p1) Cbar[pq[q] * SL_M + lane] = cb[q];
      xs[q * SL_M + lane] = cb[q];
    }
    sl_wave_fence();
    sl_matvec4(sL, xs, g);
    sl_wave_fence();
#pragma unroll
    for (int q = 0; q < SL_Q; ++q) {
      g[q] *= Ek[pq[q] * SL_M + lane];
      if (pg + q < p1) Gt[pq[q] * SL_M + lane] = g[q];
    }
    for (int j = 0; j < w1; ++j) {
      const double z1 = Z1[(long)lane * w1 + j];
      double r[SL_Q];
#pragma unroll
      for (int q = 0; q < SL_Q; ++q) r[q] = g[q] * z1;
      sl_wave_sum4(r);
      if (lane < SL_Q && pg + lane < p1) {
        const long p = pg + lane;
        R1[p * w1 + j] = lane == 0 ? r[0] : (lane == 1 ? r[1] : (lane == 2 ? r[2] : r[3]));
        X1[p * w1 + j] = j < Din ? Xin[(x_row0 + p) * Din + j] : 1.0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- backward, sums over points
// G_d += sum_p vbar_pd c_p c_p^T,  Q' += sum_p cb_p c_p^T,  du += sum_p c_p mbar_p^T,  GX += sum_p g_p [x_p | 1]^T.
// A block takes `pts_per_block` points in tiles of 64 staged in LDS; thread (ty, tx) owns the 4 x 4 block (4 ty.., 4 tx..) of
// every 64 x 64 output; the small outputs are spread over the threads; everything meets in fp64 atomics.
__global__ __launch_bounds__(256) void small_layer_sums_kernel(
    long P, int Din, int D, const double* __restrict__ Ct, const double* __restrict__ Cbar, const double* __restrict__ Gt,
    const double* __restrict__ vbar, const double* __restrict__ mbar, const double* __restrict__ X1, double* __restrict__ ws,
    long ws_stride, int pts_per_block) {
  __shared__ __attribute__((aligned(16))) double smem[3 * 64 * SL_M + 2 * 64 * SL_MAXD + 64 * 33];
  double* sC = smem;                       // [64 points][64]
  double* sB = sC + 64 * SL_M;             // Cbar
  double* sG = sB + 64 * SL_M;             // Gt
  double* sV = sG + 64 * SL_M;             // vbar [64][D]
  double* sM = sV + 64 * SL_MAXD;          // mbar [64][D]
  double* sX = sM + 64 * SL_MAXD;          // X1 [64][w1]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int w1 = Din + 1;
  double aG[SL_MAXD][4][4], aQ[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      aQ[i][j] = 0.0;
#pragma unroll
      for (int d = 0; d < SL_MAXD; ++d) aG[d][i][j] = 0.0;
    }
  // small outputs: du [64][D] and GX [64][w1]: output o = tid, tid + 256, ... of the concatenated list
  const int n_small = SL_M * (D + w1);
  double aS[9];                            // ceil(64 * (3 + 33) / 256) = 9
#pragma unroll
  for (int q = 0; q < 9; ++q) aS[q] = 0.0;
  const long p0 = (long)blockIdx.x * pts_per_block, p1 = min(P, p0 + pts_per_block);
  for (long pt = p0; pt < p1; pt += 64) {
    const int np = (int)min((long)64, p1 - pt);
    __syncthreads();
    for (int i = tid; i < 64 * SL_M; i += 256) {
      const int r = i >> 6;
      const bool ok = r < np;
      sC[i] = ok ? Ct[pt * SL_M + i] : 0.0;
      sB[i] = ok ? Cbar[pt * SL_M + i] : 0.0;
      sG[i] = ok ? Gt[pt * SL_M + i] : 0.0;
    }
    for (int i = tid; i < 64 * D; i += 256) {
      const int r = i / D, d = i - r * D;
      const bool ok = r < np;
      sV[r * SL_MAXD + d] = ok ? vbar[pt * D + i] : 0.0;
      sM[r * SL_MAXD + d] = ok ? mbar[pt * D + i] : 0.0;
    }
    for (int i = tid; i < 64 * w1; i += 256) sX[i] = (i / w1) < np ? X1[pt * w1 + i] : 0.0;
    __syncthreads();
    for (int p = 0; p < np; ++p) {
      double ci[4], cj[4], bi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { ci[i] = sC[p * SL_M + 4 * ty + i]; cj[i] = sC[p * SL_M + 4 * tx + i]; bi[i] = sB[p * SL_M + 4 * ty + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) aQ[i][j] = fma(bi[i], cj[j], aQ[i][j]);
#pragma unroll
      for (int d = 0; d < SL_MAXD; ++d) {
        if (d < D) {
          const double v = sV[p * SL_MAXD + d];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const double vc = v * ci[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) aG[d][i][j] = fma(vc, cj[j], aG[d][i][j]);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int o = tid + 256 * q;
      if (o < n_small) {
        const int m = o & 63, col = o >> 6;           // col < D: du column; else GX column col - D
        double a = 0.0;
        if (col < D) { for (int p = 0; p < np; ++p) a = fma(sC[p * SL_M + m], sM[p * SL_MAXD + col], a); }
        else { const int j = col - D; for (int p = 0; p < np; ++p) a = fma(sG[p * SL_M + m], sX[p * w1 + j], a); }
        aS[q] += a;
      }
    }
  }
  // this block's partial sums, one slab per block: [D][64][64] G, [64][64] Q', [64][D] du, [64][w1] GX (no atomics: with every
  // block adding to the same 8-20 thousand addresses the atomic adds were two thirds of the kernel's time)
  double* __restrict__ out = ws + (long)blockIdx.x * ws_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * ty + i;
#pragma unroll
    for (int d = 0; d < SL_MAXD; ++d)
      if (d < D) {
        const d4_t v = {aG[d][i][0], aG[d][i][1], aG[d][i][2], aG[d][i][3]};
        *reinterpret_cast<d4_t*>(out + (long)d * SL_M * SL_M + r * SL_M + 4 * tx) = v;
      }
    const d4_t v = {aQ[i][0], aQ[i][1], aQ[i][2], aQ[i][3]};
    *reinterpret_cast<d4_t*>(out + (long)D * SL_M * SL_M + r * SL_M + 4 * tx) = v;
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int o = tid + 256 * q;
    if (o < n_small) {
      const int m = o & 63, col = o >> 6;
      const long base = (long)(D + 1) * SL_M * SL_M;
      if (col < D) out[base + (long)m * D + col] = aS[q];
      else out[base + (long)SL_M * D + (long)m * w1 + (col - D)] = aS[q];
    }
  }
}

// acc += sum over the blocks' slabs: one thread per output value, the blocks in a fixed order
__global__ __launch_bounds__(256) void small_layer_sums_reduce_kernel(const double* __restrict__ ws, long ws_stride, int nblocks, int Din, int D,
                                                                      double* __restrict__ accG, double* __restrict__ accQ,
                                                                      double* __restrict__ accdu, double* __restrict__ accGX) {
  const int w1 = Din + 1;
  const long nG = (long)D * SL_M * SL_M, nQ = (long)SL_M * SL_M, ndu = (long)SL_M * D, nGX = (long)SL_M * w1;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= nG + nQ + ndu + nGX) return;
  double sa[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sa[i] = 0.0;
  int b = 0;
  for (; b + 8 <= nblocks; b += 8) {                 // eight independent loads in flight, fixed association
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = ws[(long)(b + i) * ws_stride + o];
#pragma unroll
    for (int i = 0; i < 8; ++i) sa[i] += x[i];
  }
  for (; b < nblocks; ++b) sa[0] += ws[(long)b * ws_stride + o];
  const double v = ((sa[0] + sa[1]) + (sa[2] + sa[3])) + ((sa[4] + sa[5]) + (sa[6] + sa[7]));
  if (o < nG) accG[o] += v;
  else if (o < nG + nQ) accQ[o - nG] += v;
  else if (o < nG + nQ + ndu) accdu[o - nG - nQ] += v;
  else accGX[o - nG - nQ - ndu] += v;
}

static int pts_per_block_for(long P) {
  // a block's staging of the layer's matrices (64-128 KB from L2) is amortised over its points; 4 waves x SL_Q points per
  // pass: at least one pass, at most ~2 blocks per CU over the whole launch
  long ppb = 4 * SL_Q;
  while (ppb < 4096 && (P + ppb - 1) / ppb > 512) ppb += 4 * SL_Q;
  return (int)ppb;
}

// ---------------------------------------------------------------------------------------------------- the layer's small matrices
// One workgroup, one launch, for a layer with Mp = 64: what prep_layer (dgp_ctx.h) issues as 11 dependent launches -
// pack_q, Kuu (+ jitter), Cholesky + inverse (the register-resident leaf, wave 0), Linv^T, W_d = Lu^-1 L_q,d, u = Lu^-1 q_mu,
// KL (layers.py:280-308), W_d^T stacked and [Z | 1] for the backward pass.  The 64 x 64 operands of the products stay in LDS
// between the steps; every output array the rest of the pass reads is written exactly as the separate kernels write it.
struct SmallPrepArgs {
  const double *q_sqrt, *q_mu, *Z, *var, *ls;
  int M, D, Din, kind, white, train;
  double *Lq, *qmu_p, *Kuu, *Euu, *Lu, *Linv, *LinvT, *Wcat, *u, *Scat, *Z1, *kl_out;
  int* info;
};

__global__ __launch_bounds__(256) void small_prep_kernel(SmallPrepArgs a) {
  __shared__ double sLi[SL_M][SL_M + 1];       // Linv (pitch 65: rows 4 ty + i of the 16 ty are on different banks)
  __shared__ double sLq[SL_M][SL_M];           // L_q,d
  __shared__ double sU[SL_M * SL_MAXD];        // q_mu padded [64][D]
  __shared__ double sred[4];
  const int tid = threadIdx.x, M = a.M, D = a.D, Din = a.Din;
  const int Mp = SL_M;
  // 1. L_q (padded: identity), q_mu (padded: zero)
  for (int idx = tid; idx < D * Mp * Mp; idx += 256) {
    const int d = idx / (Mp * Mp), i = (idx / Mp) % Mp, j = idx % Mp;
    double v;
    if (i < M && j < M) v = (j <= i) ? a.q_sqrt[((long)d * M + i) * M + j] : 0.0;
    else v = (i == j) ? 1.0 : 0.0;
    a.Lq[idx] = v;
  }
  for (int idx = tid; idx < Mp * D; idx += 256) {
    const int i = idx / D, d = idx % D;
    const double v = (i < M) ? a.q_mu[(long)i * D + d] : 0.0;
    a.qmu_p[idx] = v;
    sU[idx] = v;
  }
  // 2. Kuu + jitter (padding: identity) -> Kuu and the Cholesky's work copy Lu
  for (int idx = tid; idx < Mp * Mp; idx += 256) {
    const int m = idx / Mp, n = idx % Mp;
    double v, e = 0.0;
    if (m < M && n < M) {
      double r2 = 0.0;
      for (int j = 0; j < Din; ++j) {
        const double d = (a.Z[(long)m * Din + j] - a.Z[(long)n * Din + j]) / a.ls[j];
        r2 += d * d;
      }
      stationary_k(a.kind, a.var[0], r2, v, e);
      v += (m == n ? kJitter : 0.0);
    } else {
      v = (m == n) ? 1.0 : 0.0;
    }
    a.Kuu[idx] = v;
    a.Lu[idx] = v;
    if (a.Euu) a.Euu[idx] = e;
  }
  __syncthreads();
  // 3. Lu = chol(Kuu), Linv = Lu^-1: the leaf of potrf_inv (same choice of leaf as the host dispatcher makes), on wave 0
  if (tid < 64) {
    const VIdx b0{0u, 0u, 0u}, g1{1u, 1u, 1u}, t{(unsigned)tid, 0u, 0u}, bd{64u, 1u, 1u};
    if (M <= 32) ChainBody<CK_LEAF32>::run<CM_CHAIN_WAVE>(b0, g1, t, bd, a.Lu, a.Linv, Mp, (long)Mp * Mp, 0, M, 1, a.info, Mp);
    else ChainBody<CK_LEAF64>::run<CM_CHAIN_WAVE>(b0, g1, t, bd, a.Lu, a.Linv, Mp, (long)Mp * Mp, 0, Mp, 1, a.info, 0);
  }
  __syncthreads();
  // 4. Linv -> LDS, Linv^T -> memory
  for (int idx = tid; idx < Mp * Mp; idx += 256) {
    const int i = idx / Mp, j = idx % Mp;
    const double v = a.Linv[idx];
    sLi[i][j] = v;
    a.LinvT[(long)j * Mp + i] = v;
  }
  __syncthreads();
  // 5. W_d = Linv L_q,d (white: L_q,d), its transpose stacked, 0.5 |W|^2 over the M x M part;  u = Linv q_mu (white: q_mu)
  const int tx = tid & 15, ty = tid >> 4;
  double kl = 0.0;
  for (int d = 0; d < D; ++d) {
    for (int idx = tid; idx < Mp * Mp; idx += 256) sLq[idx / Mp][idx % Mp] = a.Lq[(long)d * Mp * Mp + idx];
    __syncthreads();
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    if (a.white) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = sLq[4 * ty + i][4 * tx + j];
    } else {
      for (int k = 0; k < Mp; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { av[i] = sLi[4 * ty + i][k]; bv[i] = sLq[k][4 * tx + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fma(av[i], bv[j], acc[i][j]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = 4 * ty + i, n = 4 * tx + j;
        const double w = acc[i][j];
        a.Wcat[(long)m * D * Mp + (long)d * Mp + n] = w;
        if (a.Scat) a.Scat[((long)d * Mp + n) * Mp + m] = w;
        if (m < M && n < M) kl += 0.5 * w * w;
      }
    __syncthreads();
  }
  for (int idx = tid; idx < Mp * D; idx += 256) {
    const int m = idx / D, d = idx % D;
    double uv;
    if (a.white) uv = sU[idx];
    else {
      uv = 0.0;
      for (int k = 0; k < Mp; ++k) uv = fma(sLi[m][k], sU[k * D + d], uv);
    }
    a.u[idx] = uv;
    if (m < M) kl += 0.5 * uv * uv;
  }
  // 6. the rest of the KL: -0.5 sum log L_q,ii^2 (+ D sum log Lu_ii) - 0.5 D M
  for (int idx = tid; idx < D * M; idx += 256) {
    const int d = idx / M, i = idx % M;
    const double q = a.Lq[((long)d * Mp + i) * Mp + i];
    kl -= 0.5 * log(q * q);
  }
  if (!a.white)
    for (int i = tid; i < M; i += 256) kl += (double)D * log(a.Lu[(long)i * Mp + i]);
  if (tid == 0) kl -= 0.5 * (double)D * (double)M;
  for (int o = 32; o > 0; o >>= 1) kl += __shfl_down(kl, o);
  if ((tid & 63) == 0) sred[tid >> 6] = kl;
  __syncthreads();
  if (tid == 0) unsafeAtomicAdd(a.kl_out, sred[0] + sred[1] + sred[2] + sred[3]);
  // 7. [Z | 1] (backward pass)
  if (a.Z1) {
    const int w1 = Din + 1;
    for (int idx = tid; idx < Mp * w1; idx += 256) {
      const int m = idx / w1, j = idx % w1;
      a.Z1[idx] = (m < M) ? (j < Din ? a.Z[(long)m * Din + j] : 1.0) : 0.0;
    }
  }
}

hipError_t small_prep(hipStream_t st, const double* q_sqrt, const double* q_mu, const double* Z, const double* var, const double* ls,
                      int M, int D, int Din, int kind, int white, int train, double* Lq, double* qmu_p, double* Kuu, double* Euu,
                      double* Lu, double* Linv, double* LinvT, double* Wcat, double* u, double* Scat, double* Z1, double* kl_out,
                      int* info) {
  SmallPrepArgs a;
  a.q_sqrt = q_sqrt; a.q_mu = q_mu; a.Z = Z; a.var = var; a.ls = ls;
  a.M = M; a.D = D; a.Din = Din; a.kind = kind; a.white = white; a.train = train;
  a.Lq = Lq; a.qmu_p = qmu_p; a.Kuu = Kuu; a.Euu = Euu; a.Lu = Lu; a.Linv = Linv; a.LinvT = LinvT; a.Wcat = Wcat; a.u = u;
  a.Scat = Scat; a.Z1 = Z1; a.kl_out = kl_out; a.info = info;
  hipLaunchKernelGGL(small_prep_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- the layer's gradient finish
// One workgroup, one launch, for a layer with Mp = 64: what finish_layer (dgp_abi.hip) issues as ~20 dependent launches - from
// the layer's sums over the points (G_d, Q', du, g^T [X | 1], sum x^2 rowsum(g), sum vbar) to its entries of d ELBO / d params:
//   dW_d = 2 sym(G_d) W_d - W_d (lower, M x M),  du' = du - u,  Q = Lu^-T Q',
//   non-white: T1 = du' u^T + sum_d dW_d W_d^T, T2 = Lu^-T T1, dL_q,d = Lu^-T dW_d, dq_mu = Lu^-T du', dLu = -tril(Q + T2) - D diag(1 / Lu)
//   white:     dL_q,d = dW_d, dq_mu = du', dLu = -tril(Q)
//   dL_q,d += diag(1 / L_q,d) (KL),  Cholesky backward S = Lu^-T Phi(Lu^T dLu) Lu^-1,
//   kernel gradients through Kuf (from g^T [X | 1]) and through Kuu (from sym(S)), SURVEY App. C steps 5-6.
// The 64 x 64 operands live in LDS (four buffers of pitch 65); products: thread (ty, tx) owns a 4 x 4 block.
struct SmallFinishArgs {
  const double *Gd, *du, *Qp, *GX, *x2, *vsum;             // the layer's slice of the partial-sum buffer
  const double *Wcat, *u, *Linv, *Lu, *Lq, *Kuu, *Euu;     // the layer's small matrices
  const double *Z, *var, *ls;                              // parameters
  int M, D, Din, white;
  double *dLq, *dqmu_p;                                    // kept for the natural-gradient step
  double *gZ, *gls, *gvar, *gqsqrt, *gqmu;                 // the layer's entries of the gradient
  double* kdot;                                            // Matern kernels: sum_p dk_p . k_p  (null: squared exponential)
};

constexpr int SF_P = SL_M + 1;       // LDS pitch

// acc[i][j] = sum_k opA(A)[4 ty + i][k] * opB(B)[k][4 tx + j],  A, B: LDS [64][SF_P];  TA: A is read transposed, TB likewise
template <bool TA, bool TB>
__device__ __forceinline__ void sf_mm(const double* __restrict__ A, const double* __restrict__ B, int ty, int tx, double (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
#pragma unroll 4
  for (int k = 0; k < SL_M; ++k) {
    double av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      av[i] = TA ? A[k * SF_P + 4 * ty + i] : A[(4 * ty + i) * SF_P + k];
      bv[i] = TB ? B[(4 * tx + i) * SF_P + k] : B[k * SF_P + 4 * tx + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fma(av[i], bv[j], acc[i][j]);
  }
}
__device__ __forceinline__ void sf_store(double* __restrict__ C, int ty, int tx, const double (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) C[(4 * ty + i) * SF_P + 4 * tx + j] = acc[i][j];
}
__device__ __forceinline__ double sf_block_sum(double v, double* sh) {     // all 256 threads; result on every thread
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void small_finish_kernel(SmallFinishArgs a) {
  __shared__ double sLi[SL_M * SF_P], sX[SL_M * SF_P], sY[SL_M * SF_P], sZ[SL_M * SF_P];
  __shared__ double sDu[SL_M * SL_MAXD], sUu[SL_M * SL_MAXD], sh[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int M = a.M, D = a.D, Din = a.Din, Mp = SL_M, w1 = Din + 1;
  const bool white = a.white != 0;
  for (int idx = tid; idx < Mp * Mp; idx += 256) sLi[(idx / Mp) * SF_P + idx % Mp] = a.Linv[idx];
  for (int idx = tid; idx < Mp * D; idx += 256) {
    const int i = idx / D;
    const double uu = a.u[idx];
    sUu[idx] = uu;
    sDu[idx] = (i < M) ? a.du[idx] - uu : 0.0;             // du' (wbar_total)
  }
  __syncthreads();
  double t1[4][4], acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) t1[i][j] = 0.0;
  for (int d = 0; d < D; ++d) {
    // sX = sym(G_d) (the lower triangle is what the sums deliver), sY = W_d
    for (int idx = tid; idx < Mp * Mp; idx += 256) {
      const int i = idx / Mp, j = idx % Mp;
      sX[i * SF_P + j] = a.Gd[(long)d * Mp * Mp + (i >= j ? (long)i * Mp + j : (long)j * Mp + i)];
      sY[i * SF_P + j] = a.Wcat[(long)i * D * Mp + (long)d * Mp + j];
    }
    __syncthreads();
    sf_mm<false, false>(sX, sY, ty, tx, acc);
    __syncthreads();
    // dW_d = tril(2 G W) - W on the M x M part (wbar_total) -> sZ
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * ty + i, n = 4 * tx + j;
        sZ[k * SF_P + n] = (k < M && n < M && n <= k) ? 2.0 * acc[i][j] - sY[k * SF_P + n] : 0.0;
      }
    __syncthreads();
    if (!white) {
      sf_mm<false, true>(sZ, sY, ty, tx, acc);              // dW_d W_d^T
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t1[i][j] += acc[i][j];
      sf_mm<true, false>(sLi, sZ, ty, tx, acc);             // dL_q,d = Linv^T dW_d
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = sZ[(4 * ty + i) * SF_P + 4 * tx + j];
    }
    // lqbar_finish + unpack_q_grads
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * ty + i, c = 4 * tx + j;
        double v = 0.0;
        if (r < M && c <= r) v = acc[i][j] + (r == c ? 1.0 / a.Lq[((long)d * Mp + r) * Mp + r] : 0.0);
        a.dLq[((long)d * Mp + r) * Mp + c] = v;
        if (r < M && c < M) a.gqsqrt[((long)d * M + r) * M + c] = (c <= r) ? v : 0.0;
      }
    __syncthreads();
  }
  // dq_mu = Linv^T du' (white: du')  -> dqmu_p, gradient of q_mu
  for (int idx = tid; idx < Mp * D; idx += 256) {
    const int m = idx / D, d = idx % D;
    double v;
    if (white) v = sDu[idx];
    else {
      v = 0.0;
      for (int k = 0; k < Mp; ++k) v = fma(sLi[k * SF_P + m], sDu[k * D + d], v);
    }
    a.dqmu_p[idx] = v;
    if (m < M) a.gqmu[(long)m * D + d] = v;
  }
  // Q = Linv^T Q'
  for (int idx = tid; idx < Mp * Mp; idx += 256) sY[(idx / Mp) * SF_P + idx % Mp] = a.Qp[idx];
  __syncthreads();
  double q[4][4];
  sf_mm<true, false>(sLi, sY, ty, tx, q);
  // Matern: kdot = sum_{i >= j < M} Lu[i][j] Q[i][j]   (before Q becomes dLu)
  if (a.kdot) {
    double kd = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * ty + i, c = 4 * tx + j;
        if (r < M && c <= r) kd += a.Lu[(long)r * Mp + c] * q[i][j];
      }
    kd = sf_block_sum(kd, sh);
    if (tid == 0) a.kdot[0] = kd;
  }
  __syncthreads();
  if (!white) {
    // T1 += du' u^T, T2 = Linv^T T1
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double v = t1[i][j];
        for (int d = 0; d < D; ++d) v = fma(sDu[(4 * ty + i) * D + d], sUu[(4 * tx + j) * D + d], v);
        sX[(4 * ty + i) * SF_P + 4 * tx + j] = v;
      }
    __syncthreads();
    sf_mm<true, false>(sLi, sX, ty, tx, acc);               // T2
  }
  // dLu = -tril(Q + T2) - D diag(1 / Lu)   (lubar_finish) -> sZ;   Lu -> sX
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * ty + i, c = 4 * tx + j;
      double v = 0.0;
      if (r < M && c <= r) {
        v = -(q[i][j] + (white ? 0.0 : acc[i][j]));
        if (r == c && !white) v -= (double)D / a.Lu[(long)r * Mp + r];
      }
      sZ[r * SF_P + c] = v;
    }
  for (int idx = tid; idx < Mp * Mp; idx += 256) sX[(idx / Mp) * SF_P + idx % Mp] = a.Lu[idx];
  __syncthreads();
  // Cholesky backward: T3 = Phi(Lu^T dLu), T4 = T3 Linv, S = Linv^T T4
  sf_mm<true, false>(sX, sZ, ty, tx, acc);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * ty + i, c = 4 * tx + j;
      sY[r * SF_P + c] = (c < r) ? acc[i][j] : (r == c ? 0.5 * acc[i][j] : 0.0);
    }
  __syncthreads();
  sf_mm<false, false>(sY, sLi, ty, tx, acc);
  __syncthreads();
  sf_store(sX, ty, tx, acc);
  __syncthreads();
  sf_mm<true, false>(sLi, sX, ty, tx, acc);
  __syncthreads();
  sf_store(sZ, ty, tx, acc);                                // S (unsymmetrised dKuu)
  __syncthreads();
  // kernel gradients: through Kuf (rbf_kuf_bwd_finish: writes) and through Kuu (rbf_kuu_bwd: adds)
  const double v0 = a.var[0];
  double vacc = 0.0;                                        // sum_{m, n} sym(S) k0   (variance through Kuu)
  for (int j = 0; j < Din; ++j) {
    const double l = a.ls[j];
    double la = 0.0, va = 0.0, lk = 0.0;
    if (tid < M) {
      const int m = tid;
      const double cs = a.GX[(long)m * w1 + Din], gx = a.GX[(long)m * w1 + j], z = a.Z[(long)m * Din + j];
      double dz = -(z * cs - gx) / (l * l);
      la = z * z * cs - 2.0 * z * gx;
      va = cs;
      double zacc = 0.0;
      for (int n = 0; n < M; ++n) {
        const double k0 = a.Kuu[(long)m * Mp + n] - (m == n ? kJitter : 0.0);
        const double sym = 0.5 * (sZ[m * SF_P + n] + sZ[n * SF_P + m]);
        const double h = sym * (a.Euu ? a.Euu[(long)m * Mp + n] : k0);
        const double dl = z - a.Z[(long)n * Din + j];
        zacc += h * dl;
        lk += h * dl * dl;
        if (j == 0) vacc += sym * k0;
      }
      dz += -2.0 * zacc / (l * l);
      a.gZ[(long)m * Din + j] = dz;
    }
    const double lt = sf_block_sum(la, sh), lkt = sf_block_sum(lk, sh);
    if (tid == 0) a.gls[j] = (lt + a.x2[j]) / (l * l * l) + lkt / (l * l * l);
    if (j == 0) {
      const double vt = sf_block_sum(va, sh), vk = sf_block_sum(vacc, sh);
      if (tid == 0) a.gvar[0] = (a.kdot ? a.kdot[0] : vt) / v0 + a.vsum[0] + vk / v0;
    }
  }
}

hipError_t small_finish(hipStream_t st, const double* Gd, const double* du, const double* Qp, const double* GX, const double* x2,
                        const double* vsum, const double* Wcat, const double* u, const double* Linv, const double* Lu,
                        const double* Lq, const double* Kuu, const double* Euu, const double* Z, const double* var,
                        const double* ls, int M, int D, int Din, int white, double* dLq, double* dqmu_p, double* gZ, double* gls,
                        double* gvar, double* gqsqrt, double* gqmu, double* kdot) {
  SmallFinishArgs a;
  a.Gd = Gd; a.du = du; a.Qp = Qp; a.GX = GX; a.x2 = x2; a.vsum = vsum;
  a.Wcat = Wcat; a.u = u; a.Linv = Linv; a.Lu = Lu; a.Lq = Lq; a.Kuu = Kuu; a.Euu = Euu;
  a.Z = Z; a.var = var; a.ls = ls;
  a.M = M; a.D = D; a.Din = Din; a.white = white;
  a.dLq = dLq; a.dqmu_p = dqmu_p; a.gZ = gZ; a.gls = gls; a.gvar = gvar; a.gqsqrt = gqsqrt; a.gqmu = gqmu; a.kdot = kdot;
  hipLaunchKernelGGL(small_finish_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- natural-gradient step
// gpflow.optimizers.NaturalGradient(gamma) on one (q_mu, q_sqrt) pair of a layer with Mp = 64 (dgp.py:312-322,343; closed
// form of SURVEY App. B), one workgroup per output d: what dgp_natgrad_step issues as 12 launches per layer -
//   Li = L_q^-1,  dSigma = Li^T Phi(L_q^T dL_q) Li,  Sigma'^-1 = Li^T Li - 2 gamma sym(dSigma)   (loss = -ELBO),
//   Ri = chol(Sigma'^-1)^-1,  Sigma' = Ri^T Ri,  mu' = mu + gamma Sigma' dmu,  L_q' = chol(Sigma').
// The three factorisations / inversions are the register-resident leaf on wave 0 (operands through the scratch in memory).
struct SmallNatArgs {
  double* Lq;                 // [D][64][64] in / out
  const double* dLq;          // [D][64][64]
  const double* dqmu_p;       // [64][D]
  double* qmu_p;              // [64][D] in / out
  double *w0, *w1, *w2;       // scratch [D][64][64] each
  double *q_sqrt, *q_mu;      // parameters out: [D][M][M], [M][D]
  double gamma;
  int M, D;
  int* info;
};

__global__ __launch_bounds__(256) void small_natgrad_kernel(SmallNatArgs a) {
  __shared__ double sLi[SL_M * SF_P], sX[SL_M * SF_P], sY[SL_M * SF_P], sZ[SL_M * SF_P];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, d = blockIdx.x;
  const int M = a.M, D = a.D, Mp = SL_M;
  const long MM = (long)Mp * Mp;
  double* Lq = a.Lq + d * MM;
  const double* dLq = a.dLq + d * MM;
  double *Li = a.w0 + d * MM, *Pn = a.w1 + d * MM, *Ri = a.w2 + d * MM;
  const VIdx b0{0u, 0u, 0u}, g1{1u, 1u, 1u}, t{(unsigned)(tid & 63), 0u, 0u}, bd{64u, 1u, 1u};
  if (tid == 0) g_body_ext = reinterpret_cast<char*>(sZ);       // the leaf's transpose buffer: sZ whenever the leaf runs
  __syncthreads();
  // Li = L_q^-1
  if (tid < 64) ChainBody<CK_LEAF64>::run<CM_WAVE_EXT>(b0, g1, t, bd, Lq, Li, Mp, MM, 0, Mp, 0, a.info, 0);
  __syncthreads();
  for (int idx = tid; idx < Mp * Mp; idx += 256) {
    const int i = idx / Mp, j = idx % Mp;
    sLi[i * SF_P + j] = Li[idx];
    sX[i * SF_P + j] = Lq[idx];
    sY[i * SF_P + j] = dLq[idx];
  }
  __syncthreads();
  double acc[4][4], pinv[4][4];
  sf_mm<true, false>(sX, sY, ty, tx, acc);                  // L_q^T dL_q
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * ty + i, c = 4 * tx + j;
      sZ[r * SF_P + c] = (c < r) ? acc[i][j] : (r == c ? 0.5 * acc[i][j] : 0.0);       // Phi
    }
  __syncthreads();
  sf_mm<false, false>(sZ, sLi, ty, tx, acc);                // Phi Li
  __syncthreads();
  sf_store(sX, ty, tx, acc);
  __syncthreads();
  sf_mm<true, false>(sLi, sX, ty, tx, acc);                 // dSigma (unsymmetrised)
  sf_mm<true, false>(sLi, sLi, ty, tx, pinv);               // Sigma^-1
  __syncthreads();
  sf_store(sY, ty, tx, acc);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * ty + i, c = 4 * tx + j;
      Pn[(long)r * Mp + c] = pinv[i][j] + (-2.0 * a.gamma) * 0.5 * (sY[r * SF_P + c] + sY[c * SF_P + r]);
    }
  __syncthreads();
  // Ri = chol(Sigma'^-1)^-1
  if (tid < 64) ChainBody<CK_LEAF64>::run<CM_WAVE_EXT>(b0, g1, t, bd, Pn, Ri, Mp, MM, 0, Mp, 1, a.info, 0);
  __syncthreads();
  for (int idx = tid; idx < Mp * Mp; idx += 256) sX[(idx / Mp) * SF_P + idx % Mp] = Ri[idx];
  __syncthreads();
  sf_mm<true, false>(sX, sX, ty, tx, acc);                  // Sigma' = Ri^T Ri
  __syncthreads();
  sf_store(sZ, ty, tx, acc);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) Lq[(long)(4 * ty + i) * Mp + 4 * tx + j] = acc[i][j];
  __syncthreads();
  // mu' = mu + gamma Sigma' dmu   (loss = -ELBO)
  if (tid < M) {
    double sacc = 0.0;
    for (int n = 0; n < M; ++n) sacc += sZ[tid * SF_P + n] * a.dqmu_p[(long)n * D + d];
    const double mu = a.qmu_p[(long)tid * D + d] + a.gamma * sacc;
    a.qmu_p[(long)tid * D + d] = mu;
    a.q_mu[(long)tid * D + d] = mu;
  }
  __syncthreads();
  // L_q' = chol(Sigma')  (in place; its inverse goes to the scratch)
  if (tid < 64) ChainBody<CK_LEAF64>::run<CM_WAVE_EXT>(b0, g1, t, bd, Lq, Li, Mp, MM, 0, Mp, 1, a.info, 0);
  __syncthreads();
  for (int idx = tid; idx < M * M; idx += 256) {
    const int i = idx / M, j = idx % M;
    a.q_sqrt[(long)d * M * M + idx] = (j <= i) ? Lq[(long)i * Mp + j] : 0.0;
  }
}

hipError_t small_natgrad(hipStream_t st, double* Lq, const double* dLq, const double* dqmu_p, double* qmu_p, double* w0, double* w1,
                         double* w2, double* q_sqrt, double* q_mu, double gamma, int M, int D, int* info) {
  SmallNatArgs a;
  a.Lq = Lq; a.dLq = dLq; a.dqmu_p = dqmu_p; a.qmu_p = qmu_p; a.w0 = w0; a.w1 = w1; a.w2 = w2; a.q_sqrt = q_sqrt; a.q_mu = q_mu;
  a.gamma = gamma; a.M = M; a.D = D; a.info = info;
  hipLaunchKernelGGL(small_natgrad_kernel, dim3(D), dim3(256), 0, st, a);
  return hipGetLastError();
}

// LDS of the per-point kernels: the layer's (1 + D) matrices + the waves' vectors; above 64 KB the kernels' limit has to be
// raised (hipFuncSetAttribute) - per DEVICE (the attribute belongs to the device's copy of the function), and a refusal
// makes the fused path unavailable there (small_layer_ok false -> the layer runs on the general kernels) instead of an
// opaque launch failure later.
static size_t sl_point_lds(int D) { return (size_t)((1 + D) * SL_M * SL_M + 4 * SL_Q * SL_M) * 8; }
static bool sl_lds_limit_raised() {
  constexpr int kMaxDev = 64;
  static signed char state[kMaxDev];            // 0 unknown, 1 raised, -1 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return false;
  if (state[dev] == 0) {
    const int want = (int)sl_point_lds(SL_MAXD);
    const bool ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(small_layer_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(small_layer_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    state[dev] = ok ? 1 : -1;
  }
  return state[dev] > 0;
}

bool small_layer_ok(int kind, int Mp, int D, int Din) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("DGP_SMALL_FUSED"); enabled = e ? atoi(e) : 1; }
  return enabled && kind >= 0 && kind <= 2 && Mp == SL_M && D >= 1 && D <= SL_MAXD && Din >= 1 && Din <= 32 && sl_lds_limit_raised();
}

hipError_t small_layer_fwd(hipStream_t st, int kind, const double* Xin, long P, long x_row0, int Din, const double* Z,
                           const double* var, const double* ls, int M, int D, const double* LinvT, const double* Wcat,
                           const double* u, double* Kt, double* Et, double* Ct, double* Tt, double* cn, double* tn,
                           long pstride, double* mean0) {
  if (P <= 0) return hipSuccess;
  if (!sl_lds_limit_raised()) return hipErrorInvalidValue;
  const int ppb = pts_per_block_for(P);
  hipLaunchKernelGGL(small_layer_fwd_kernel, dim3((unsigned)((P + ppb - 1) / ppb)), dim3(256), sl_point_lds(D), st, kind, Xin, P, x_row0, Din, Z,
                     var, ls, M, D, LinvT, Wcat, u, Kt, Et, Ct, Tt, cn, tn, pstride, mean0, ppb);
  return hipGetLastError();
}

hipError_t small_layer_bwd(hipStream_t st, const double* Xin, long P, long x_row0, int Din, int D, const double* Linv,
                           const double* Scat, const double* u, const double* Z1, const double* Ct, const double* Tt,
                           const double* Ek, const double* vbar, const double* mbar, double* Cbar, double* Gt, double* R1,
                           double* X1) {
  if (P <= 0) return hipSuccess;
  if (!sl_lds_limit_raised()) return hipErrorInvalidValue;
  const int ppb = pts_per_block_for(P);
  hipLaunchKernelGGL(small_layer_bwd_kernel, dim3((unsigned)((P + ppb - 1) / ppb)), dim3(256), sl_point_lds(D), st, Xin, P, x_row0, Din, D, Linv,
                     Scat, u, Z1, Ct, Tt, Ek, vbar, mbar, Cbar, Gt, R1, X1, ppb);
  return hipGetLastError();
}

long small_layer_sums_ws_doubles() { return 256L * ((SL_MAXD + 1L) * SL_M * SL_M + SL_M * (SL_MAXD + 33L)); }

hipError_t small_layer_sums(hipStream_t st, long P, int Din, int D, const double* Ct, const double* Cbar, const double* Gt,
                            const double* vbar, const double* mbar, const double* X1, double* accG, double* accQ, double* accdu,
                            double* accGX, double* ws) {
  if (P <= 0) return hipSuccess;
  // tiles of 64 points, at most 256 blocks (the scratch holds 256 slabs)
  long ppb = 64;
  while ((P + ppb - 1) / ppb > 256) ppb += 64;
  const int nblocks = (int)((P + ppb - 1) / ppb);
  const long stride = (long)(D + 1) * SL_M * SL_M + (long)SL_M * (D + Din + 1);
  hipLaunchKernelGGL(small_layer_sums_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, P, Din, D, Ct, Cbar, Gt, vbar, mbar, X1, ws,
                     stride, (int)ppb);
  hipLaunchKernelGGL(small_layer_sums_reduce_kernel, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, st, ws, stride, nblocks, Din, D,
                     accG, accQ, accdu, accGX);
  return hipGetLastError();
}

}  // namespace dgp
