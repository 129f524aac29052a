// Small-matrix (M x M, batched) kernels of the SVGP layer: Kuu assembly, Cholesky, triangular
// inverse, KL, and the element-wise pieces of the hand-derived backward chain and of the
// natural-gradient step.  Reference: layers.py:227-234 (Kuu + jitter, Cholesky), layers.py:280-308
// (KL), gpflow NaturalGradient [ext] (dgp.py:312,343).  Latency-bound work: M <= 1024.
#include "dgp_internal.h"
#include "chain_bodies.h"

namespace dgp {

#define LAUNCH_CHECK() return hipGetLastError()

// ---------------------------------------------------------------------------------------- Kuu
hipError_t rbf_kuu(hipStream_t st, int kind, const double* Z, const double* var, const double* ls, int M, int Mp, int Din,
                   double* Kuu, double* Euu) {
  const long n = (long)Mp * Mp;
  (void)chain_launch<rbf_kuu_kernel>(st, dim3((unsigned)((n + 255) / 256)), kind, Z, var, ls, M, Mp, Din, Kuu, Euu);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- Cholesky + L^-1
// Recursive blocked factorisation driven from the host, all O(n^3) work in the MFMA GEMM engine:
//   A = [A11 . ; A21 A22]:  (L11, X11) = potrf_inv(A11);  L21 = A21 X11^T;  A22 -= L21 L21^T;
//   (L22, X22) = potrf_inv(A22);  X21 = -X22 (L21 X11)
// Leaves (n <= 64) are factored and inverted in LDS by one workgroup.  Replaces tf.linalg.cholesky /
// triangular_solve on Kuu (layers.py:231,245-247) and GPflow's chol(-2 nat2), its inverse and chol(S).
constexpr int LEAF = 64;

// One wave per matrix, the matrix in REGISTERS: lane i owns row i (64 doubles), every loop is unrolled so that the
// row is addressed with compile-time indices, and a value of another row travels through v_readlane (an SGPR
// operand of the FMA) instead of LDS.  The LDS version of this kernel (dot products over LDS rows, one barrier per
// column) took ~100 us per 64 x 64 block and four of them sit on the critical path of every Kuu factorisation.
//   Cholesky: right-looking; column c is scaled by 1/sqrt(pivot) and its outer product leaves the trailing rows.
//   Inverse : rows of L are scaled to a unit diagonal (L' = D^-1 L), X' = L'^-1 by forward substitution IN PLACE
//             (entry k of a row is consumed by step k and replaced by the entry of X'), then X = X' D^-1.
// Global reads and writes go row by row with the lanes along the row (coalesced) through an LDS transpose.

// LF = 64: the leaf of the recursion.  LF = 32: models with at most 32 inducing points (the Bayesian-optimisation
// surrogates; Mp = 64 with an identity padding): a quarter of the dependent readlane / fma chain (57 -> 16 us), the padding
// rows [n, npad) of X are written as identity rows.
// LF = 64 on FOUR waves (round 4).  The one-wave leaf above walks 2 x 2016 dependent (v_readlane, fma) pairs: 52 us, and four leaves
// sit on the critical path of every Kuu factorisation - at the head of every iteration, where nothing can overlap them.  Here thread
// (row i = t % 64, column group q = t / 64) keeps a[m] = A[i][q + 4 m]: a column step costs 16 FMAs per thread instead of up to 63;
// the column being eliminated travels through LDS (double buffered: one barrier per step), a row of X' through v_readlane inside
// each wave.  Every element sees the same FMAs in the same order as in the one-wave leaf: the results are bit-identical.
template <int B, int E, class F>
__device__ __forceinline__ void sm_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    sm_static_for<B + 1, E>(f);
  }
}
template <int Q>
__device__ __forceinline__ void leaf64_coop_body(double* __restrict__ A, double* __restrict__ X, int ld, int n, int do_chol, int* info,
                                                 double (&T)[64][65], double* Lc, double* Dg) {
  const int tid = threadIdx.x, i = tid & 63;
  double a[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int j = Q + 4 * m;
    const double v = T[i][j];
    a[m] = (i < n && j < n) ? (j <= i ? v : 0.0) : (i == j ? 1.0 : 0.0);     // rows and columns beyond n: an identity block
  }
  if (do_chol) {
    bool bad = false;
    sm_static_for<0, 64>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value, qc = c & 3, mc = c >> 2;
      if constexpr (Q == qc) {
        const double piv = lane_bcast(a[mc], c);
        // 1/sqrt(pivot): hardware estimate + two Newton steps (as the one-wave leaf)
        double rs = __builtin_amdgcn_rsq(piv);
        rs = rs * (1.5 - 0.5 * piv * rs * rs);
        rs = rs * (1.5 - 0.5 * piv * rs * rs);
        if (!(piv > 0.0)) { rs = nan(""); bad = true; }
        const double l = (i >= c) ? a[mc] * rs : 0.0;       // column c of L (zero above the diagonal)
        a[mc] = l;
        Lc[(c & 1) * 64 + i] = l;
        if (i == c) Dg[c] = l;
      }
      __syncthreads();
      const double li = Lc[(c & 1) * 64 + i];
#pragma unroll
      for (int m = 0; m < 16; ++m)
        if (Q + 4 * m > c) a[m] = fma(-li, Lc[(c & 1) * 64 + Q + 4 * m], a[m]);   // only j <= i is meaningful
    });
    if (bad && i == 0) atomicOr(info, 1);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; ++m) T[i][Q + 4 * m] = a[m];
    __syncthreads();
    for (int r = tid >> 6; r < n; r += 4)
      if (i < n) A[(long)r * ld + i] = T[r][i];
  }
  __syncthreads();
  const double dinv = 1.0 / Dg[i];
#pragma unroll
  for (int m = 0; m < 16; ++m) a[m] = (i > Q + 4 * m) ? a[m] * dinv : 0.0;      // strictly lower part of L' = D^-1 L
  sm_static_for<0, 64>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value, qk = k & 3, mk = k >> 2;
    if constexpr (Q == qk) Lc[(k & 1) * 64 + i] = a[mk];                        // L'[i][k] for the rows below k, 0 for the others
    __syncthreads();
    const double mi = Lc[(k & 1) * 64 + i];
#pragma unroll
    for (int m = 0; m < 16; ++m)
      if (Q + 4 * m < k) a[m] = fma(-mi, lane_bcast(a[m], k), a[m]);          // row k of X' is final in lane k of every wave
    if constexpr (Q == qk) a[mk] = -mi;
  });
  __syncthreads();
  if (tid < 64) Lc[tid] = dinv;                // (1 / diagonal per row, for the columns)
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int j = Q + 4 * m;
    T[i][j] = (i > j) ? a[m] * Lc[j] : (i == j ? dinv : 0.0);
  }
  __syncthreads();
  for (int r = tid >> 6; r < n; r += 4)
    if (i < n) X[(long)r * ld + i] = T[r][i];
}

__global__ __launch_bounds__(256) void leaf64_coop_kernel(double* Aall, double* Xall, int ld, long stride, int off, int n, int do_chol, int* info) {
  __shared__ double T[64][65];
  __shared__ double Lc[128];
  __shared__ double Dg[64];
  double* A = Aall + (long)blockIdx.x * stride + (long)off * ld + off;
  double* X = Xall + (long)blockIdx.x * stride + (long)off * ld + off;
  const int tid = threadIdx.x, i = tid & 63;
  for (int r = tid >> 6; r < 64; r += 4) T[r][i] = (r < n && i < n) ? A[(long)r * ld + i] : 0.0;
  if (tid < 64) Dg[tid] = (tid < n) ? A[(long)tid * ld + tid] : 1.0;       // (the diagonal of L when only the inverse is wanted)
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: leaf64_coop_body<0>(A, X, ld, n, do_chol, info, T, Lc, Dg); break;
    case 1: leaf64_coop_body<1>(A, X, ld, n, do_chol, info, T, Lc, Dg); break;
    case 2: leaf64_coop_body<2>(A, X, ld, n, do_chol, info, T, Lc, Dg); break;
    default: leaf64_coop_body<3>(A, X, ld, n, do_chol, info, T, Lc, Dg); break;
  }
}

// batched strided block copy: dst[b][r][c] = src[b][r][c] for an nr x nc block (leading dimension ld, batch stride)
static hipError_t sub_gemm(hipStream_t st, GemmOp op, long M, long N, long K, const double* A, const double* B, double* C,
                           int ld, long stride, int batch, double alpha, int beta, long sA = -1, long sB = -1) {
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.lda = g.ldb = g.ldc = ld; g.M = M; g.N = N; g.K = K;
  g.sA = sA < 0 ? stride : sA; g.sB = sB < 0 ? stride : sB; g.sC = stride;
  g.batch = batch; g.splits = 1; g.ksplit = 0; g.alpha = alpha; g.beta = beta; g.tri = TRI_NONE; g.triblk = 0;
  return gemm_f64(st, op, g);
}

// A (lower part used; overwritten by L, upper zeroed), X = L^-1; tmp: scratch of the same shape as A
static hipError_t potrf_inv_rec(hipStream_t st, double* A, double* X, double* tmp, int ld, long stride, int batch, int off,
                                int n, int do_chol, int* info) {
  hipError_t e;
  if (n <= LEAF) {
    static int coop = -1;
    if (coop < 0) { const char* e = getenv("DGP_LEAF_COOP"); coop = e ? atoi(e) : 1; }
    if (coop && !chain_recorder()) {      // (a recorded chain replays the one-wave body: chain.h)
      hipLaunchKernelGGL(leaf64_coop_kernel, dim3(batch), dim3(256), 0, st, A, X, ld, stride, off, n, do_chol, info);
      return hipGetLastError();
    }
    (void)chain_launch<leaf_potrf_inv_64_kernel>(st, dim3(batch), A, X, ld, stride, off, n, do_chol, info, 0);
    return hipGetLastError();
  }
  int n1 = ((n / 2 + 15) / 16) * 16;
  if (n1 > n - 16) n1 = n - 16;
  const int n2 = n - n1;
  double* A11 = A + (long)off * ld + off;           (void)A11;
  double* A21 = A + (long)(off + n1) * ld + off;
  double* A22 = A + (long)(off + n1) * ld + off + n1;
  double* X11 = X + (long)off * ld + off;
  double* X21 = X + (long)(off + n1) * ld + off;
  double* X22 = X + (long)(off + n1) * ld + off + n1;
  double* T = tmp + (long)(off + n1) * ld + off;    // [n2 x n1] scratch at the same position
  if ((e = potrf_inv_rec(st, A, X, tmp, ld, stride, batch, off, n1, do_chol, info)) != hipSuccess) return e;
  if (do_chol) {
    // L21 = A21 X11^T   (into T, then copy back into A21 by a second product-free pass: write directly)
    if ((e = sub_gemm(st, GEMM_NT, n2, n1, n1, A21, X11, T, ld, stride, batch, 1.0, 0)) != hipSuccess) return e;
    // A22 -= L21 L21^T
    if ((e = sub_gemm(st, GEMM_NT, n2, n2, n1, T, T, A22, ld, stride, batch, -1.0, 1)) != hipSuccess) return e;
    // A21 <- L21 : copy T -> A21 as  A21 = T * I  is wasteful; use a strided 2D copy per batch
    {
      const long nc = (long)n2 * n1;
      (void)chain_launch<copy_block_kernel>(st, dim3((unsigned)((nc + 255) / 256), batch), T, A21, ld, stride, n2, n1);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
  }
  if ((e = potrf_inv_rec(st, A, X, tmp, ld, stride, batch, off + n1, n2, do_chol, info)) != hipSuccess) return e;
  // X21 = -X22 (L21 X11)
  if ((e = sub_gemm(st, GEMM_NN, n2, n1, n1, A21, X11, T, ld, stride, batch, 1.0, 0)) != hipSuccess) return e;
  if ((e = sub_gemm(st, GEMM_NN, n2, n1, n2, X22, T, X21, ld, stride, batch, -1.0, 0)) != hipSuccess) return e;
  // clean upper-right blocks (structural zeros are relied upon by the triangular hints of the GEMMs)
  const long nz = (long)n1 * n2;
  if (do_chol)
    (void)chain_launch<zero_block_kernel>(st, dim3((unsigned)((nz + 255) / 256), batch), A, ld, stride, off, off + n1, n1, n2);
  (void)chain_launch<zero_block_kernel>(st, dim3((unsigned)((nz + 255) / 256), batch), X, ld, stride, off, off + n1, n1, n2);
  return hipGetLastError();
}

hipError_t potrf_inv(hipStream_t st, double* A, double* X, double* tmp, int Mp, int batch, int* info, int n_act) {
  if (Mp <= LEAF && n_act > 0 && n_act <= 32) {      // A = blockdiag(A[:n, :n], I): only the leading block is factorised
    (void)chain_launch<leaf_potrf_inv_32_kernel>(st, dim3(batch), A, X, Mp, (long)Mp * Mp, 0, n_act, 1, info, Mp);
    return hipGetLastError();
  }
  return potrf_inv_rec(st, A, X, tmp, Mp, (long)Mp * Mp, batch, 0, Mp, 1, info);
}

hipError_t trinv_lower(hipStream_t st, const double* L, double* X, double* tmp, int Mp, int batch) {
  return potrf_inv_rec(st, const_cast<double*>(L), X, tmp, Mp, (long)Mp * Mp, batch, 0, Mp, 0, nullptr);
}

// ---------------------------------------------------------------------------------------- packing
hipError_t pack_q(hipStream_t st, const double* q_sqrt, const double* q_mu, int M, int Mp, int D, double* Lq,
                  double* qmu_p) {
  const long n = (long)D * Mp * Mp;
  (void)chain_launch<pack_q_kernel>(st, dim3((unsigned)((n + 255) / 256)), q_sqrt, q_mu, M, Mp, D, Lq, qmu_p);
  LAUNCH_CHECK();
}

__global__ void store_q_kernel(const double* __restrict__ Lq, const double* __restrict__ qmu_p, int M, int Mp, int D,
                               double* __restrict__ q_sqrt, double* __restrict__ q_mu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long)D * M * M) {
    const int d = (int)(idx / ((long)M * M));
    const int i = (int)((idx / M) % M), j = (int)(idx % M);
    q_sqrt[idx] = (j <= i) ? Lq[((long)d * Mp + i) * Mp + j] : 0.0;
  }
  if (idx < (long)M * D) q_mu[idx] = qmu_p[idx];   // same [i][d] indexing for i < M
}

hipError_t store_q(hipStream_t st, const double* Lq, const double* qmu_p, int M, int Mp, int D, double* q_sqrt,
                   double* q_mu) {
  const long n = (long)D * M * M;
  hipLaunchKernelGGL(store_q_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Lq, qmu_p, M, Mp, D, q_sqrt,
                     q_mu);
  LAUNCH_CHECK();
}

hipError_t lq_to_wcat(hipStream_t st, const double* Lq, int Mp, int D, double* Wcat) {
  const long n = (long)D * Mp * Mp;
  (void)chain_launch<lq_to_wcat_kernel>(st, dim3((unsigned)((n + 255) / 256)), Lq, Mp, D, Wcat);
  LAUNCH_CHECK();
}

hipError_t copy_2d(hipStream_t st, const double* src, long lds, double* dst, long ldd, int nr, int nc) {
  const long n = (long)nr * nc;
  (void)chain_launch<copy_2d_kernel>(st, dim3((unsigned)((n + 255) / 256)), src, lds, dst, ldd, nr, nc);
  LAUNCH_CHECK();
}
hipError_t copy_mat(hipStream_t st, const double* src, double* dst, long n) {
  (void)chain_launch<copy_kernel>(st, dim3((unsigned)((n + 255) / 256)), src, dst, n);
  LAUNCH_CHECK();
}

hipError_t make_z1(hipStream_t st, const double* Z, int M, int Mp, int Din, double* Z1) {
  const long n = (long)Mp * (Din + 1);
  (void)chain_launch<make_z1_kernel>(st, dim3((unsigned)((n + 255) / 256)), Z, M, Mp, Din, Z1);
  LAUNCH_CHECK();
}

// writes (not accumulates) the Kuf + K_diag parts of d ELBO / d (Z, lengthscales, variance); one block per j
hipError_t rbf_kuf_bwd_finish(hipStream_t st, const double* GX, const double* x2rs, const double* vsum, const double* Z,
                              const double* var, const double* ls, int M, int Din, double* dZ, double* dls, double* dvar,
                              const double* kdot) {
  (void)chain_launch<rbf_kuf_bwd_finish_kernel>(st, dim3(Din), GX, x2rs, vsum, Z, var, ls, M, Din, dZ, dls, dvar, kdot);
  LAUNCH_CHECK();
}

hipError_t lower_dot(hipStream_t st, const double* Lu, const double* Q, int M, int Mp, double* out) {
  if (chain_recorder()) {
    (void)chain_launch<zero_doubles_kernel>(st, dim3(1), out, 1L);
  } else {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
    if (e != hipSuccess) return e;
  }
  (void)chain_launch<lower_dot_kernel>(st, dim3(Mp <= 64 ? 1 : 16), Lu, Q, M, Mp, out);
  LAUNCH_CHECK();
}

// WT[(d*Mp + j)*Mp + i] = Wcat[i*(D*Mp) + d*Mp + j]: the W_d^T stacked vertically ([D*Mp x Mp]), the B operand of
// dC = [2 vbar .* T] * WTcat when t_d = W_d^T c is kept from the forward pass
hipError_t wcat_transpose(hipStream_t st, const double* Wcat, int Mp, int D, double* WT) {
  (void)chain_launch<wcat_transpose_kernel>(st, dim3(Mp / 32, Mp / 32, D), Wcat, Mp, D, WT);
  LAUNCH_CHECK();
}

hipError_t sub_identity(hipStream_t st, double* S, int M, int Mp, int batch) {
  const long n = (long)batch * Mp * Mp;
  (void)chain_launch<sub_identity_kernel>(st, dim3((unsigned)((n + 255) / 256)), S, M, Mp, n);
  LAUNCH_CHECK();
}

// G (lower triangle valid) -> symmetric: tile (bi, bj) with bj > bi takes the transpose of tile (bj, bi) through LDS
// (coalesced reads and writes; the element-wise version read G column-wise right after the atomics of the Gram product
// had left it in HBM: 167 us for 8 x 256^2, on the critical path of the last layer's chain), diagonal tiles in place.
hipError_t symmetrize_lower(hipStream_t st, double* G, int Mp, int batch) {
  const int nb = Mp / 32;
  (void)chain_launch<symmetrize_lower_kernel>(st, dim3((unsigned)(nb * (nb + 1) / 2), (unsigned)batch), G, Mp);
  LAUNCH_CHECK();
}

hipError_t sub_scalars(hipStream_t st, const double* a, const double* b, double* out) {
  (void)chain_launch<sub_scalars_kernel>(st, dim3(1), a, b, out);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- KL
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;    // valid on thread 0
}

hipError_t layer_kl(hipStream_t st, const double* Wcat, const double* u, const double* Lq, const double* Lu, int M,
                    int Mp, int D, int white, double* kl_out) {
  (void)chain_launch<layer_kl_kernel>(st, dim3(Mp <= 64 ? 1 : 64), Wcat, u, Lq, Lu, M, Mp, D, white, kl_out);   // (grid-stride loops: one block covers a small layer)
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- backward chain pieces
hipError_t wbar_total(hipStream_t st, double* dWcat, const double* Wcat, double* du, const double* u, int M, int Mp,
                      int D) {
  const long n = (long)Mp * D * Mp;
  (void)chain_launch<wbar_total_kernel>(st, dim3((unsigned)((n + 255) / 256)), dWcat, Wcat, du, u, M, Mp, D);
  LAUNCH_CHECK();
}

hipError_t lqbar_finish(hipStream_t st, double* dLq, const double* Lq, int M, int Mp, int D) {
  const long n = (long)D * Mp * Mp;
  (void)chain_launch<lqbar_finish_kernel>(st, dim3((unsigned)((n + 255) / 256)), dLq, Lq, M, Mp, D);
  LAUNCH_CHECK();
}

hipError_t lubar_finish(hipStream_t st, double* dLu, const double* T2, const double* Lu, int M, int Mp, int D,
                        int white) {
  const long n = (long)Mp * Mp;
  (void)chain_launch<lubar_finish_kernel>(st, dim3((unsigned)((n + 255) / 256)), dLu, T2, Lu, M, Mp, D, white);
  LAUNCH_CHECK();
}

hipError_t phi_tril_halfdiag(hipStream_t st, double* T, int Mp, int batch) {
  const long n = (long)batch * Mp * Mp;
  (void)chain_launch<phi_tril_halfdiag_kernel>(st, dim3((unsigned)((n + 255) / 256)), T, Mp, n);
  LAUNCH_CHECK();
}

// RBF backward through Kuu = K(Z,Z) + jitter*I with dKuu = sym(S)   
hipError_t rbf_kuu_bwd(hipStream_t st, const double* S, const double* Kuu, const double* Euu, const double* Z,
                       const double* var, const double* ls, int M, int Mp, int Din, double* dZ, double* dls, double* dvar,
                       const double* white) {
  (void)chain_launch<rbf_kuu_bwd_kernel>(st, dim3(M, Din), S, Kuu, Euu, Z, var, ls, M, Mp, Din, dZ, dls, dvar, white);
  LAUNCH_CHECK();
}

hipError_t unpack_q_grads(hipStream_t st, const double* dLq, const double* dqmu_p, int M, int Mp, int D,
                          double* g_q_sqrt, double* g_q_mu) {
  const long n = (long)D * M * M;
  (void)chain_launch<unpack_q_grads_kernel>(st, dim3((unsigned)((n + 255) / 256)), dLq, dqmu_p, M, Mp, D, g_q_sqrt, g_q_mu);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- natural gradient pieces
__global__ void symmetrize_axpy_kernel(const double* __restrict__ Gm, const double* __restrict__ Pinv, double two_gamma,
                                       double* __restrict__ out, int Mp, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long b = idx / ((long)Mp * Mp);
  const int i = (int)((idx / Mp) % Mp), j = (int)(idx % Mp);
  const double* G = Gm + b * Mp * Mp;
  out[idx] = Pinv[idx] + two_gamma * 0.5 * (G[(long)i * Mp + j] + G[(long)j * Mp + i]);
}
hipError_t symmetrize_axpy(hipStream_t st, const double* Gm, const double* Pinv, double two_gamma, double* out, int Mp,
                           int batch) {
  const long n = (long)batch * Mp * Mp;
  hipLaunchKernelGGL(symmetrize_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Gm, Pinv, two_gamma,
                     out, Mp, n);
  LAUNCH_CHECK();
}

// g_qmu_p holds d ELBO / d q_mu (padded [Mp][D]); loss = -ELBO, so mu' = mu + gamma * Sn * g_elbo
__global__ void natgrad_mu_kernel(const double* __restrict__ Sn, const double* __restrict__ g, double gamma,
                                  double* __restrict__ qmu_p, int M, int Mp, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * D) return;
  const int m = (int)(idx / D), d = (int)(idx % D);
  const double* Sd = Sn + ((long)d * Mp + m) * Mp;
  double s = 0.0;
  for (int n = 0; n < M; ++n) s += Sd[n] * g[(long)n * D + d];
  qmu_p[(long)m * D + d] += gamma * s;
}
hipError_t natgrad_mu(hipStream_t st, const double* Sn, const double* g_qmu_p, double gamma, double* qmu_p, int M,
                      int Mp, int D) {
  const long n = (long)M * D;
  hipLaunchKernelGGL(natgrad_mu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Sn, g_qmu_p, gamma, qmu_p,
                     M, Mp, D);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- exact GP (gpflow GPR) pieces
// A[i][i] += v for i < M
__global__ void add_diag_kernel(double* __restrict__ A, int M, int Mp, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) A[(long)i * Mp + i] += v;
}
hipError_t add_diag(hipStream_t st, double* A, int M, int Mp, double v) {
  hipLaunchKernelGGL(add_diag_kernel, dim3((M + 255) / 256), dim3(256), 0, st, A, M, Mp, v);
  LAUNCH_CHECK();
}
// out[0] = -0.5 sum A^2 - Dy sum_i log L_ii - 0.5 N Dy log(2 pi);  out[1] = sum_i S_ii   (one block)
__global__ __launch_bounds__(256) void gpr_scalars_kernel(const double* __restrict__ A, const double* __restrict__ L,
                                                          const double* __restrict__ S, int N, int Np, int Dy,
                                                          double* __restrict__ out) {
  __shared__ double sh[8];
  double a = 0.0, b = 0.0, c = 0.0;
  for (int idx = threadIdx.x; idx < N * Dy; idx += blockDim.x) { const double v = A[idx]; a += v * v; }
  for (int i = threadIdx.x; i < N; i += blockDim.x) { b += log(L[(long)i * Np + i]); c += S[(long)i * Np + i]; }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); c += __shfl_down(c, o); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[w] = a; sh[4 + w] = b; }
  __syncthreads();
  double a2 = sh[0] + sh[1] + sh[2] + sh[3], b2 = sh[4] + sh[5] + sh[6] + sh[7];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = -0.5 * a2 - Dy * b2 - 0.5 * N * Dy * 1.8378770664093454836;
    out[1] = sh[0] + sh[1] + sh[2] + sh[3];
  }
}
hipError_t gpr_scalars(hipStream_t st, const double* A, const double* L, const double* S, int N, int Np, int Dy, double* out) {
  hipLaunchKernelGGL(gpr_scalars_kernel, dim3(1), dim3(256), 0, st, A, L, S, N, Np, Dy, out);
  LAUNCH_CHECK();
}
// S = 0.5 (S - Dy * Kinv)     (S holds alpha alpha^T on entry)
__global__ void gpr_dk_kernel(double* __restrict__ S, const double* __restrict__ Kinv, int Dy, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) S[idx] = 0.5 * (S[idx] - Dy * Kinv[idx]);
}
hipError_t gpr_dk(hipStream_t st, double* S, const double* Kinv, int Dy, int Np) {
  const long n = (long)Np * Np;
  hipLaunchKernelGGL(gpr_dk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, Kinv, Dy, n);
  LAUNCH_CHECK();
}
// var[i][d] = kvar - cn[i] (+ noise), cn = sum of the row-norm partial planes
__global__ void gpr_predict_var_kernel(const double* __restrict__ cnp, int nplane, long pstride, long N, int Dy, double kvar,
                                       double add, double* __restrict__ var) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * Dy) return;
  const long i = idx / Dy;
  double cn = 0.0;
  for (int q = 0; q < nplane; ++q) cn += cnp[(long)q * pstride + i];
  var[idx] = kvar - cn + add;
}
hipError_t gpr_predict_var(hipStream_t st, const double* cnp, int nplane, long pstride, long N, int Dy, double kvar, double add,
                           double* var) {
  hipLaunchKernelGGL(gpr_predict_var_kernel, dim3((unsigned)((N * Dy + 255) / 256)), dim3(256), 0, st, cnp, nplane, pstride, N, Dy,
                     kvar, add, var);
  LAUNCH_CHECK();
}

// dC[p][i] = sum_d mbar[p][d] A[i][d] - 2 (sum_d vbar[p][d]) C[p][i]: cotangent of c = L^-1 k(X, x_p) for the exact GP,
// where mean = c^T A and var = kdiag - |c|^2
__global__ void gpr_cbar_kernel(const double* __restrict__ mbar, const double* __restrict__ vbar, const double* __restrict__ A,
                                const double* __restrict__ Ct, long P, int Np, int Dy, double* __restrict__ Cbar) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * Np) return;
  const long p = idx / Np;
  const int i = (int)(idx % Np);
  double a = 0.0, sv = 0.0;
  for (int d = 0; d < Dy; ++d) { a += mbar[p * Dy + d] * A[(long)i * Dy + d]; sv += vbar[p * Dy + d]; }
  Cbar[idx] = a - 2.0 * sv * Ct[idx];
}
hipError_t gpr_cbar(hipStream_t st, const double* mbar, const double* vbar, const double* A, const double* Ct, long P, int Np,
                    int Dy, double* Cbar) {
  const long n = P * Np;
  hipLaunchKernelGGL(gpr_cbar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mbar, vbar, A, Ct, P, Np, Dy, Cbar);
  LAUNCH_CHECK();
}

}  // namespace dgp
