// Small-matrix (M x M, batched) kernels of the SVGP layer: Kuu assembly, Cholesky, triangular
// inverse, KL, and the element-wise pieces of the hand-derived backward chain and of the
// natural-gradient step.  Reference: layers.py:227-234 (Kuu + jitter, Cholesky), layers.py:280-308
// (KL), gpflow NaturalGradient [ext] (dgp.py:312,343).  Latency-bound work: M <= 1024.
#include "dgp_internal.h"

namespace dgp {

#define LAUNCH_CHECK() return hipGetLastError()

// ---------------------------------------------------------------------------------------- Kuu
__global__ void rbf_kuu_kernel(int kind, const double* __restrict__ Z, const double* __restrict__ var,
                               const double* __restrict__ ls, int M, int Mp, int Din, double* __restrict__ Kuu,
                               double* __restrict__ Euu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Mp * Mp) return;
  const int m = (int)(idx / Mp), n = (int)(idx % Mp);
  double v, e = 0.0;
  if (m < M && n < M) {
    double r2 = 0.0;
    for (int j = 0; j < Din; ++j) {
      const double d = (Z[(long)m * Din + j] - Z[(long)n * Din + j]) / ls[j];
      r2 += d * d;
    }
    stationary_k(kind, var[0], r2, v, e);
    v += (m == n ? kJitter : 0.0);
  } else {
    v = (m == n) ? 1.0 : 0.0;
  }
  Kuu[idx] = v;
  if (Euu) Euu[idx] = e;
}

hipError_t rbf_kuu(hipStream_t st, int kind, const double* Z, const double* var, const double* ls, int M, int Mp, int Din,
                   double* Kuu, double* Euu) {
  const long n = (long)Mp * Mp;
  hipLaunchKernelGGL(rbf_kuu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kind, Z, var, ls, M, Mp, Din, Kuu,
                     Euu);
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
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// LF = 64: the leaf of the recursion.  LF = 32: models with at most 32 inducing points (the Bayesian-optimisation
// surrogates; Mp = 64 with an identity padding): a quarter of the dependent readlane / fma chain (57 -> 16 us), the padding
// rows [n, npad) of X are written as identity rows.
template <int LF>
__global__ __launch_bounds__(64) void leaf_potrf_inv_kernel(double* __restrict__ Aall, double* __restrict__ Xall, int ld,
                                                            long stride, int off, int n, int do_chol,
                                                            int* __restrict__ info, int npad) {
  __shared__ double T[LF][LF + 1];
  double* A = Aall + (long)blockIdx.x * stride + (long)off * ld + off;
  double* X = Xall + (long)blockIdx.x * stride + (long)off * ld + off;
  const int i = threadIdx.x;
  for (int r = 0; r < n; ++r)
    if (i < n) T[r][i] = A[(long)r * ld + i];
  __syncthreads();
  double a[LF];
  // rows and columns beyond n behave as an identity block
#pragma unroll
  for (int j = 0; j < LF; ++j) {
    const double v = T[i < n ? i : 0][j];
    a[j] = (i < n && j < n) ? (j <= i ? v : 0.0) : (i == j ? 1.0 : 0.0);
  }
  double diag = 1.0;
  if (do_chol) {
    bool bad = false;
#pragma unroll
    for (int c = 0; c < LF; ++c) {
      const double piv = lane_bcast(a[c], c);
      // 1/sqrt(pivot): hardware estimate + two Newton steps (a library sqrt and a division are ~40 dependent
      // fp64 operations on the critical path of every column)
      double rs = __builtin_amdgcn_rsq(piv);
      rs = rs * (1.5 - 0.5 * piv * rs * rs);
      rs = rs * (1.5 - 0.5 * piv * rs * rs);
      if (!(piv > 0.0)) { rs = nan(""); bad = true; }
      const double l = (i >= c) ? a[c] * rs : 0.0;       // column c of L (zero above the diagonal)
      a[c] = l;
      if (i == c) diag = l;
#pragma unroll
      for (int j = c + 1; j < LF; ++j) a[j] = fma(-l, lane_bcast(l, j), a[j]);   // only j <= i is meaningful
    }
    if (bad && i == 0) atomicOr(info, 1);
    __syncthreads();
    if (i < LF)
#pragma unroll
      for (int j = 0; j < LF; ++j) T[i][j] = a[j];
    __syncthreads();
    for (int r = 0; r < n; ++r)
      if (i < n) A[(long)r * ld + i] = T[r][i];
  } else {
#pragma unroll
    for (int j = 0; j < LF; ++j) diag = (i == j) ? a[j] : diag;
  }
  const double dinv = 1.0 / diag;
#pragma unroll
  for (int k = 0; k < LF; ++k) a[k] = (i > k) ? a[k] * dinv : 0.0;      // strictly lower part of L' = D^-1 L
#pragma unroll
  for (int k = 0; k < LF; ++k) {
    const double m = a[k];                                  // L'[i][k] for the rows below k, 0 for the others
#pragma unroll
    for (int j = 0; j < k; ++j) a[j] = fma(-m, lane_bcast(a[j], k), a[j]);   // row k of X' is final in lane k
    a[k] = -m;
  }
#pragma unroll
  for (int j = 0; j < LF; ++j) {
    const double dj = lane_bcast(dinv, j);
    a[j] = (i > j) ? a[j] * dj : (i == j ? dinv : 0.0);
  }
  __syncthreads();
  if (i < LF)
#pragma unroll
    for (int j = 0; j < LF; ++j) T[i][j] = a[j];
  __syncthreads();
  for (int r = 0; r < n; ++r)
    if (i < n) X[(long)r * ld + i] = T[r][i];
  // identity padding of X (and zeros beside it) when the factorisation covers only the leading n of npad rows
  for (int r = 0; r < npad; ++r)
    if (i < npad && (r >= n || i >= n)) X[(long)r * ld + i] = (r == i) ? 1.0 : 0.0;
}

// batched strided block copy: dst[b][r][c] = src[b][r][c] for an nr x nc block (leading dimension ld, batch stride)
__global__ void copy_block_kernel(const double* __restrict__ src, double* __restrict__ dst, int ld, long stride, int nr,
                                  int nc) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)nr * nc) return;
  const long o = (long)blockIdx.y * stride + (idx / nc) * ld + idx % nc;
  dst[o] = src[o];
}

__global__ void zero_block_kernel(double* __restrict__ Aall, int ld, long stride, int r0, int c0, int nr, int nc) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)nr * nc) return;
  double* A = Aall + (long)blockIdx.y * stride;
  A[(long)(r0 + idx / nc) * ld + c0 + idx % nc] = 0.0;
}

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
    hipLaunchKernelGGL(leaf_potrf_inv_kernel<LEAF>, dim3(batch), dim3(64), 0, st, A, X, ld, stride, off, n, do_chol, info, 0);
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
      hipLaunchKernelGGL(copy_block_kernel, dim3((unsigned)((nc + 255) / 256), batch), dim3(256), 0, st, T, A21, ld, stride, n2, n1);
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
    hipLaunchKernelGGL(zero_block_kernel, dim3((unsigned)((nz + 255) / 256), batch), dim3(256), 0, st, A, ld, stride, off, off + n1, n1, n2);
  hipLaunchKernelGGL(zero_block_kernel, dim3((unsigned)((nz + 255) / 256), batch), dim3(256), 0, st, X, ld, stride, off, off + n1, n1, n2);
  return hipGetLastError();
}

hipError_t potrf_inv(hipStream_t st, double* A, double* X, double* tmp, int Mp, int batch, int* info, int n_act) {
  if (Mp <= LEAF && n_act > 0 && n_act <= 32) {      // A = blockdiag(A[:n, :n], I): only the leading block is factorised
    hipLaunchKernelGGL(leaf_potrf_inv_kernel<32>, dim3(batch), dim3(64), 0, st, A, X, Mp, (long)Mp * Mp, 0, n_act, 1, info, Mp);
    return hipGetLastError();
  }
  return potrf_inv_rec(st, A, X, tmp, Mp, (long)Mp * Mp, batch, 0, Mp, 1, info);
}

hipError_t trinv_lower(hipStream_t st, const double* L, double* X, double* tmp, int Mp, int batch) {
  return potrf_inv_rec(st, const_cast<double*>(L), X, tmp, Mp, (long)Mp * Mp, batch, 0, Mp, 0, nullptr);
}

// ---------------------------------------------------------------------------------------- packing
__global__ void pack_q_kernel(const double* __restrict__ q_sqrt, const double* __restrict__ q_mu, int M, int Mp, int D,
                              double* __restrict__ Lq, double* __restrict__ qmu_p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nL = (long)D * Mp * Mp;
  if (idx < nL) {
    const int d = (int)(idx / ((long)Mp * Mp));
    const int i = (int)((idx / Mp) % Mp), j = (int)(idx % Mp);
    double v;
    if (i < M && j < M) v = (j <= i) ? q_sqrt[((long)d * M + i) * M + j] : 0.0;
    else v = (i == j) ? 1.0 : 0.0;
    Lq[idx] = v;
  }
  if (idx < (long)Mp * D) {
    const int i = (int)(idx / D), d = (int)(idx % D);
    qmu_p[idx] = (i < M) ? q_mu[(long)i * D + d] : 0.0;
  }
}

hipError_t pack_q(hipStream_t st, const double* q_sqrt, const double* q_mu, int M, int Mp, int D, double* Lq,
                  double* qmu_p) {
  const long n = (long)D * Mp * Mp;
  hipLaunchKernelGGL(pack_q_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q_sqrt, q_mu, M, Mp, D, Lq,
                     qmu_p);
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

__global__ void lq_to_wcat_kernel(const double* __restrict__ Lq, int Mp, int D, double* __restrict__ Wcat) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)D * Mp * Mp) return;
  const int d = (int)(idx / ((long)Mp * Mp));
  const int k = (int)((idx / Mp) % Mp), n = (int)(idx % Mp);
  Wcat[(long)k * D * Mp + (long)d * Mp + n] = Lq[idx];
}

hipError_t lq_to_wcat(hipStream_t st, const double* Lq, int Mp, int D, double* Wcat) {
  const long n = (long)D * Mp * Mp;
  hipLaunchKernelGGL(lq_to_wcat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Lq, Mp, D, Wcat);
  LAUNCH_CHECK();
}

__global__ void copy_kernel(const double* __restrict__ s, double* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = s[i];
}
hipError_t copy_mat(hipStream_t st, const double* src, double* dst, long n) {
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
  LAUNCH_CHECK();
}

__global__ void make_z1_kernel(const double* __restrict__ Z, int M, int Mp, int Din, double* __restrict__ Z1) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = Din + 1;
  if (idx >= (long)Mp * w) return;
  const int m = (int)(idx / w), j = (int)(idx % w);
  Z1[idx] = (m < M) ? (j < Din ? Z[(long)m * Din + j] : 1.0) : 0.0;
}
hipError_t make_z1(hipStream_t st, const double* Z, int M, int Mp, int Din, double* Z1) {
  const long n = (long)Mp * (Din + 1);
  hipLaunchKernelGGL(make_z1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Z, M, Mp, Din, Z1);
  LAUNCH_CHECK();
}

// writes (not accumulates) the Kuf + K_diag parts of d ELBO / d (Z, lengthscales, variance); one block per j
__global__ __launch_bounds__(256) void rbf_kuf_bwd_finish_kernel(const double* __restrict__ GX, const double* __restrict__ x2rs,
                                                                 const double* __restrict__ vsum, const double* __restrict__ Z,
                                                                 const double* __restrict__ var, const double* __restrict__ ls,
                                                                 int M, int Din, double* __restrict__ dZ,
                                                                 double* __restrict__ dls, double* __restrict__ dvar,
                                                                 const double* __restrict__ kdot) {
  __shared__ double sh[16];
  const int j = blockIdx.x, w = Din + 1;
  const double l = ls[j];
  double la = 0.0, va = 0.0;
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    const double cs = GX[(long)m * w + Din], gx = GX[(long)m * w + j], z = Z[(long)m * Din + j];
    dZ[(long)m * Din + j] = -(z * cs - gx) / (l * l);
    la += z * z * cs - 2.0 * z * gx;
    va += cs;
  }
  for (int o = 32; o > 0; o >>= 1) { la += __shfl_down(la, o); va += __shfl_down(va, o); }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { sh[wv] = la; sh[8 + wv] = va; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double lt = 0.0, vt = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { lt += sh[i]; vt += sh[8 + i]; }
    dls[j] = (lt + x2rs[j]) / (l * l * l);
    if (j == 0) dvar[0] = (kdot ? kdot[0] : vt) / var[0] + vsum[0];
  }
}
hipError_t rbf_kuf_bwd_finish(hipStream_t st, const double* GX, const double* x2rs, const double* vsum, const double* Z,
                              const double* var, const double* ls, int M, int Din, double* dZ, double* dls, double* dvar,
                              const double* kdot) {
  hipLaunchKernelGGL(rbf_kuf_bwd_finish_kernel, dim3(Din), dim3(256), 0, st, GX, x2rs, vsum, Z, var, ls, M, Din, dZ, dls, dvar,
                     kdot);
  LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void lower_dot_kernel(const double* __restrict__ Lu, const double* __restrict__ Q, int M,
                                                        int Mp, double* __restrict__ out) {
  __shared__ double sh[4];
  double a = 0.0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)M * M; idx += (long)gridDim.x * blockDim.x) {
    const int i = (int)(idx / M), j = (int)(idx % M);
    if (j <= i) a += Lu[(long)i * Mp + j] * Q[(long)i * Mp + j];
  }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}
hipError_t lower_dot(hipStream_t st, const double* Lu, const double* Q, int M, int Mp, double* out) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lower_dot_kernel, dim3(16), dim3(256), 0, st, Lu, Q, M, Mp, out);
  LAUNCH_CHECK();
}

// WT[(d*Mp + j)*Mp + i] = Wcat[i*(D*Mp) + d*Mp + j]: the W_d^T stacked vertically ([D*Mp x Mp]), the B operand of
// dC = [2 vbar .* T] * WTcat when t_d = W_d^T c is kept from the forward pass
__global__ void wcat_transpose_kernel(const double* __restrict__ Wcat, int Mp, int D, double* __restrict__ WT) {
  __shared__ double tile[32][33];
  const int d = blockIdx.z;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  for (int r = ty; r < 32; r += 8) tile[r][tx] = Wcat[(long)(i0 + r) * D * Mp + (long)d * Mp + j0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) WT[((long)d * Mp + j0 + r) * Mp + i0 + tx] = tile[tx][r];
}
hipError_t wcat_transpose(hipStream_t st, const double* Wcat, int Mp, int D, double* WT) {
  hipLaunchKernelGGL(wcat_transpose_kernel, dim3(Mp / 32, Mp / 32, D), dim3(256), 0, st, Wcat, Mp, D, WT);
  LAUNCH_CHECK();
}

__global__ void sub_identity_kernel(double* __restrict__ S, int M, int Mp, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int i = (int)((idx / Mp) % Mp), j = (int)(idx % Mp);
  if (i >= M || j >= M) S[idx] = 0.0;            // padding: W_pad = I so W W^T - I = 0 there
  else if (i == j) S[idx] -= 1.0;
}
hipError_t sub_identity(hipStream_t st, double* S, int M, int Mp, int batch) {
  const long n = (long)batch * Mp * Mp;
  hipLaunchKernelGGL(sub_identity_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, M, Mp, n);
  LAUNCH_CHECK();
}

// G (lower triangle valid) -> symmetric: tile (bi, bj) with bj > bi takes the transpose of tile (bj, bi) through LDS
// (coalesced reads and writes; the element-wise version read G column-wise right after the atomics of the Gram product
// had left it in HBM: 167 us for 8 x 256^2, on the critical path of the last layer's chain), diagonal tiles in place.
__global__ __launch_bounds__(256) void symmetrize_lower_kernel(double* __restrict__ G, int Mp) {
  __shared__ double tile[32][33];
  const int nb = Mp / 32;
  // enumerate the pairs bi <= bj
  int t = blockIdx.x, bi = 0;
  while (t >= nb - bi) { t -= nb - bi; ++bi; }
  const int bj = bi + t;
  double* __restrict__ Gb = G + (long)blockIdx.y * Mp * Mp;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  for (int r = ty; r < 32; r += 8) tile[r][tx] = Gb[(long)(bj * 32 + r) * Mp + bi * 32 + tx];     // lower tile (bj, bi)
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    if (j > i) Gb[(long)i * Mp + j] = tile[tx][r];
  }
}
hipError_t symmetrize_lower(hipStream_t st, double* G, int Mp, int batch) {
  const int nb = Mp / 32;
  hipLaunchKernelGGL(symmetrize_lower_kernel, dim3((unsigned)(nb * (nb + 1) / 2), (unsigned)batch), dim3(256), 0, st, G, Mp);
  LAUNCH_CHECK();
}

__global__ void sub_scalars_kernel(const double* a, const double* b, double* out) { out[0] = a[0] - b[0]; }
hipError_t sub_scalars(hipStream_t st, const double* a, const double* b, double* out) {
  hipLaunchKernelGGL(sub_scalars_kernel, dim3(1), dim3(1), 0, st, a, b, out);
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

__global__ __launch_bounds__(256) void layer_kl_kernel(const double* __restrict__ Wcat, const double* __restrict__ u,
                                                       const double* __restrict__ Lq, const double* __restrict__ Lu,
                                                       int M, int Mp, int D, int white, double* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  const long nW = (long)Mp * D * Mp, gstride = (long)gridDim.x * blockDim.x, g0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long idx = g0; idx < nW; idx += gstride) {
    const int k = (int)(idx / ((long)D * Mp)), n = (int)(idx % Mp);
    if (k < M && n < M) { const double w = Wcat[idx]; acc += 0.5 * w * w; }
  }
  for (long idx = g0; idx < (long)M * D; idx += gstride) { const double x = u[idx]; acc += 0.5 * x * x; }
  for (long idx = g0; idx < (long)D * M; idx += gstride) {
    const int d = (int)(idx / M), i = (int)(idx % M);
    const double q = Lq[((long)d * Mp + i) * Mp + i];
    acc -= 0.5 * log(q * q);
  }
  if (!white)
    for (long i = g0; i < M; i += gstride) acc += (double)D * log(Lu[i * Mp + i]);
  if (g0 == 0) acc -= 0.5 * (double)D * (double)M;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

hipError_t layer_kl(hipStream_t st, const double* Wcat, const double* u, const double* Lq, const double* Lu, int M,
                    int Mp, int D, int white, double* kl_out) {
  hipLaunchKernelGGL(layer_kl_kernel, dim3(64), dim3(256), 0, st, Wcat, u, Lq, Lu, M, Mp, D, white, kl_out);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- backward chain pieces
__global__ void wbar_total_kernel(double* __restrict__ dW, const double* __restrict__ W, double* __restrict__ du,
                                  const double* __restrict__ u, int M, int Mp, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nW = (long)Mp * D * Mp;
  if (idx < nW) {
    const int k = (int)(idx / ((long)D * Mp)), n = (int)(idx % Mp);
    dW[idx] = (k < M && n < M && n <= k) ? (dW[idx] - W[idx]) : 0.0;
  }
  if (idx < (long)Mp * D) {
    const int i = (int)(idx / D);
    du[idx] = (i < M) ? (du[idx] - u[idx]) : 0.0;
  }
}
hipError_t wbar_total(hipStream_t st, double* dWcat, const double* Wcat, double* du, const double* u, int M, int Mp,
                      int D) {
  const long n = (long)Mp * D * Mp;
  hipLaunchKernelGGL(wbar_total_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dWcat, Wcat, du, u, M, Mp,
                     D);
  LAUNCH_CHECK();
}

__global__ void lqbar_finish_kernel(double* __restrict__ dLq, const double* __restrict__ Lq, int M, int Mp, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)D * Mp * Mp) return;
  const int i = (int)((idx / Mp) % Mp), j = (int)(idx % Mp);
  double v = 0.0;
  if (i < M && j <= i) v = dLq[idx] + (i == j ? 1.0 / Lq[idx] : 0.0);
  dLq[idx] = v;
}
hipError_t lqbar_finish(hipStream_t st, double* dLq, const double* Lq, int M, int Mp, int D) {
  const long n = (long)D * Mp * Mp;
  hipLaunchKernelGGL(lqbar_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dLq, Lq, M, Mp, D);
  LAUNCH_CHECK();
}

__global__ void lubar_finish_kernel(double* __restrict__ dLu, const double* __restrict__ T2, const double* __restrict__ Lu,
                                    int M, int Mp, int D, int white) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Mp * Mp) return;
  const int i = (int)(idx / Mp), j = (int)(idx % Mp);
  double v = 0.0;
  if (i < M && j <= i) {
    v = -(dLu[idx] + (T2 ? T2[idx] : 0.0));
    if (i == j && !white) v -= (double)D / Lu[idx];
  }
  dLu[idx] = v;
}
hipError_t lubar_finish(hipStream_t st, double* dLu, const double* T2, const double* Lu, int M, int Mp, int D,
                        int white) {
  const long n = (long)Mp * Mp;
  hipLaunchKernelGGL(lubar_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dLu, T2, Lu, M, Mp, D,
                     white);
  LAUNCH_CHECK();
}

__global__ void phi_kernel(double* __restrict__ T, int Mp, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int i = (int)((idx / Mp) % Mp), j = (int)(idx % Mp);
  const double v = T[idx];
  T[idx] = (j < i) ? v : (i == j ? 0.5 * v : 0.0);
}
hipError_t phi_tril_halfdiag(hipStream_t st, double* T, int Mp, int batch) {
  const long n = (long)batch * Mp * Mp;
  hipLaunchKernelGGL(phi_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, T, Mp, n);
  LAUNCH_CHECK();
}

// RBF backward through Kuu = K(Z,Z) + jitter*I with dKuu = sym(S)   
__global__ __launch_bounds__(64) void rbf_kuu_bwd_kernel(const double* __restrict__ S, const double* __restrict__ Kuu,
                                                         const double* __restrict__ Euu,
                                                         const double* __restrict__ Z, const double* __restrict__ var,
                                                         const double* __restrict__ ls, int M, int Mp, int Din,
                                                         double* __restrict__ dZ, double* __restrict__ dls,
                                                         double* __restrict__ dvar, const double* __restrict__ white) {
  // one wave per (inducing point m, input dimension j): lanes stride over n, so rows of Kuu and S are read coalesced
  const int m = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
  const double zm = Z[(long)m * Din + j], l = ls[j];
  double zacc = 0.0, lacc = 0.0, vacc = 0.0;
  for (int n = lane; n < M; n += 64) {
    const double k0 = Kuu[(long)m * Mp + n] - (m == n ? kJitter + (white ? white[0] : 0.0) : 0.0);
    const double sym = 0.5 * (S[(long)m * Mp + n] + S[(long)n * Mp + m]);
    const double h = sym * (Euu ? Euu[(long)m * Mp + n] : k0);       // dKuu .* e: inputs and lengthscales
    const double dl = zm - Z[(long)n * Din + j];
    zacc += h * dl;
    lacc += h * dl * dl;
    vacc += sym * k0;                                                 // dKuu .* k: variance
  }
  for (int o = 32; o > 0; o >>= 1) {
    zacc += __shfl_down(zacc, o);
    lacc += __shfl_down(lacc, o);
    vacc += __shfl_down(vacc, o);
  }
  if (lane == 0) {
    dZ[(long)m * Din + j] += -2.0 * zacc / (l * l);
    unsafeAtomicAdd(dls + j, lacc / (l * l * l));
    if (j == 0) unsafeAtomicAdd(dvar, vacc / var[0]);
  }
}
hipError_t rbf_kuu_bwd(hipStream_t st, const double* S, const double* Kuu, const double* Euu, const double* Z,
                       const double* var, const double* ls, int M, int Mp, int Din, double* dZ, double* dls, double* dvar,
                       const double* white) {
  hipLaunchKernelGGL(rbf_kuu_bwd_kernel, dim3(M, Din), dim3(64), 0, st, S, Kuu, Euu, Z, var, ls, M, Mp, Din, dZ, dls, dvar,
                     white);
  LAUNCH_CHECK();
}

__global__ void unpack_q_grads_kernel(const double* __restrict__ dLq, const double* __restrict__ dqmu_p, int M, int Mp,
                                      int D, double* __restrict__ g_q_sqrt, double* __restrict__ g_q_mu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long)D * M * M) {
    const int d = (int)(idx / ((long)M * M));
    const int i = (int)((idx / M) % M), j = (int)(idx % M);
    g_q_sqrt[idx] = (j <= i) ? dLq[((long)d * Mp + i) * Mp + j] : 0.0;
  }
  if (idx < (long)M * D) g_q_mu[idx] = dqmu_p[idx];
}
hipError_t unpack_q_grads(hipStream_t st, const double* dLq, const double* dqmu_p, int M, int Mp, int D,
                          double* g_q_sqrt, double* g_q_mu) {
  const long n = (long)D * M * M;
  hipLaunchKernelGGL(unpack_q_grads_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dLq, dqmu_p, M, Mp, D,
                     g_q_sqrt, g_q_mu);
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
