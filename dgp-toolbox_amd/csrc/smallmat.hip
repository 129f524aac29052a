// Small-matrix (M x M, batched) kernels of the SVGP layer: Kuu assembly, Cholesky, triangular
// inverse, KL, and the element-wise pieces of the hand-derived backward chain and of the
// natural-gradient step.  Reference: layers.py:227-234 (Kuu + jitter, Cholesky), layers.py:280-308
// (KL), gpflow NaturalGradient [ext] (dgp.py:312,343).  Latency-bound work: M <= 1024.
#include "dgp_internal.h"

namespace dgp {

#define LAUNCH_CHECK() return hipGetLastError()

// ---------------------------------------------------------------------------------------- Kuu
__global__ void rbf_kuu_kernel(const double* __restrict__ Z, const double* __restrict__ var,
                               const double* __restrict__ ls, int M, int Mp, int Din, double* __restrict__ Kuu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Mp * Mp) return;
  const int m = (int)(idx / Mp), n = (int)(idx % Mp);
  double v;
  if (m < M && n < M) {
    double r2 = 0.0;
    for (int j = 0; j < Din; ++j) {
      const double d = (Z[(long)m * Din + j] - Z[(long)n * Din + j]) / ls[j];
      r2 += d * d;
    }
    v = var[0] * exp(-0.5 * r2) + (m == n ? kJitter : 0.0);
  } else {
    v = (m == n) ? 1.0 : 0.0;
  }
  Kuu[idx] = v;
}

hipError_t rbf_kuu(hipStream_t st, const double* Z, const double* var, const double* ls, int M, int Mp, int Din,
                   double* Kuu) {
  const long n = (long)Mp * Mp;
  hipLaunchKernelGGL(rbf_kuu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Z, var, ls, M, Mp, Din, Kuu);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- Cholesky
// Right-looking blocked Cholesky, one 1024-thread workgroup per matrix; the diagonal block is
// factored in LDS, the panel solve keeps one row per thread in registers, the trailing update reads
// the panel rows from global memory (L2-resident: M*M*8 B <= 8 MiB).
constexpr int CH_NB = 32;

__global__ __launch_bounds__(1024) void chol_kernel(double* __restrict__ Aall, int Mp, int* __restrict__ info) {
  double* A = Aall + (long)blockIdx.x * Mp * Mp;
  __shared__ double Dg[CH_NB][CH_NB + 1];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j0 = 0; j0 < Mp; j0 += CH_NB) {
    const int nb = min(CH_NB, Mp - j0);
    for (int idx = tid; idx < nb * nb; idx += nt) {
      const int r = idx / nb, c = idx % nb;
      Dg[r][c] = A[(long)(j0 + r) * Mp + j0 + c];
    }
    __syncthreads();
    for (int c = 0; c < nb; ++c) {
      if (tid == 0) {
        const double d = Dg[c][c];
        if (!(d > 0.0)) { atomicOr(info, 1); Dg[c][c] = nan(""); }
        else Dg[c][c] = sqrt(d);
      }
      __syncthreads();
      if (tid > c && tid < nb) Dg[tid][c] /= Dg[c][c];
      __syncthreads();
      for (int idx = tid; idx < nb * nb; idx += nt) {
        const int i = idx / nb, k = idx % nb;
        if (k > c && i >= k) Dg[i][k] -= Dg[i][c] * Dg[k][c];
      }
      __syncthreads();
    }
    for (int idx = tid; idx < nb * nb; idx += nt) {
      const int r = idx / nb, c = idx % nb;
      A[(long)(j0 + r) * Mp + j0 + c] = (c <= r) ? Dg[r][c] : 0.0;
    }
    // panel: rows below the diagonal block,  X = A21 * L11^-T
    const int T = Mp - j0 - nb;
    for (int r = tid; r < T; r += nt) {
      double* row = A + (long)(j0 + nb + r) * Mp + j0;
      double x[CH_NB];
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) x[c] = (c < nb) ? row[c] : 0.0;
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) {
        if (c < nb) {
          double s = x[c];
#pragma unroll
          for (int t = 0; t < CH_NB; ++t)
            if (t < c) s -= x[t] * Dg[c][t];
          x[c] = s / Dg[c][c];
        }
      }
#pragma unroll
      for (int c = 0; c < CH_NB; ++c)
        if (c < nb) row[c] = x[c];
    }
    // zero the strictly-upper part right of the diagonal block (clean lower-triangular output)
    for (long idx = tid; idx < (long)nb * T; idx += nt) {
      const int r = (int)(idx / T), c = (int)(idx % T);
      A[(long)(j0 + r) * Mp + j0 + nb + c] = 0.0;
    }
    __syncthreads();
    // trailing update  A22 -= X X^T  (lower part)
    for (long idx = tid; idx < (long)T * T; idx += nt) {
      const int i = (int)(idx / T), k = (int)(idx % T);
      if (k <= i) {
        const double* ri = A + (long)(j0 + nb + i) * Mp + j0;
        const double* rk = A + (long)(j0 + nb + k) * Mp + j0;
        double s = 0.0;
        for (int c = 0; c < nb; ++c) s += ri[c] * rk[c];
        A[(long)(j0 + nb + i) * Mp + j0 + nb + k] -= s;
      }
    }
    __syncthreads();
  }
}

hipError_t chol_lower(hipStream_t st, double* A, int Mp, int batch, int* info) {
  hipLaunchKernelGGL(chol_kernel, dim3(batch), dim3(1024), 0, st, A, Mp, info);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- L^-1
// Column-parallel forward substitution: thread j owns column j of X = L^-1.
__global__ __launch_bounds__(1024) void trinv_kernel(const double* __restrict__ Lall, double* __restrict__ Xall,
                                                     int Mp) {
  const double* L = Lall + (long)blockIdx.x * Mp * Mp;
  double* X = Xall + (long)blockIdx.x * Mp * Mp;
  for (int j = threadIdx.x; j < Mp; j += blockDim.x) {
    for (int i = 0; i < j; ++i) X[(long)i * Mp + j] = 0.0;
    X[(long)j * Mp + j] = 1.0 / L[(long)j * Mp + j];
    for (int i = j + 1; i < Mp; ++i) {
      const double* Li = L + (long)i * Mp;
      double s = 0.0;
      for (int k = j; k < i; ++k) s += Li[k] * X[(long)k * Mp + j];
      X[(long)i * Mp + j] = -s / Li[i];
    }
  }
}

hipError_t trinv_lower(hipStream_t st, const double* L, double* X, int Mp, int batch) {
  hipLaunchKernelGGL(trinv_kernel, dim3(batch), dim3(Mp < 1024 ? ((Mp + 63) / 64) * 64 : 1024), 0, st, L, X, Mp);
  LAUNCH_CHECK();
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

__global__ __launch_bounds__(1024) void layer_kl_kernel(const double* __restrict__ Wcat, const double* __restrict__ u,
                                                        const double* __restrict__ Lq, const double* __restrict__ Lu,
                                                        int M, int Mp, int D, int white, double* __restrict__ out) {
  __shared__ double sh[16];
  double acc = 0.0;
  const long nW = (long)Mp * D * Mp;
  for (long idx = threadIdx.x; idx < nW; idx += blockDim.x) {
    const int k = (int)(idx / ((long)D * Mp)), n = (int)(idx % Mp);
    if (k < M && n < M) { const double w = Wcat[idx]; acc += 0.5 * w * w; }
  }
  for (long idx = threadIdx.x; idx < (long)M * D; idx += blockDim.x) { const double x = u[idx]; acc += 0.5 * x * x; }
  for (long idx = threadIdx.x; idx < (long)D * M; idx += blockDim.x) {
    const int d = (int)(idx / M), i = (int)(idx % M);
    const double q = Lq[((long)d * Mp + i) * Mp + i];
    acc -= 0.5 * log(q * q);
  }
  if (!white)
    for (int i = threadIdx.x; i < M; i += blockDim.x) acc += (double)D * log(Lu[(long)i * Mp + i]);
  const double t = block_sum_1024(acc, sh);
  if (threadIdx.x == 0) out[0] += t - 0.5 * (double)D * (double)M;
}

hipError_t layer_kl(hipStream_t st, const double* Wcat, const double* u, const double* Lq, const double* Lu, int M,
                    int Mp, int D, int white, double* kl_out) {
  hipLaunchKernelGGL(layer_kl_kernel, dim3(1), dim3(1024), 0, st, Wcat, u, Lq, Lu, M, Mp, D, white, kl_out);
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
__global__ void rbf_kuu_bwd_kernel(const double* __restrict__ S, const double* __restrict__ Kuu,
                                   const double* __restrict__ Z, const double* __restrict__ var,
                                   const double* __restrict__ ls, int M, int Mp, int Din, double* __restrict__ dZ,
                                   double* __restrict__ dls, double* __restrict__ dvar) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  double zacc = 0.0, lacc = 0.0, vacc = 0.0;
  if (m < M) {
    const double zm = Z[(long)m * Din + j], l = ls[j];
    for (int n = 0; n < M; ++n) {
      const double k0 = Kuu[(long)m * Mp + n] - (m == n ? kJitter : 0.0);
      const double h = 0.5 * (S[(long)m * Mp + n] + S[(long)n * Mp + m]) * k0;
      const double dl = zm - Z[(long)n * Din + j];
      zacc += h * dl;
      lacc += h * dl * dl;
      vacc += h;
    }
    dZ[(long)m * Din + j] += -2.0 * zacc / (l * l);
  }
  // wave-reduce the scalar accumulators, one atomic per wave
  for (int o = 32; o > 0; o >>= 1) { lacc += __shfl_down(lacc, o); vacc += __shfl_down(vacc, o); }
  if ((threadIdx.x & 63) == 0) {
    const double l = ls[j];
    unsafeAtomicAdd(dls + j, lacc / (l * l * l));
    if (j == 0) unsafeAtomicAdd(dvar, vacc / var[0]);
  }
}
hipError_t rbf_kuu_bwd(hipStream_t st, const double* S, const double* Kuu, const double* Z, const double* var,
                       const double* ls, int M, int Mp, int Din, double* dZ, double* dls, double* dvar) {
  hipLaunchKernelGGL(rbf_kuu_bwd_kernel, dim3((M + 255) / 256, Din), dim3(256), 0, st, S, Kuu, Z, var, ls, M, Mp, Din,
                     dZ, dls, dvar);
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

}  // namespace dgp
