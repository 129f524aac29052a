// Bodies of the small-matrix chain kernels (see chain.h): each is the stand-alone kernel's code, with blockIdx / gridDim /
// threadIdx / blockDim as PARAMETERS (they shadow the built-ins), so that the same code runs as a kernel of its own
// (chain_global) and as a step of a recorded chain (chain_exec_kernel, one workgroup walking virtual blocks).
// Reference arithmetic: layers.py:227-234 (Kuu + jitter, Cholesky), layers.py:280-308 (KL), the backward chain of
// SURVEY App. B / C.  (Moved here from smallmat.hip / gemm_small.hip; the launchers stay there.)
#pragma once
#include "chain.h"
#include "dgp_internal.h"

namespace dgp {

template <typename... T> struct ArgList {};

// LDS of a body: a kernel of its own declares what it needs; the steps of a chain share ONE pool (the largest need: the two
// operand images of the small product), otherwise every inlined body would get an allocation of its own
constexpr int kChainPoolBytes = 2 * 64 * 66 * 8;
__shared__ __attribute__((aligned(16))) char g_chain_pool[kChainPoolBytes];
__shared__ char* g_body_ext;          // CM_WAVE_EXT: the calling kernel's buffer (it sets the pointer before the call)
template <int BYTES, int MODE>
__device__ __forceinline__ char* body_smem() {
  static_assert(BYTES <= kChainPoolBytes, "chain LDS pool too small");
  if constexpr (MODE == CM_WAVE_EXT) {
    return g_body_ext;
  } else if constexpr (MODE == CM_KERNEL) {
    __shared__ __attribute__((aligned(16))) char buf[BYTES];
    return buf;
  } else {
    return g_chain_pool;
  }
}

__device__ __forceinline__ double lane_bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

template <> struct ChainBody<CK_RBF_KUU> {      // rbf_kuu_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<int, const double*, const double*, const double*, int, int, int, double*, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, int kind, const double* Z, const double* var, const double* ls, int M, int Mp, int Din, double* Kuu, double* Euu) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
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
};

template <> struct ChainBody<CK_COPY_BLOCK> {      // copy_block_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, double*, int, long, int, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* src, double* dst, int ld, long stride, int nr, int nc) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)nr * nc) return;
  const long o = (long)blockIdx.y * stride + (idx / nc) * ld + idx % nc;
  dst[o] = src[o];
}
};

template <> struct ChainBody<CK_ZERO_BLOCK> {      // zero_block_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<double*, int, long, int, int, int, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* Aall, int ld, long stride, int r0, int c0, int nr, int nc) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)nr * nc) return;
  double* A = Aall + (long)blockIdx.y * stride;
  A[(long)(r0 + idx / nc) * ld + c0 + idx % nc] = 0.0;
}
};

template <> struct ChainBody<CK_PACK_Q> {      // pack_q_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, const double*, int, int, int, double*, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* q_sqrt, const double* q_mu, int M, int Mp, int D, double* Lq, double* qmu_p) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
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
};

template <> struct ChainBody<CK_LQ_TO_WCAT> {      // lq_to_wcat_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, int, int, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* Lq, int Mp, int D, double* Wcat) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)D * Mp * Mp) return;
  const int d = (int)(idx / ((long)Mp * Mp));
  const int k = (int)((idx / Mp) % Mp), n = (int)(idx % Mp);
  Wcat[(long)k * D * Mp + (long)d * Mp + n] = Lq[idx];
}
};

template <> struct ChainBody<CK_COPY> {      // copy_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, double*, long>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* s, double* d, long n) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = s[i];
}
};

template <> struct ChainBody<CK_MAKE_Z1> {      // make_z1_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, int, int, int, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* Z, int M, int Mp, int Din, double* Z1) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = Din + 1;
  if (idx >= (long)Mp * w) return;
  const int m = (int)(idx / w), j = (int)(idx % w);
  Z1[idx] = (m < M) ? (j < Din ? Z[(long)m * Din + j] : 1.0) : 0.0;
}
};

template <> struct ChainBody<CK_KUF_BWD_FINISH> {      // rbf_kuf_bwd_finish_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, const double*, const double*, const double*, const double*, const double*, int, int, double*, double*, double*, const double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* GX, const double* x2rs, const double* vsum, const double* Z, const double* var, const double* ls, int M, int Din, double* dZ, double* dls, double* dvar, const double* kdot) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  double (&sh)[16] = *reinterpret_cast<double (*)[16]>(body_smem<(16) * 8, MODE>());
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
  body_sync<MODE>();
  if (threadIdx.x == 0) {
    double lt = 0.0, vt = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { lt += sh[i]; vt += sh[8 + i]; }
    dls[j] = (lt + x2rs[j]) / (l * l * l);
    if (j == 0) dvar[0] = (kdot ? kdot[0] : vt) / var[0] + vsum[0];
  }
}
};

template <> struct ChainBody<CK_LOWER_DOT> {      // lower_dot_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, const double*, int, int, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* Lu, const double* Q, int M, int Mp, double* out) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  double (&sh)[4] = *reinterpret_cast<double (*)[4]>(body_smem<(4) * 8, MODE>());
  double a = 0.0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)M * M; idx += (long)gridDim.x * blockDim.x) {
    const int i = (int)(idx / M), j = (int)(idx % M);
    if (j <= i) a += Lu[(long)i * Mp + j] * Q[(long)i * Mp + j];
  }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  body_sync<MODE>();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}
};

template <> struct ChainBody<CK_WCAT_T> {      // wcat_transpose_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, int, int, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* Wcat, int Mp, int D, double* WT) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  double (&tile)[32][33] = *reinterpret_cast<double (*)[32][33]>(body_smem<(32) * (33) * 8, MODE>());
  const int d = blockIdx.z;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  for (int r = ty; r < 32; r += 8) tile[r][tx] = Wcat[(long)(i0 + r) * D * Mp + (long)d * Mp + j0 + tx];
  body_sync<MODE>();
  for (int r = ty; r < 32; r += 8) WT[((long)d * Mp + j0 + r) * Mp + i0 + tx] = tile[tx][r];
}
};

template <> struct ChainBody<CK_SUB_IDENTITY> {      // sub_identity_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<double*, int, int, long>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* S, int M, int Mp, long total) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int i = (int)((idx / Mp) % Mp), j = (int)(idx % Mp);
  if (i >= M || j >= M) S[idx] = 0.0;            // padding: W_pad = I so W W^T - I = 0 there
  else if (i == j) S[idx] -= 1.0;
}
};

template <> struct ChainBody<CK_SYMMETRIZE> {      // symmetrize_lower_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<double*, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* G, int Mp) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  double (&tile)[32][33] = *reinterpret_cast<double (*)[32][33]>(body_smem<(32) * (33) * 8, MODE>());
  const int nb = Mp / 32;
  // enumerate the pairs bi <= bj
  int t = blockIdx.x, bi = 0;
  while (t >= nb - bi) { t -= nb - bi; ++bi; }
  const int bj = bi + t;
  double* Gb = G + (long)blockIdx.y * Mp * Mp;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  for (int r = ty; r < 32; r += 8) tile[r][tx] = Gb[(long)(bj * 32 + r) * Mp + bi * 32 + tx];     // lower tile (bj, bi)
  body_sync<MODE>();
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    if (j > i) Gb[(long)i * Mp + j] = tile[tx][r];
  }
}
};

template <> struct ChainBody<CK_SUB_SCALARS> {      // sub_scalars_kernel
  static constexpr int kBlock = 1;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, const double*, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* a, const double* b, double* out) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim; out[0] = a[0] - b[0]; }
};

template <> struct ChainBody<CK_LAYER_KL> {      // layer_kl_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, const double*, const double*, const double*, int, int, int, int, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* Wcat, const double* u, const double* Lq, const double* Lu, int M, int Mp, int D, int white, double* out) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  double (&sh)[4] = *reinterpret_cast<double (*)[4]>(body_smem<(4) * 8, MODE>());
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
  body_sync<MODE>();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}
};

template <> struct ChainBody<CK_WBAR_TOTAL> {      // wbar_total_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<double*, const double*, double*, const double*, int, int, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* dW, const double* W, double* du, const double* u, int M, int Mp, int D) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
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
};

template <> struct ChainBody<CK_LQBAR> {      // lqbar_finish_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<double*, const double*, int, int, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* dLq, const double* Lq, int M, int Mp, int D) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)D * Mp * Mp) return;
  const int i = (int)((idx / Mp) % Mp), j = (int)(idx % Mp);
  double v = 0.0;
  if (i < M && j <= i) v = dLq[idx] + (i == j ? 1.0 / Lq[idx] : 0.0);
  dLq[idx] = v;
}
};

template <> struct ChainBody<CK_LUBAR> {      // lubar_finish_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<double*, const double*, const double*, int, int, int, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* dLu, const double* T2, const double* Lu, int M, int Mp, int D, int white) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
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
};

template <> struct ChainBody<CK_PHI> {      // phi_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<double*, int, long>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* T, int Mp, long total) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int i = (int)((idx / Mp) % Mp), j = (int)(idx % Mp);
  const double v = T[idx];
  T[idx] = (j < i) ? v : (i == j ? 0.5 * v : 0.0);
}
};

template <> struct ChainBody<CK_KUU_BWD> {      // rbf_kuu_bwd_kernel
  static constexpr int kBlock = 64;
  static constexpr bool kWaveParallel = true;
  using Args = ArgList<const double*, const double*, const double*, const double*, const double*, const double*, int, int, int, double*, double*, double*, const double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* S, const double* Kuu, const double* Euu, const double* Z, const double* var, const double* ls, int M, int Mp, int Din, double* dZ, double* dls, double* dvar, const double* white) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
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
};

template <> struct ChainBody<CK_UNPACK_Q> {      // unpack_q_grads_kernel
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, const double*, int, int, int, double*, double*>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* dLq, const double* dqmu_p, int M, int Mp, int D, double* g_q_sqrt, double* g_q_mu) {
    (void)blockIdx; (void)gridDim; (void)threadIdx; (void)blockDim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long)D * M * M) {
    const int d = (int)(idx / ((long)M * M));
    const int i = (int)((idx / M) % M), j = (int)(idx % M);
    g_q_sqrt[idx] = (j <= i) ? dLq[((long)d * Mp + i) * Mp + j] : 0.0;
  }
  if (idx < (long)M * D) g_q_mu[idx] = dqmu_p[idx];
}
};

template <> struct ChainBody<CK_LEAF32> {      // leaf_potrf_inv_kernel<32>: one wave per matrix, the matrix in registers
  static constexpr int kBlock = 64;
  static constexpr bool kWaveParallel = false;     // (its LDS transpose buffer is one per workgroup: a chain runs the batch on wave 0)
  static constexpr int LF = 32;
  using Args = ArgList<double*, double*, int, long, int, int, int, int*, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* Aall, double* Xall, int ld, long stride, int off, int n, int do_chol, int* info, int npad) {
    (void)gridDim; (void)blockDim;
  double (&T)[LF][LF + 1] = *reinterpret_cast<double (*)[LF][LF + 1]>(body_smem<(LF) * (LF + 1) * 8, MODE>());
  double* A = Aall + (long)blockIdx.x * stride + (long)off * ld + off;
  double* X = Xall + (long)blockIdx.x * stride + (long)off * ld + off;
  const int i = threadIdx.x;
  for (int r = 0; r < n; ++r)
    if (i < n) T[r][i] = A[(long)r * ld + i];
  body_sync<MODE>();
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
    body_sync<MODE>();
    if (i < LF)
#pragma unroll
      for (int j = 0; j < LF; ++j) T[i][j] = a[j];
    body_sync<MODE>();
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
  body_sync<MODE>();
  if (i < LF)
#pragma unroll
    for (int j = 0; j < LF; ++j) T[i][j] = a[j];
  body_sync<MODE>();
  for (int r = 0; r < n; ++r)
    if (i < n) X[(long)r * ld + i] = T[r][i];
  // identity padding of X (and zeros beside it) when the factorisation covers only the leading n of npad rows
  for (int r = 0; r < npad; ++r)
    if (i < npad && (r >= n || i >= n)) X[(long)r * ld + i] = (r == i) ? 1.0 : 0.0;
}
};

template <> struct ChainBody<CK_LEAF64> {      // leaf_potrf_inv_kernel<64>: one wave per matrix, the matrix in registers
  static constexpr int kBlock = 64;
  static constexpr bool kWaveParallel = false;     // (its LDS transpose buffer is one per workgroup: a chain runs the batch on wave 0)
  static constexpr int LF = 64;
  using Args = ArgList<double*, double*, int, long, int, int, int, int*, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* Aall, double* Xall, int ld, long stride, int off, int n, int do_chol, int* info, int npad) {
    (void)gridDim; (void)blockDim;
  double (&T)[LF][LF + 1] = *reinterpret_cast<double (*)[LF][LF + 1]>(body_smem<(LF) * (LF + 1) * 8, MODE>());
  double* A = Aall + (long)blockIdx.x * stride + (long)off * ld + off;
  double* X = Xall + (long)blockIdx.x * stride + (long)off * ld + off;
  const int i = threadIdx.x;
  for (int r = 0; r < n; ++r)
    if (i < n) T[r][i] = A[(long)r * ld + i];
  body_sync<MODE>();
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
    body_sync<MODE>();
    if (i < LF)
#pragma unroll
      for (int j = 0; j < LF; ++j) T[i][j] = a[j];
    body_sync<MODE>();
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
  body_sync<MODE>();
  if (i < LF)
#pragma unroll
    for (int j = 0; j < LF; ++j) T[i][j] = a[j];
  body_sync<MODE>();
  for (int r = 0; r < n; ++r)
    if (i < n) X[(long)r * ld + i] = T[r][i];
  // identity padding of X (and zeros beside it) when the factorisation covers only the leading n of npad rows
  for (int r = 0; r < npad; ++r)
    if (i < npad && (r >= n || i >= n)) X[(long)r * ld + i] = (r == i) ? 1.0 : 0.0;
}
};

template <> struct ChainBody<CK_ZERO1> {        // p[0 .. n) = 0 (the memset in front of an accumulating reduction)
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<double*, long>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, double* p, long n) {
    (void)gridDim;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
  }
};

template <> struct ChainBody<CK_COPY2D> {       // dst[r][c] = src[r][c], r < nr, c < nc, separate leading dimensions
  static constexpr int kBlock = 256;
  static constexpr bool kWaveParallel = false;
  using Args = ArgList<const double*, long, double*, long, int, int>;
  template <int MODE>
  __device__ static __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const VIdx threadIdx, const VIdx blockDim, const double* src, long lds,
                                             double* dst, long ldd, int nr, int nc) {
    (void)gridDim;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)nr * nc) return;
    const long r = idx / nc, c = idx % nc;
    dst[r * ldd + c] = src[r * lds + c];
  }
};

constexpr int SG = 64;          // largest M, N, K of the one-workgroup products
constexpr int SG_LD = SG + 2;   // row pitch of the LDS images (doubles): 16-byte aligned rows, odd multiple of 16 bytes

// C = alpha op(A) op(B) (+ C), M, N, K <= 64, one workgroup per batch member: both operands staged once (two 33 KB LDS
// images, k-major), every thread accumulates a 4 x 4 block of C (gemm_small.hip)
template <bool TA, bool TB, int MODE>
__device__ __forceinline__ void gemm_small_body(const VIdx blockIdx, const VIdx threadIdx, const GemmArgs& g) {
  char* sm_ = body_smem<2 * SG * SG_LD * 8, MODE>();
  double (&As)[SG][SG_LD] = *reinterpret_cast<double (*)[SG][SG_LD]>(sm_);                         // As[k][m] = op(A)[m][k]
  double (&Bs)[SG][SG_LD] = *reinterpret_cast<double (*)[SG][SG_LD]>(sm_ + SG * SG_LD * 8);        // Bs[k][n] = op(B)[k][n]
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const double* A = g.A + b * g.sA;
  const double* B = g.B + b * g.sB;
  double* C = g.C + b * g.sC;
  const int M = (int)g.M, N = (int)g.N, K = (int)g.K;
  // stage: the fast index of each global read runs along the operand's contiguous dimension.  All 32 loads of a thread are
  // requested before the first LDS write (clamped addresses + a select instead of a branch around the load: with the
  // branch the 16 rounds were 16 dependent trips to L2, 12 us per product)
  const int c = tid & 63, r0 = tid >> 6;
  const int ra_n = TA ? K : M, ca_n = TA ? M : K;      // rows / columns of A as stored
  const int rb_n = TB ? N : K, cb_n = TB ? K : N;      // rows / columns of B as stored
  double va[16], vb[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = r0 + 4 * it;
    va[it] = A[(long)(r < ra_n ? r : ra_n - 1) * g.lda + (c < ca_n ? c : ca_n - 1)];
    vb[it] = B[(long)(r < rb_n ? r : rb_n - 1) * g.ldb + (c < cb_n ? c : cb_n - 1)];
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = r0 + 4 * it;
    const double xa = (r < ra_n && c < ca_n) ? va[it] : 0.0, xb = (r < rb_n && c < cb_n) ? vb[it] : 0.0;
    if (TA) As[r][c] = xa; else As[c][r] = xa;         // As[k][m]
    if (TB) Bs[c][r] = xb; else Bs[r][c] = xb;         // Bs[k][n]
  }
  body_sync<MODE>();
  const int tx = tid & 15, ty = tid >> 4;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  if (4 * ty < M && 4 * tx < N) {
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const d2_t a0 = *reinterpret_cast<const d2_t*>(&As[k][4 * ty]), a1 = *reinterpret_cast<const d2_t*>(&As[k][4 * ty + 2]);
      const d2_t b0 = *reinterpret_cast<const d2_t*>(&Bs[k][4 * tx]), b1 = *reinterpret_cast<const d2_t*>(&Bs[k][4 * tx + 2]);
      const double av[4] = {a0[0], a0[1], a1[0], a1[1]}, bv[4] = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 4 * ty + i;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = 4 * tx + j;
        if (n >= N) continue;
        double* c = C + (long)m * g.ldc + n;
        const double v = g.alpha * acc[i][j];
        *c = g.beta ? *c + v : v;
      }
    }
  }
}

}  // namespace dgp
