// Per-point kernels of the SVGP-layer path (HBM-streaming stages around the MFMA contractions):
// RBF Kuf assembly, the variance/mean/sample stage, Gaussian variational expectations, and their
// hand-derived backward counterparts.  Point-major layouts: a point's M-vector
// is contiguous, so every kernel here reads and writes fully coalesced rows.
//
// Reference lines:  Kuf = covs.Kuf(Z, kern, X) layers.py:243;  mean/var layers.py:249-278;
// reparameterize utils.py:40-41;  Gaussian variational_expectations via utils.py:89-93 [ext];
// the backward is what tf.GradientTape derives in dgp.py:272-275.
#include "dgp_internal.h"

namespace dgp {

#define LAUNCH_CHECK() return hipGetLastError()

// ---------------------------------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

// N(0,1) keyed by (seed, global point n, sample s, layer, output d): identical for any sharding.
__device__ __forceinline__ double philox_normal(uint64_t seed, uint64_t n, uint32_t s, uint32_t layer, uint32_t d) {
  uint32_t c[4] = {(uint32_t)n, (uint32_t)(n >> 32), s, (layer << 16) | d};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const double two53 = 9007199254740992.0;
  const double u1 = ((double)(((uint64_t)(c[0] >> 5) << 26) + (uint64_t)(c[1] >> 6)) + 0.5) / two53;
  const double u2 = ((double)(((uint64_t)(c[2] >> 5) << 26) + (uint64_t)(c[3] >> 6)) + 0.5) / two53;
  return sqrt(-2.0 * log(u1)) * cos(2.0 * 3.14159265358979323846 * u2);
}

__device__ __forceinline__ double draw_z(const ZSource& zs, int s, long n_local, int d, int D) {
  if (zs.zs) return zs.zs[((long)s * zs.Ntot + n_local) * D + d];
  const uint64_t seed = zs.seed_dev ? *zs.seed_dev : zs.seed;
  return philox_normal(seed, (uint64_t)(zs.n_global0 + n_local), (uint32_t)s, (uint32_t)zs.layer, (uint32_t)d);
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum -> one atomic
__device__ __forceinline__ void block_atomic_add(double v, double* dst, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    unsafeAtomicAdd(dst, t);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------- Kuf (point-major)
// Kt[p][m] = k(x_p, z_m): var * exp(-0.5 |(x_p - z_m)/l|^2) for the squared exponential, Matern32/52 through
// stationary_k (reference: covs.Kuf, layers.py:243; gpflow kernels).  Et (optional) = -2 dk/d(r2), kept for the backward.
// One thread per inducing point m (its pre-scaled z row lives in registers), the block walks over points: the
// point's x is the same for every lane (scalar loads), so the kernel issues no per-element vector loads at all —
// the first version (16 loads per element) was load-issue bound at 3x this time.
template <int DIN>
__global__ __launch_bounds__(256) void rbf_kuf_kernel(const double* __restrict__ Xin, long P, long x_row0,
                                                      const double* __restrict__ Z, const double* __restrict__ var,
                                                      const double* __restrict__ ls, int M, int Mp, int Din,
                                                      double* __restrict__ Kt, int pts_per_block, int kind,
                                                      double* __restrict__ Et) {
  const int m = blockIdx.y * 256 + threadIdx.x;
  const int din = DIN > 0 ? DIN : Din;
  double zs[DIN > 0 ? DIN : 1], il[DIN > 0 ? DIN : 1];
  if constexpr (DIN > 0) {
#pragma unroll
    for (int j = 0; j < DIN; ++j) {
      il[j] = 1.0 / ls[j];
      zs[j] = (m < M) ? Z[(long)m * DIN + j] * il[j] : 0.0;
    }
  }
  const double v0 = var[0];
  const long p0 = (long)blockIdx.x * pts_per_block;
  const long p1 = min(P, p0 + pts_per_block);
#pragma unroll 2
  for (long p = p0; p < p1; ++p) {
    const double* __restrict__ x = Xin + (x_row0 + p) * din;       // uniform address: scalar loads
    double r2 = 0.0;
    if constexpr (DIN > 0) {
#pragma unroll
      for (int j = 0; j < DIN; ++j) { const double d = x[j] * il[j] - zs[j]; r2 += d * d; }
    } else {
      for (int j = 0; j < din; ++j) { const double d = (x[j] - ((m < M) ? Z[(long)m * din + j] : 0.0)) / ls[j]; r2 += d * d; }
    }
    if (m < Mp) {
      if (kind == 0) {
        Kt[p * Mp + m] = (m < M) ? v0 * exp_nonpos(-0.5 * r2) : 0.0;
      } else {
        double k, e;
        stationary_k(kind, v0, r2, k, e);
        Kt[p * Mp + m] = (m < M) ? k : 0.0;
        if (Et) Et[p * Mp + m] = (m < M) ? e : 0.0;
      }
    }
  }
}

// The same for Mp a multiple of 128 (the large-model shapes): TWO adjacent inducing points per thread, so that a lane stores
// 16 bytes, and the row written with a nontemporal hint - the 8-byte-per-lane stores of the kernel above reach 3.3 TB/s
// (the store path, not the exponentials, set its 0.73 ms per 10^6 x 256 values; tools/kuf_bench.hip: 0.62 ms with the
// exponential removed, 0.45 ms with the store removed); this form writes at 4.5-4.8 TB/s: 0.43-0.46 ms.  A block covers
// up to 256 columns; with TPR = columns / 2 threads per row its 256 threads take 256 / TPR point rows at a time, a row per
// whole waves (TPR >= 64), so the point's x stays a scalar load.
template <int DIN>
__global__ __launch_bounds__(256) void rbf_kuf2_kernel(const double* __restrict__ Xin, long P, long x_row0,
                                                       const double* __restrict__ Z, const double* __restrict__ var,
                                                       const double* __restrict__ ls, int M, int Mp, double* __restrict__ Kt,
                                                       int pts_per_block, int kind, double* __restrict__ Et, int tpr) {
  const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x / tpr);     // which of the block's concurrent rows
  const int nsub = 256 / tpr;
  const int m = blockIdx.y * 256 + 2 * ((int)threadIdx.x % tpr);
  double zs[2][DIN], il[DIN];
#pragma unroll
  for (int j = 0; j < DIN; ++j) {
    il[j] = 1.0 / ls[j];
    zs[0][j] = (m < M) ? Z[(long)m * DIN + j] * il[j] : 0.0;
    zs[1][j] = (m + 1 < M) ? Z[(long)(m + 1) * DIN + j] * il[j] : 0.0;
  }
  const double v0 = var[0];
  const long p0 = (long)blockIdx.x * pts_per_block;
  const long p1 = min(P, p0 + pts_per_block);
  for (long p = p0 + sub; p < p1; p += nsub) {
    const double* __restrict__ x = Xin + (x_row0 + p) * DIN;       // wave-uniform address: scalar loads
    double r0 = 0.0, r1 = 0.0;
#pragma unroll
    for (int j = 0; j < DIN; ++j) {
      const double xs = x[j] * il[j];
      const double d0 = xs - zs[0][j], d1 = xs - zs[1][j];
      r0 += d0 * d0; r1 += d1 * d1;
    }
    d2_t k, e;
    if (kind == 0) {
      k[0] = v0 * exp_nonpos(-0.5 * r0); k[1] = v0 * exp_nonpos(-0.5 * r1);
    } else {
      double k0, e0, k1, e1;
      stationary_k(kind, v0, r0, k0, e0);
      stationary_k(kind, v0, r1, k1, e1);
      k[0] = k0; k[1] = k1; e[0] = e0; e[1] = e1;
    }
    if (m >= M) k[0] = 0.0;
    if (m + 1 >= M) k[1] = 0.0;
    __builtin_nontemporal_store(k, reinterpret_cast<d2_t*>(Kt + p * Mp + m));
    if (kind != 0 && Et) {
      if (m >= M) e[0] = 0.0;
      if (m + 1 >= M) e[1] = 0.0;
      __builtin_nontemporal_store(e, reinterpret_cast<d2_t*>(Et + p * Mp + m));
    }
  }
}

template <int DIN>
static void launch_kuf2(hipStream_t st, int kind, const double* Xin, long P, long x_row0, const double* Z, const double* var,
                        const double* ls, int M, int Mp, double* Kt, double* Et) {
  const int cols = Mp % 256 == 0 ? 256 : 128, tpr = cols / 2;
  int ppb = 64;
  while (ppb > 256 / tpr && (P + ppb - 1) / ppb < 2048) ppb >>= 1;      // enough blocks to fill the chip
  dim3 grid((unsigned)((P + ppb - 1) / ppb), (unsigned)(Mp / cols));
  // (a block's column base is blockIdx.y * 256 in the kernel: with 128-column blocks only Mp = 128 takes this path)
  hipLaunchKernelGGL(rbf_kuf2_kernel<DIN>, grid, dim3(256), 0, st, Xin, P, x_row0, Z, var, ls, M, Mp, Kt, ppb, kind, Et, tpr);
}

hipError_t rbf_kuf(hipStream_t st, int kind, const double* Xin, long P, long x_row0, const double* Z, const double* var,
                   const double* ls, int M, int Mp, int Din, double* Kt, double* Et) {
  if (P == 0) return hipSuccess;
  if ((Mp % 256 == 0 || Mp == 128) && (reinterpret_cast<uintptr_t>(Kt) & 15u) == 0 && (reinterpret_cast<uintptr_t>(Et) & 15u) == 0) {
    bool done = true;
    switch (Din) {
      case 1: launch_kuf2<1>(st, kind, Xin, P, x_row0, Z, var, ls, M, Mp, Kt, Et); break;
      case 2: launch_kuf2<2>(st, kind, Xin, P, x_row0, Z, var, ls, M, Mp, Kt, Et); break;
      case 3: launch_kuf2<3>(st, kind, Xin, P, x_row0, Z, var, ls, M, Mp, Kt, Et); break;
      case 4: launch_kuf2<4>(st, kind, Xin, P, x_row0, Z, var, ls, M, Mp, Kt, Et); break;
      case 8: launch_kuf2<8>(st, kind, Xin, P, x_row0, Z, var, ls, M, Mp, Kt, Et); break;
      case 16: launch_kuf2<16>(st, kind, Xin, P, x_row0, Z, var, ls, M, Mp, Kt, Et); break;
      default: done = false; break;
    }
    if (done) LAUNCH_CHECK();
  }
  int ppb = 64;
  while (ppb > 1 && (P + ppb - 1) / ppb < 2048) ppb >>= 1;           // enough blocks to fill the chip
  dim3 grid((unsigned)((P + ppb - 1) / ppb), (unsigned)((Mp + 255) / 256));
  switch (Din) {
    case 1: hipLaunchKernelGGL(rbf_kuf_kernel<1>, grid, dim3(256), 0, st, Xin, P, x_row0, Z, var, ls, M, Mp, Din, Kt, ppb, kind, Et); break;
    case 2: hipLaunchKernelGGL(rbf_kuf_kernel<2>, grid, dim3(256), 0, st, Xin, P, x_row0, Z, var, ls, M, Mp, Din, Kt, ppb, kind, Et); break;
    case 3: hipLaunchKernelGGL(rbf_kuf_kernel<3>, grid, dim3(256), 0, st, Xin, P, x_row0, Z, var, ls, M, Mp, Din, Kt, ppb, kind, Et); break;
    case 4: hipLaunchKernelGGL(rbf_kuf_kernel<4>, grid, dim3(256), 0, st, Xin, P, x_row0, Z, var, ls, M, Mp, Din, Kt, ppb, kind, Et); break;
    case 8: hipLaunchKernelGGL(rbf_kuf_kernel<8>, grid, dim3(256), 0, st, Xin, P, x_row0, Z, var, ls, M, Mp, Din, Kt, ppb, kind, Et); break;
    case 16: hipLaunchKernelGGL(rbf_kuf_kernel<16>, grid, dim3(256), 0, st, Xin, P, x_row0, Z, var, ls, M, Mp, Din, Kt, ppb, kind, Et); break;
    default: hipLaunchKernelGGL(rbf_kuf_kernel<0>, grid, dim3(256), 0, st, Xin, P, x_row0, Z, var, ls, M, Mp, Din, Kt, ppb, kind, Et); break;
  }
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- var / mean / sample
// Finalises a layer from the GEMM epilogue partials:  cn = |c|^2 and tn_d = |W_d^T c|^2 arrive as per-wave
// row sums of squares (planes of P doubles), mean0 = c^T u_d from a skinny GEMM.
//   var = k_diag - |c|^2 + |W_d^T c|^2   (triangular form of layers.py:254-276)
//   mean = c^T u_d + mean_function(x)    (layers.py:249,278)
//   F = mean + z sqrt(var + jitter)      (utils.py:41)
// One thread per (d, p), p fastest (coalesced plane reads).
__global__ __launch_bounds__(256) void finalize_layer_kernel(
    const double* __restrict__ cnp, const double* __restrict__ tnp, int nplane, int nplane_t, long pstride,
    const double* __restrict__ mean0,
    const double* __restrict__ Xin, long x_row0, long P, long Nc, int S, int dedup, int Din, int D,
    const double* __restrict__ kvar, int mean_kind, const double* __restrict__ meanW, const double* __restrict__ meanb,
    ZSource zsrc, long n_chunk0, double* __restrict__ mean, double* __restrict__ var, double* __restrict__ F,
    int kernel_kind, const double* __restrict__ white) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * D) return;
  const int d = (int)(idx / P);
  const long p = idx % P;
  double cn = 0.0, tn = 0.0;
  for (int q = 0; q < nplane; ++q) cn += cnp[(long)q * pstride + p];
  for (int q = 0; q < nplane_t; ++q) tn += tnp[((long)d * nplane_t + q) * pstride + p];
  const double* x = Xin + (x_row0 + p) * Din;
  double mf = 0.0;
  if (mean_kind == 1) mf = x[d];
  else if (mean_kind == 2) {
    for (int j = 0; j < Din; ++j) mf += x[j] * meanW[(long)j * D + d];
    mf += meanb[d];
  }
  double kd = kvar[0];
  if (kernel_kind == 3) {             // DGP_KERNEL_MF: K_diag = vC (vP + vL f^2) + vI, f = last input column
    const double f = x[Din - 1];
    kd = kvar[0] * (kvar[2] + kvar[4] * f * f) + kvar[5];
  }
  if (white) kd += white[0];
  const double mv = mean0[p * D + d] + mf, vv = kd - cn + tn;
  mean[p * D + d] = mv;
  var[p * D + d] = vv;
  if (F) {
    const double sd = sqrt(vv + kJitter);
    if (dedup) {
      for (int s = 0; s < S; ++s) F[((long)s * Nc + p) * D + d] = mv + draw_z(zsrc, s, n_chunk0 + p, d, D) * sd;
    } else {
      const int s = (int)(p / Nc);
      const long i = p % Nc;
      F[p * D + d] = mv + draw_z(zsrc, s, n_chunk0 + i, d, D) * sd;
    }
  }
}

// The same stage for large P, every HBM access coalesced: a block takes 256 consecutive points.  Phase 1: thread = point -
// sums the row-norm planes (reads with p fastest) and leaves |c|^2 and up to 8 of the |t_d|^2 in LDS.  Phase 2: thread =
// (point, output) in memory order of the [P, D] arrays - mean0 / mean / var / F are read and written as contiguous rows
// (the per-point kernel above wrote them 8 bytes per lane at a stride of D doubles: 0.45 ms at P = 1e6, D = 8; the planes
// are summed in the same order, so the results are bit-identical).
__global__ __launch_bounds__(256) void finalize_layer_tiled_kernel(
    const double* __restrict__ cnp, const double* __restrict__ tnp, int nplane, int nplane_t, long pstride,
    const double* __restrict__ mean0,
    const double* __restrict__ Xin, long x_row0, long P, long Nc, int S, int dedup, int Din, int D,
    const double* __restrict__ kvar, int mean_kind, const double* __restrict__ meanW, const double* __restrict__ meanb,
    ZSource zsrc, long n_chunk0, double* __restrict__ mean, double* __restrict__ var, double* __restrict__ F,
    int kernel_kind, const double* __restrict__ white) {
  __shared__ double s_cn[256];
  __shared__ double s_tn[8][257];
  const long pb = (long)blockIdx.x * 256;
  const int t = threadIdx.x;
  const long pt = pb + t;
  {
    double cn = 0.0;
    if (pt < P)
      for (int q = 0; q < nplane; ++q) cn += cnp[(long)q * pstride + pt];
    s_cn[t] = cn;
  }
  const int npts = (int)min((long)256, P - pb);
  for (int d0 = 0; d0 < D; d0 += 8) {
    const int dn = min(8, D - d0);
    __syncthreads();
    if (pt < P)
      for (int dd = 0; dd < dn; ++dd) {
        double tn = 0.0;
        for (int q = 0; q < nplane_t; ++q) tn += tnp[((long)(d0 + dd) * nplane_t + q) * pstride + pt];
        s_tn[dd][t] = tn;
      }
    __syncthreads();
    for (int o = t; o < npts * dn; o += 256) {
      const int pl = o / dn, dd = o - pl * dn, d = d0 + dd;
      const long p = pb + pl;
      const double* x = Xin + (x_row0 + p) * Din;
      double mf = 0.0;
      if (mean_kind == 1) mf = x[d];
      else if (mean_kind == 2) {
        for (int j = 0; j < Din; ++j) mf += x[j] * meanW[(long)j * D + d];
        mf += meanb[d];
      }
      double kd = kvar[0];
      if (kernel_kind == 3) {
        const double f = x[Din - 1];
        kd = kvar[0] * (kvar[2] + kvar[4] * f * f) + kvar[5];
      }
      if (white) kd += white[0];
      const double mv = mean0[p * D + d] + mf, vv = kd - s_cn[pl] + s_tn[dd][pl];
      mean[p * D + d] = mv;
      var[p * D + d] = vv;
      if (F) {
        const double sd = sqrt(vv + kJitter);
        if (dedup) {
          for (int s = 0; s < S; ++s) F[((long)s * Nc + p) * D + d] = mv + draw_z(zsrc, s, n_chunk0 + p, d, D) * sd;
        } else {
          const int s1 = (int)(p / Nc);
          const long i1 = p % Nc;
          F[p * D + d] = mv + draw_z(zsrc, s1, n_chunk0 + i1, d, D) * sd;
        }
      }
    }
  }
}

// F[s][p][d] = mean[p][d] + z sqrt(var[p][d] + jitter) for the S samples of the first layer's points, one thread per value
__global__ __launch_bounds__(256) void expand_f_kernel(const double* __restrict__ mean, const double* __restrict__ var, long Nc,
                                                       int S, int D, ZSource zsrc, long n_chunk0, double* __restrict__ F) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nd = Nc * D;
  if (idx >= (long)S * nd) return;
  const int s = (int)(idx / nd);
  const long r = idx - (long)s * nd, p = r / D;
  const int d = (int)(r - p * D);
  F[idx] = mean[r] + draw_z(zsrc, s, n_chunk0 + p, d, D) * sqrt(var[r] + kJitter);
}

hipError_t finalize_layer(hipStream_t st, const double* cnp, const double* tnp, int nplane, int nplane_t, long pstride,
                          const double* mean0,
                          const double* Xin, long x_row0, long P, long Nc, int S, int dedup, int Din, int D,
                          const double* kvar, int mean_kind, const double* meanW, const double* meanb, ZSource zsrc,
                          long n_chunk0, double* mean, double* var, double* F, int kernel_kind, const double* white) {
  const long n = P * D;
  if (n == 0) return hipSuccess;
  // first layer with many samples (the acquisition side draws S = 1000 samples of a handful of candidates): the kernels
  // above would walk the S samples of a point in one thread (0.75 ms for one candidate) - they leave F to expand_f_kernel
  const bool wide_s = dedup && F != nullptr && S > 16;
  double* Fk = wide_s ? nullptr : F;
  if (P >= 32768 && D > 1)
    hipLaunchKernelGGL(finalize_layer_tiled_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, cnp, tnp, nplane, nplane_t, pstride,
                       mean0, Xin, x_row0, P, Nc, S, dedup, Din, D, kvar, mean_kind, meanW, meanb, zsrc, n_chunk0, mean, var, Fk,
                       kernel_kind, white);
  else
    hipLaunchKernelGGL(finalize_layer_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cnp, tnp, nplane, nplane_t, pstride, mean0,
                       Xin, x_row0, P, Nc, S, dedup, Din, D, kvar, mean_kind, meanW, meanb, zsrc, n_chunk0, mean, var, Fk,
                       kernel_kind, white);
  if (wide_s) {
    const long tot = (long)S * Nc * D;
    hipLaunchKernelGGL(expand_f_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, mean, var, Nc, S, D, zsrc, n_chunk0, F);
  }
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- Gaussian likelihood
// ELBO data term  sum_n mean_s VE   (dgp.py:79-87,96) and the seeds d/dmean, d/dvar of the backward.
__global__ __launch_bounds__(256) void gauss_lik_kernel(const double* __restrict__ mean, const double* __restrict__ var,
                                                        const double* __restrict__ Y, long y_row0, long Nc, int S,
                                                        int dedup, int Dy, const double* __restrict__ lik_var,
                                                        double* __restrict__ acc_elbo, double* __restrict__ acc_dlik,
                                                        double* __restrict__ mbar, double* __restrict__ vbar,
                                                        double* __restrict__ acc_dkvar, double scale) {
  __shared__ double sh[4];
  const long P = dedup ? Nc : (long)S * Nc;
  const long total = P * Dy;
  const double s2 = lik_var[0];
  const double w = scale * (dedup ? 1.0 : 1.0 / (double)S);     // scale: N / batch size of a minibatch (dgp.py:95-99)
  double e = 0.0, dl = 0.0, dk = 0.0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long p = idx / Dy;
    const int d = (int)(idx % Dy);
    const long i = p % Nc;
    const double y = Y[(y_row0 + i) * Dy + d];
    const double r = y - mean[idx], v = var[idx];
    const double q = r * r + v;
    e += w * (-0.91893853320467274178 - 0.5 * log(s2) - 0.5 * q / s2);
    dl += w * (-0.5 / s2 + 0.5 * q / (s2 * s2));
    if (mbar) {
      mbar[idx] = w * r / s2;
      const double vb = -0.5 * w / s2;
      vbar[idx] = vb;
      dk += vb;
    }
  }
  block_atomic_add(e, acc_elbo, sh);
  if (mbar) {
    block_atomic_add(dl, acc_dlik, sh);
    block_atomic_add(dk, acc_dkvar, sh);
  }
}

hipError_t gauss_lik(hipStream_t st, const double* mean, const double* var, const double* Y, long y_row0, long Nc, int S,
                     int dedup, int Dy, const double* lik_var, double* acc_elbo, double* acc_dlik, double* mbar,
                     double* vbar, double* acc_dkvar, double scale) {
  const long total = (dedup ? Nc : (long)S * Nc) * Dy;
  if (total == 0) return hipSuccess;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gauss_lik_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mean, var, Y, y_row0, Nc, S, dedup, Dy,
                     lik_var, acc_elbo, acc_dlik, mbar, vbar, acc_dkvar, scale);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- fold dF into (mbar, vbar)
// F = mean + z sqrt(var + eps):  mbar = dF,  vbar = dF z / (2 sqrt(var + eps));  summed over s when the
// layer's mean/var are shared by all samples (first layer).
__global__ __launch_bounds__(256) void fold_kernel(const double* __restrict__ Fbar, const double* __restrict__ var,
                                                   long Nc, int S, int dedup, int D, ZSource zsrc, long n_chunk0,
                                                   double* __restrict__ mbar, double* __restrict__ vbar,
                                                   double* __restrict__ acc_dkvar) {
  __shared__ double sh[4];
  const long P = dedup ? Nc : (long)S * Nc;
  const long total = P * D;
  double dk = 0.0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long p = idx / D;
    const int d = (int)(idx % D);
    const double inv = 0.5 / sqrt(var[idx] + kJitter);
    double mb = 0.0, vb = 0.0;
    if (dedup) {
      for (int s = 0; s < S; ++s) {
        const double fb = Fbar[((long)s * Nc + p) * D + d];
        mb += fb;
        vb += fb * draw_z(zsrc, s, n_chunk0 + p, d, D) * inv;
      }
    } else {
      const int s = (int)(p / Nc);
      const long i = p % Nc;
      const double fb = Fbar[idx];
      mb = fb;
      vb = fb * draw_z(zsrc, s, n_chunk0 + i, d, D) * inv;
    }
    mbar[idx] = mb;
    vbar[idx] = vb;
    dk += vb;
  }
  if (acc_dkvar) block_atomic_add(dk, acc_dkvar, sh);
}

// The first layer's fold with many samples: one wave per (point, output) strides over the S samples (the loop above
// in one thread took 0.8 ms for one candidate at S = 1000).
__global__ __launch_bounds__(256) void fold_dedup_wide_kernel(const double* __restrict__ Fbar, const double* __restrict__ var,
                                                              long Nc, int S, int D, ZSource zsrc, long n_chunk0,
                                                              double* __restrict__ mbar, double* __restrict__ vbar,
                                                              double* __restrict__ acc_dkvar) {
  __shared__ double sh[4];
  const long pd = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double vb = 0.0;
  if (pd < Nc * D) {
    const long p = pd / D;
    const int d = (int)(pd - p * D);
    const double inv = 0.5 / sqrt(var[pd] + kJitter);
    double mb = 0.0;
    for (int s = lane; s < S; s += 64) {
      const double fb = Fbar[((long)s * Nc + p) * D + d];
      mb += fb;
      vb += fb * draw_z(zsrc, s, n_chunk0 + p, d, D) * inv;
    }
    mb = wave_sum(mb);
    vb = wave_sum(vb);
    if (lane == 0) { mbar[pd] = mb; vbar[pd] = vb; }
    if (lane != 0) vb = 0.0;
  }
  if (acc_dkvar) block_atomic_add(vb, acc_dkvar, sh);
}

hipError_t fold_sample_grad(hipStream_t st, const double* Fbar, const double* var, long Nc, int S, int dedup, int D,
                            ZSource zsrc, long n_chunk0, double* mbar, double* vbar, double* acc_dkvar) {
  const long total = (dedup ? Nc : (long)S * Nc) * D;
  if (total == 0) return hipSuccess;
  if (dedup && S > 16) {
    hipLaunchKernelGGL(fold_dedup_wide_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, Fbar, var, Nc, S, D, zsrc,
                       n_chunk0, mbar, vbar, acc_dkvar);
    LAUNCH_CHECK();
  }
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, Fbar, var, Nc, S, dedup, D, zsrc, n_chunk0,
                     mbar, vbar, acc_dkvar);
  LAUNCH_CHECK();
}

// Seeds of the vector-Jacobian product through the last layer (acquisition side; reference: tf.GradientTape on x,
// Infill_criteria.py:79-85).  With cotangents of the layer's sample F, mean and variance ([S, Ntot, D], any may be
// null):  mbar = meanbar + Fbar,  vbar = varbar + Fbar z / (2 sqrt(var + eps));  summed over s for a single-layer model.
__global__ __launch_bounds__(256) void vjp_seed_kernel(const double* __restrict__ fbar, const double* __restrict__ meanbar,
                                                       const double* __restrict__ varbar, const double* __restrict__ var,
                                                       long Nc, int S, int dedup, int D, ZSource zsrc, long Ntot,
                                                       long n_chunk0, double* __restrict__ mbar, double* __restrict__ vbar,
                                                       double* __restrict__ acc_dkvar) {
  __shared__ double sh[4];
  double dk = 0.0;
  const long P = dedup ? Nc : (long)S * Nc;
  const long total = P * D;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long p = idx / D;
    const int d = (int)(idx % D);
    const double inv = 0.5 / sqrt(var[idx] + kJitter);
    double mb = 0.0, vb = 0.0;
    const int s_lo = dedup ? 0 : (int)(p / Nc), s_hi = dedup ? S : s_lo + 1;
    const long i = dedup ? p : p % Nc;
    for (int s = s_lo; s < s_hi; ++s) {
      const long src = ((long)s * Ntot + n_chunk0 + i) * D + d;
      if (meanbar) mb += meanbar[src];
      if (varbar) vb += varbar[src];
      if (fbar) {
        const double fb = fbar[src];
        mb += fb;
        vb += fb * draw_z(zsrc, s, n_chunk0 + i, d, D) * inv;
      }
    }
    mbar[idx] = mb;
    vbar[idx] = vb;
    dk += vb;
  }
  if (acc_dkvar) block_atomic_add(dk, acc_dkvar, sh);
}
// single-layer model with many samples: one wave per (point, output) strides over the S samples (as fold_dedup_wide_kernel)
__global__ __launch_bounds__(256) void vjp_seed_dedup_wide_kernel(const double* __restrict__ fbar, const double* __restrict__ meanbar,
                                                                  const double* __restrict__ varbar, const double* __restrict__ var,
                                                                  long Nc, int S, int D, ZSource zsrc, long Ntot, long n_chunk0,
                                                                  double* __restrict__ mbar, double* __restrict__ vbar,
                                                                  double* __restrict__ acc_dkvar) {
  __shared__ double sh[4];
  const long pd = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double vb = 0.0;
  if (pd < Nc * D) {
    const long p = pd / D;
    const int d = (int)(pd - p * D);
    const double inv = 0.5 / sqrt(var[pd] + kJitter);
    double mb = 0.0;
    for (int s = lane; s < S; s += 64) {
      const long src = ((long)s * Ntot + n_chunk0 + p) * D + d;
      if (meanbar) mb += meanbar[src];
      if (varbar) vb += varbar[src];
      if (fbar) {
        const double fb = fbar[src];
        mb += fb;
        vb += fb * draw_z(zsrc, s, n_chunk0 + p, d, D) * inv;
      }
    }
    mb = wave_sum(mb);
    vb = wave_sum(vb);
    if (lane == 0) { mbar[pd] = mb; vbar[pd] = vb; }
    if (lane != 0) vb = 0.0;
  }
  if (acc_dkvar) block_atomic_add(vb, acc_dkvar, sh);
}

hipError_t vjp_seed(hipStream_t st, const double* fbar, const double* meanbar, const double* varbar, const double* var,
                    long Nc, int S, int dedup, int D, ZSource zsrc, long Ntot, long n_chunk0, double* mbar, double* vbar,
                    double* acc_dkvar) {
  const long total = (dedup ? Nc : (long)S * Nc) * D;
  if (total == 0) return hipSuccess;
  if (dedup && S > 16) {
    hipLaunchKernelGGL(vjp_seed_dedup_wide_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, fbar, meanbar, varbar, var, Nc,
                       S, D, zsrc, Ntot, n_chunk0, mbar, vbar, acc_dkvar);
    LAUNCH_CHECK();
  }
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(vjp_seed_kernel, dim3((unsigned)blocks), dim3(256), 0, st, fbar, meanbar, varbar, var, Nc, S, dedup, D,
                     zsrc, Ntot, n_chunk0, mbar, vbar, acc_dkvar);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- dC += mbar u^T
// dC_p = u mbar_p + sum_d 2 vbar_pd (W_d W_d^T - I) c_p   (whitened); the second term is
// the GEMM  [2 vbar .* C] * S'cat  with S'_d = W_d W_d^T - I; this kernel adds the first.
__global__ void cbar_fix_kernel(double* __restrict__ Cbar, const double* __restrict__ mbar, const double* __restrict__ u,
                                long P, int Mp, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * Mp) return;
  const long p = idx / Mp;
  const int m = (int)(idx % Mp);
  double a = 0.0;
  for (int d = 0; d < D; ++d) a += mbar[p * D + d] * u[(long)m * D + d];
  Cbar[idx] += a;
}
hipError_t cbar_fix(hipStream_t st, double* Cbar, const double* mbar, const double* u, long P, int Mp, int D) {
  const long n = P * Mp;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(cbar_fix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Cbar, mbar, u, P, Mp, D);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- RBF backward (Kuf)
// With g = dK .* K (second output of the dK GEMM):  
//   dx_pj = (sum_m g z_mj - x_pj sum_m g) / l_j^2 (+ mean-function path)     <- R1 = g [Z | 1]
//   dz_mj = -(z_mj sum_p g - sum_p g x_pj) / l_j^2                           <- GX = g^T [X | 1]  (over points)
//   dl_j  = (sum_m z_mj^2 cs_m - 2 sum_m z_mj GX_mj + sum_p x_pj^2 rs_p) / l_j^3,   dvar = sum_m cs_m / var
// so the contractions run in the GEMM engine and only thin element-wise kernels remain.
__global__ void make_x1_kernel(const double* __restrict__ Xin, long x_row0, long P, int Din, double* __restrict__ X1) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = Din + 1;
  if (idx >= P * w) return;
  const long p = idx / w;
  const int j = (int)(idx % w);
  X1[idx] = (j < Din) ? Xin[(x_row0 + p) * Din + j] : 1.0;
}
hipError_t make_x1(hipStream_t st, const double* Xin, long x_row0, long P, int Din, double* X1) {
  const long n = P * (Din + 1);
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(make_x1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Xin, x_row0, P, Din, X1);
  LAUNCH_CHECK();
}

// R1 = g [Z | 1] and GX += g^T [X | 1] in ONE pass over g (Mp = 256, D_in <= 8).  As two skinny products on the 128 x 64 engine
// each of them streamed g (2 GB per 10^6 points) at HBM speed: 0.41 + 0.36 ms.  Here a workgroup walks a contiguous range of
// 16-point tiles: thread t loads column t of the tile's 16 rows (coalesced 2 KB rows, the next tile's 16 loads in flight while
// this one is worked on) and parks them in LDS; then, on the matrix cores (v_mfma_f64_4x4x4_4b_f64, lane maps: gemm_f64.h),
//   R1 tile [16 x 12] = g tile [16 x 256] Z1p [256 x 12]: the 64 k-steps split over the four waves, partials met in LDS;
//   GX [256 x 12]    += g tile^T [256 x 16] X1p tile [16 x 12]: wave w owns columns 64 w .. 64 w + 63, accumulators kept over
//                       the workgroup's whole range, written once to a slab that rg_reduce_kernel adds in a fixed order.
// "p": the W <= 9 real columns sit at 4 (j / 3) + j % 3 of 12 (the e-th of three MFMAs yields columns 4 q + e).
// (A first version did both on the VALU, one LDS operand read per FMA: LDS-bound, 1.07 ms per 10^6 points.)
constexpr int RG_TP = 16, RG_PITCH = 260;
template <int W>
__global__ __launch_bounds__(256) void rg_contract_kernel(const double* __restrict__ G, long P, const double* __restrict__ Z1,
                                                          const double* __restrict__ X1, double* __restrict__ R1,
                                                          double* __restrict__ slab, long tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) double sG[RG_TP * RG_PITCH];
  __shared__ __attribute__((aligned(16))) double sZ[256 * 12];
  __shared__ __attribute__((aligned(16))) double sX[RG_TP * 12];
  __shared__ double sP[4 * 64 * 3];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lk = lane >> 4;
  for (int x = t; x < 256 * 12; x += 256) {
    const int m = x / 12, jj = x - m * 12, j = 3 * (jj >> 2) + (jj & 3);
    sZ[x] = ((jj & 3) < 3 && j < W) ? Z1[m * W + j] : 0.0;
  }
  if (t < RG_TP * 12) sX[t] = 0.0;
  double gacc[4][3];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int e = 0; e < 3; ++e) gacc[b][e] = 0.0;
  const long ntile = (P + RG_TP - 1) / RG_TP;
  const long t0 = (long)blockIdx.x * tiles_per_wg;
  const long t1 = t0 + tiles_per_wg < ntile ? t0 + tiles_per_wg : ntile;
  // thread -> element of the X1 tile it stages (t < 16 W) and of the R1 tile it writes
  const int xp = t / W, xj = t - xp * W, xjj = 4 * (xj / 3) + xj % 3;
  double gv[RG_TP], xv = 0.0;
  auto load_tile = [&](long tile) __attribute__((always_inline)) {
    const long p0 = tile * RG_TP;
    const int np = (int)(P - p0 < RG_TP ? P - p0 : RG_TP);
#pragma unroll
    for (int pp = 0; pp < RG_TP; ++pp) gv[pp] = pp < np ? G[(p0 + pp) * 256 + t] : 0.0;
    xv = (t < RG_TP * W && xp < np) ? X1[(p0 + xp) * W + xj] : 0.0;
  };
  if (t0 < t1) load_tile(t0);
  for (long tile = t0; tile < t1; ++tile) {
    const long p0 = tile * RG_TP;
    const int np = (int)(P - p0 < RG_TP ? P - p0 : RG_TP);
    __syncthreads();              // (the previous tile's MFMAs have left sG / sX / sP; first tile: sZ is in place)
#pragma unroll
    for (int pp = 0; pp < RG_TP; ++pp) sG[pp * RG_PITCH + t] = gv[pp];
    if (t < RG_TP * W) sX[xp * 12 + xjj] = xv;
    __syncthreads();
    if (tile + 1 < t1) load_tile(tile + 1);
    // ---- R1 partial of this wave: k-steps 16 wave .. 16 wave + 15 (columns m = 4 s + lk of g)
    double racc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int ss = 0; ss < 16; ++ss) {
      const int s = 16 * wave + ss;
      const double a = sG[li * RG_PITCH + 4 * s + lk];
      const double* zp = sZ + (4 * s + lk) * 12 + 4 * (li & 3);
      const d2_t z01 = *reinterpret_cast<const d2_t*>(zp);
      const double z2 = zp[2];
      racc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, z01[0], racc[0], 0, 0, 0);
      racc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, z01[1], racc[1], 0, 0, 0);
      racc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, z2, racc[2], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) sP[(wave * 64 + lane) * 3 + e] = racc[e];
    // ---- GX: this wave's four 16-column blocks, k = the tile's 16 points
    if (slab != nullptr) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double* xq = sX + (4 * s + lk) * 12 + 4 * (li & 3);
        const d2_t x01 = *reinterpret_cast<const d2_t*>(xq);
        const double x2 = xq[2];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double a = sG[(4 * s + lk) * RG_PITCH + 16 * (4 * wave + b) + li];
          gacc[b][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x01[0], gacc[b][0], 0, 0, 0);
          gacc[b][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x01[1], gacc[b][1], 0, 0, 0);
          gacc[b][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x2, gacc[b][2], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    // R1[p0 + xp][xj] = the four waves' partials of lane (c = 4 (xp >> 2) + xj / 3, lk = xp & 3), MFMA e = xj % 3
    if (t < RG_TP * W && xp < np) {
      const int ln = 4 * (xp >> 2) + xj / 3 + 16 * (xp & 3), e = xj % 3;
      R1[(p0 + xp) * W + xj] = (sP[(0 * 64 + ln) * 3 + e] + sP[(1 * 64 + ln) * 3 + e]) + (sP[(2 * 64 + ln) * 3 + e] + sP[(3 * 64 + ln) * 3 + e]);
    }
  }
  if (slab != nullptr) {
    // gacc[b][e] of lane (c = li, lk) is GX[16 (4 wave + b) + 4 (c >> 2) + lk][3 (c & 3) + e]
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int m = 16 * (4 * wave + b) + 4 * (li >> 2) + lk, j = 3 * (li & 3) + e;
        if (j < W) slab[((long)blockIdx.x * 256 + m) * W + j] = gacc[b][e];
      }
  }
}
// GX[m][j] += sum over the workgroups' slabs, in workgroup order; one thread per (m, j)
__global__ __launch_bounds__(256) void rg_reduce_kernel(const double* __restrict__ slab, int nslab, int n, double* __restrict__ GX) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = 0;
  for (; b + 4 <= nslab; b += 4) {
    s0 += slab[(long)(b + 0) * n + i]; s1 += slab[(long)(b + 1) * n + i];
    s2 += slab[(long)(b + 2) * n + i]; s3 += slab[(long)(b + 3) * n + i];
  }
  for (; b < nslab; ++b) s0 += slab[(long)b * n + i];
  GX[i] += (s0 + s1) + (s2 + s3);
}
// (also the reduction of gemm_gpanel.h's slabs)
hipError_t rg_reduce(hipStream_t st, const double* slab, int nslab, int n, double* GX) {
  hipLaunchKernelGGL(rg_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, nslab, n, GX);
  LAUNCH_CHECK();
}
constexpr int RG_MAX_WG = 512;      // two workgroups per CU (65 KB of LDS each)
long rbf_bwd_contract_ws_doubles() { return (long)RG_MAX_WG * 256 * 9; }
bool rbf_bwd_contract_ok(int Mp, int w1, long P) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("DGP_RG_FUSED"); enabled = e ? atoi(e) : 1; }
  return enabled && Mp == 256 && w1 >= 2 && w1 <= 9 && P >= 2048;
}
// R1 [P x w1] = G Z1; GX [256 x w1] += G^T X1 when GX != nullptr (ws: rbf_bwd_contract_ws_doubles() doubles)
hipError_t rbf_bwd_contract(hipStream_t st, const double* G, long P, const double* Z1, const double* X1, int w1, double* R1,
                            double* GX, double* ws) {
  if (P <= 0) return hipSuccess;
  if (GX != nullptr && ws == nullptr) return hipErrorInvalidValue;
  const long ntile = (P + RG_TP - 1) / RG_TP;
  const long grid = ntile < RG_MAX_WG ? ntile : RG_MAX_WG;
  const long per = (ntile + grid - 1) / grid;
  const long used = (ntile + per - 1) / per;          // workgroups that get tiles (the others would write zero slabs)
  double* slab = GX ? ws : nullptr;
#define RG_LAUNCH(W) hipLaunchKernelGGL(rg_contract_kernel<W>, dim3((unsigned)used), dim3(256), 0, st, G, P, Z1, X1, R1, slab, per)
  switch (w1) {
    case 2: RG_LAUNCH(2); break; case 3: RG_LAUNCH(3); break; case 4: RG_LAUNCH(4); break; case 5: RG_LAUNCH(5); break;
    case 6: RG_LAUNCH(6); break; case 7: RG_LAUNCH(7); break; case 8: RG_LAUNCH(8); break; case 9: RG_LAUNCH(9); break;
    default: return hipErrorInvalidValue;
  }
#undef RG_LAUNCH
  if (GX) {
    const int n = 256 * w1;
    hipLaunchKernelGGL(rg_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, (int)used, n, GX);
  }
  LAUNCH_CHECK();
}

// x-gradient of the stationary kernels and the sum over the points of x_j^2 rowsum(g) (the lengthscale gradient's second
// part).  One thread per (point, input j) in memory order of the [P, Din] arrays: xbar, mbar (identity mean) and the rows of
// R1 / X1 are touched as contiguous pieces (the first version gave every j its own blocks, each striding over 72-byte rows:
// 0.2-0.3 ms per 10^6 points against 0.06 ms of traffic).  The per-j sums meet in LDS, one global atomic per block and j.
__global__ __launch_bounds__(256) void xbar_finish_kernel(const double* __restrict__ R1, const double* __restrict__ X1, long P,
                                                          const double* __restrict__ ls, int Din, int D, int mean_kind,
                                                          const double* __restrict__ meanW, const double* __restrict__ mbar,
                                                          int want_xbar, double* __restrict__ xbar,
                                                          double* __restrict__ acc_x2rs) {
  __shared__ double s_x2[64];
  const int w = Din + 1;
  if (threadIdx.x < 64) s_x2[threadIdx.x] = 0.0;
  __syncthreads();
  const long total = P * Din;
  const long stride = (long)gridDim.x * blockDim.x;
  const bool fixed_j = stride % Din == 0;        // a thread then keeps its j: its sum stays in a register until the end
  double x2 = 0.0;
  int j_last = 0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const long p = idx / Din;
    const int j = (int)(idx - p * Din);
    j_last = j;
    const double l = ls[j], il2 = 1.0 / (l * l);
    const double rs = R1[p * w + Din], x = X1[p * w + j];
    if (acc_x2rs) {
      if (fixed_j) x2 += x * x * rs;
      else atomicAdd(&s_x2[j], x * x * rs);
    }
    if (want_xbar) {
      double a = (R1[p * w + j] - x * rs) * il2;
      const double* mb = mbar + p * D;
      if (mean_kind == 1) a += mb[j];
      else if (mean_kind == 2)
        for (int d = 0; d < D; ++d) a += meanW[(long)j * D + d] * mb[d];
      xbar[idx] = a;
    }
  }
  if (acc_x2rs && fixed_j && x2 != 0.0) atomicAdd(&s_x2[j_last], x2);
  __syncthreads();
  if (acc_x2rs && (int)threadIdx.x < Din) unsafeAtomicAdd(acc_x2rs + threadIdx.x, s_x2[threadIdx.x]);
}
hipError_t xbar_finish(hipStream_t st, const double* R1, const double* X1, long P, const double* ls, int Din, int D,
                       int mean_kind, const double* meanW, const double* mbar, int want_xbar, double* xbar,
                       double* acc_x2rs) {
  if (P == 0) return hipSuccess;
  if (Din > 64) return hipErrorInvalidValue;
  long blocks = (P * Din + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(xbar_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, R1, X1, P, ls, Din, D, mean_kind, meanW,
                     mbar, want_xbar, xbar, acc_x2rs);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- output helpers
__global__ void expand_rows_kernel(const double* __restrict__ src, long Nc, int S, int D, int dedup,
                                   double* __restrict__ dst, long Ntot, long n0) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)S * Nc * D;
  if (idx >= total) return;
  const int d = (int)(idx % D);
  const long r = idx / D;
  const long s = r / Nc, i = r % Nc;
  dst[(s * Ntot + n0 + i) * D + d] = src[(dedup ? i : r) * D + d];
}
hipError_t expand_rows(hipStream_t st, const double* src, long Nc, int S, int D, int dedup, double* dst, long Ntot,
                       long n0) {
  const long n = (long)S * Nc * D;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, Nc, S, D, dedup, dst,
                     Ntot, n0);
  LAUNCH_CHECK();
}

__global__ void add_scalar_kernel(double* __restrict__ v, long n, const double* __restrict__ s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] += s[0];
}
hipError_t lik_predict_var(hipStream_t st, double* var, long n, const double* lik_var) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(add_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, var, n, lik_var);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- full covariance (small N)
// Pieces of the full_cov=True branches of layers.py:77-80,265-268 and utils.py:43-51, one sample s at a time.
// mean[i][d] = mean0[i][d] + mean_function(x_i)
__global__ void fc_mean_kernel(const double* __restrict__ mean0, const double* __restrict__ X, long N, int Din, int D,
                               int mean_kind, const double* __restrict__ meanW, const double* __restrict__ meanb,
                               double* __restrict__ mean) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const long i = idx / D;
  const int d = (int)(idx % D);
  double mf = 0.0;
  if (mean_kind == 1) mf = X[i * Din + d];
  else if (mean_kind == 2) {
    for (int j = 0; j < Din; ++j) mf += X[i * Din + j] * meanW[(long)j * D + d];
    mf += meanb[d];
  }
  mean[idx] = mean0[idx] + mf;
}
// var_out[i][j][d] = V_d[i][j] - jitter * (i == j)      (V carries the jitter of utils.py:47 for its Cholesky)
__global__ void fc_export_var_kernel(const double* __restrict__ V, long N, int Np, int D, double* __restrict__ var_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * N * D) return;
  const int d = (int)(idx % D);
  const long ij = idx / D, i = ij / N, j = ij % N;
  var_out[idx] = V[((long)d * Np + i) * Np + j] - (i == j ? kJitter : 0.0);
}
// F[i][d] = mean[i][d] + sum_{j <= i} L_d[i][j] z[s][j][d]
__global__ void fc_sample_kernel(const double* __restrict__ L, const double* __restrict__ mean, long N, int Np, int D,
                                 ZSource zsrc, int s, double* __restrict__ F) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const long i = idx / D;
  const int d = (int)(idx % D);
  const double* Ld = L + ((long)d * Np + i) * Np;
  double a = mean[idx];
  for (long j = 0; j <= i; ++j) a += Ld[j] * draw_z(zsrc, s, j, d, D);
  F[idx] = a;
}
hipError_t fc_mean(hipStream_t st, const double* mean0, const double* X, long N, int Din, int D, int mean_kind,
                   const double* meanW, const double* meanb, double* mean) {
  hipLaunchKernelGGL(fc_mean_kernel, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, st, mean0, X, N, Din, D, mean_kind,
                     meanW, meanb, mean);
  LAUNCH_CHECK();
}
hipError_t fc_export_var(hipStream_t st, const double* V, long N, int Np, int D, double* var_out) {
  hipLaunchKernelGGL(fc_export_var_kernel, dim3((unsigned)((N * N * D + 255) / 256)), dim3(256), 0, st, V, N, Np, D, var_out);
  LAUNCH_CHECK();
}
hipError_t fc_sample(hipStream_t st, const double* L, const double* mean, long N, int Np, int D, ZSource zsrc, int s,
                     double* F) {
  hipLaunchKernelGGL(fc_sample_kernel, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, st, L, mean, N, Np, D, zsrc, s, F);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- unit-level helpers
__global__ void normals_kernel(ZSource z, int S, long N, int D, double* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)S * N * D) return;
  const int d = (int)(idx % D);
  const long r = idx / D;
  out[idx] = draw_z(z, (int)(r / N), r % N, d, D);
}
hipError_t launch_normals(hipStream_t st, ZSource z, int S, long N, int D, double* out) {
  const long n = (long)S * N * D;
  hipLaunchKernelGGL(normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, S, N, D, out);
  LAUNCH_CHECK();
}

// issue-rate micro-benchmark of v_mfma_f64_16x16x4_f64: 8 independent accumulators per wave
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, double* __restrict__ sink) {
  d4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    a += 1e-9;
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[(long)blockIdx.x * 256 + threadIdx.x] = s;
}
hipError_t launch_mfma_peak(hipStream_t st, int blocks, int iters, double* sink) {
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, iters, sink);
  LAUNCH_CHECK();
}

}  // namespace dgp
