// The multi-fidelity layer kernel (DGP_KERNEL_MF) and the White-kernel variance, forward and backward.
//
// Reference: MF_DGP_EM.py:341-367 builds, for every fidelity above the first,
//     k([x,f],[x',f']) = k_corr(x,x') * (k_prev(f,f') + Linear(f,f')) + k_in(x,x')        (+ White on all but the last)
// with gpflow SquaredExponential factors (one lengthscale each) on `active_dims` of the augmented input [x, f], f being
// the previous fidelity's sample (MF_DGP_EM.py:163).  The reference differentiates it by autodiff; here the pair
// function and its partial derivatives are written out once (mf_pair) and every kernel below is a reduction of it:
//   forward   Kuu, Kuf (points.hip's finalize_layer evaluates the non-constant K_diag)
//   backward  through Kuf: x-gradient (one wave per point), Z / hyper-parameter sums (lanes along the inducing points,
//             a strip of points per workgroup, atomics into the all-reduce accumulator); through Kuu; through K_diag.
// This kernel is not stationary (Linear), so the g = dK .* e contraction trick of the stationary kernels does not
// apply; these models are small (M = N per fidelity, MF_DGP_EM.py:435-447), the pair loops are direct.
#include "dgp_internal.h"

namespace dgp {

#define LAUNCH_CHECK() return hipGetLastError()

constexpr int kMfMaxD = 33;   // D_in = D_x + 1 <= 33

struct MfHyp { double vC, lC, vP, lP, vL, vI, lI; };
__device__ __forceinline__ MfHyp mf_load(const double* __restrict__ h) { return MfHyp{h[0], h[1], h[2], h[3], h[4], h[5], h[6]}; }

// K(a, b) for a = [xa, fa], b = [xb, fb] given d2 = |xa - xb|^2.
//   gx : dK/d xa_j = -(xa_j - xb_j) * gx,  dK/d xb_j = +(xa_j - xb_j) * gx
//   dfa, dfb : dK/d fa, dK/d fb
//   dh[7]    : dK/d (vC, lC, vP, lP, vL, vI, lI)   (when wanted)
template <bool HYP>
__device__ __forceinline__ double mf_pair(const MfHyp& h, double d2, double fa, double fb, double& gx, double& dfa, double& dfb,
                                          double* dh) {
  const double rc = exp(-0.5 * d2 / (h.lC * h.lC)), ri = exp(-0.5 * d2 / (h.lI * h.lI));
  const double df = fa - fb, rp = exp(-0.5 * df * df / (h.lP * h.lP));
  const double kc = h.vC * rc, kp = h.vP * rp, kl = h.vL * fa * fb, ki = h.vI * ri;
  const double s = kp + kl;
  gx = kc * s / (h.lC * h.lC) + ki / (h.lI * h.lI);
  const double t = kp * df / (h.lP * h.lP);
  dfa = kc * (-t + h.vL * fb);
  dfb = kc * (t + h.vL * fa);
  if constexpr (HYP) {
    dh[0] = rc * s;
    dh[1] = kc * s * d2 / (h.lC * h.lC * h.lC);
    dh[2] = kc * rp;
    dh[3] = kc * kp * df * df / (h.lP * h.lP * h.lP);
    dh[4] = kc * fa * fb;
    dh[5] = ri;
    dh[6] = ki * d2 / (h.lI * h.lI * h.lI);
  }
  return kc * s + ki;
}

// ---------------------------------------------------------------------------------------- forward
__global__ void mf_kuu_kernel(const double* __restrict__ Z, const double* __restrict__ hyp, int M, int Mp, int Din,
                              double* __restrict__ Kuu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Mp * Mp) return;
  const int m = (int)(idx / Mp), n = (int)(idx % Mp);
  double v;
  if (m < M && n < M) {
    const MfHyp h = mf_load(hyp);
    const int Dx = Din - 1;
    double d2 = 0.0;
    for (int j = 0; j < Dx; ++j) { const double d = Z[(long)m * Din + j] - Z[(long)n * Din + j]; d2 += d * d; }
    double gx, a, b;
    v = mf_pair<false>(h, d2, Z[(long)m * Din + Dx], Z[(long)n * Din + Dx], gx, a, b, nullptr) + (m == n ? kJitter : 0.0);
  } else {
    v = (m == n) ? 1.0 : 0.0;
  }
  Kuu[idx] = v;
}
hipError_t mf_kuu(hipStream_t st, const double* Z, const double* hyp, int M, int Mp, int Din, double* Kuu) {
  const long n = (long)Mp * Mp;
  hipLaunchKernelGGL(mf_kuu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Z, hyp, M, Mp, Din, Kuu);
  LAUNCH_CHECK();
}

__global__ void add_diag_dev_kernel(double* __restrict__ A, int M, int Mp, const double* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) A[(long)i * Mp + i] += v[0];
}
hipError_t add_diag_dev(hipStream_t st, double* A, int M, int Mp, const double* v) {
  hipLaunchKernelGGL(add_diag_dev_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, A, M, Mp, v);
  LAUNCH_CHECK();
}

// Kt[p][m] = K(z_m, x_p); lanes along m (coalesced store), the point's row is a uniform (scalar) load
__global__ __launch_bounds__(256) void mf_kuf_kernel(const double* __restrict__ Xin, long P, long x_row0,
                                                     const double* __restrict__ Z, const double* __restrict__ hyp, int M,
                                                     int Mp, int Din, double* __restrict__ Kt, int pts_per_block) {
  const int m = blockIdx.y * 256 + threadIdx.x;
  const MfHyp h = mf_load(hyp);
  const int Dx = Din - 1;
  double z[kMfMaxD];
  for (int j = 0; j < Din; ++j) z[j] = (m < M) ? Z[(long)m * Din + j] : 0.0;
  const long p0 = (long)blockIdx.x * pts_per_block, p1 = min(P, p0 + pts_per_block);
  for (long p = p0; p < p1; ++p) {
    const double* __restrict__ x = Xin + (x_row0 + p) * Din;
    double d2 = 0.0;
    for (int j = 0; j < Dx; ++j) { const double d = z[j] - x[j]; d2 += d * d; }
    double gx, a, b;
    const double k = mf_pair<false>(h, d2, z[Dx], x[Dx], gx, a, b, nullptr);
    if (m < Mp) Kt[p * Mp + m] = (m < M) ? k : 0.0;
  }
}
hipError_t mf_kuf(hipStream_t st, const double* Xin, long P, long x_row0, const double* Z, const double* hyp, int M, int Mp,
                  int Din, double* Kt) {
  if (P == 0) return hipSuccess;
  int ppb = 64;
  while (ppb > 1 && (P + ppb - 1) / ppb < 2048) ppb >>= 1;
  dim3 grid((unsigned)((P + ppb - 1) / ppb), (unsigned)((Mp + 255) / 256));
  hipLaunchKernelGGL(mf_kuf_kernel, grid, dim3(256), 0, st, Xin, P, x_row0, Z, hyp, M, Mp, Din, Kt, ppb);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- backward through Kuf
// xbar[p][:] = sum_m dK[p][m] dK(z_m, x_p)/dx_p  +  (K_diag path) 2 vC vL f_p sum_d vbar[p][d]; one wave per point
__global__ __launch_bounds__(256) void mf_kuf_bwd_x_kernel(const double* __restrict__ Kbar, const double* __restrict__ Xin,
                                                           long x_row0, long P, const double* __restrict__ Z,
                                                           const double* __restrict__ hyp, int M, int Mp, int Din,
                                                           const double* __restrict__ vbar, int D,
                                                           double* __restrict__ xbar) {
  const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const int lane = threadIdx.x & 63, Dx = Din - 1;
  const MfHyp h = mf_load(hyp);
  const double* __restrict__ x = Xin + (x_row0 + p) * Din;
  double acc[kMfMaxD];
  for (int j = 0; j < Din; ++j) acc[j] = 0.0;
  for (int m = lane; m < M; m += 64) {
    const double* __restrict__ z = Z + (long)m * Din;
    double d2 = 0.0;
    for (int j = 0; j < Dx; ++j) { const double d = z[j] - x[j]; d2 += d * d; }
    double gx, dfa, dfb;
    mf_pair<false>(h, d2, z[Dx], x[Dx], gx, dfa, dfb, nullptr);
    const double w = Kbar[p * Mp + m];
    for (int j = 0; j < Dx; ++j) acc[j] += w * gx * (z[j] - x[j]);
    acc[Dx] += w * dfb;
  }
  for (int j = 0; j < Din; ++j) {
    double v = acc[j];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    acc[j] = v;
  }
  if (lane == 0) {
    double vs = 0.0;
    for (int d = 0; d < D; ++d) vs += vbar[p * D + d];
    for (int j = 0; j < Dx; ++j) xbar[p * Din + j] = acc[j];
    xbar[p * Din + Dx] = acc[Dx] + 2.0 * h.vC * h.vL * x[Dx] * vs;
  }
}
hipError_t mf_kuf_bwd_x(hipStream_t st, const double* Kbar, const double* Xin, long x_row0, long P, const double* Z,
                        const double* hyp, int M, int Mp, int Din, const double* vbar, int D, double* xbar) {
  if (P == 0) return hipSuccess;
  hipLaunchKernelGGL(mf_kuf_bwd_x_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, Kbar, Xin, x_row0, P, Z, hyp, M, Mp,
                     Din, vbar, D, xbar);
  LAUNCH_CHECK();
}

// accZ[m][:] += sum_p dK[p][m] dK(z_m, x_p)/dz_m ;  accH[:] += sum_{p,m} dK[p][m] dK/d hyp.   Lane = inducing point m
// (coalesced reads of dK rows), a strip of points per workgroup.
__global__ __launch_bounds__(64) void mf_kuf_bwd_p_kernel(const double* __restrict__ Kbar, const double* __restrict__ Xin,
                                                          long x_row0, long P, const double* __restrict__ Z,
                                                          const double* __restrict__ hyp, int M, int Mp, int Din,
                                                          long strip, double* __restrict__ accZ, double* __restrict__ accH) {
  const int m = blockIdx.x * 64 + threadIdx.x, Dx = Din - 1;
  const MfHyp h = mf_load(hyp);
  const bool live = m < M;
  double z[kMfMaxD], za[kMfMaxD], ha[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < Din; ++j) { z[j] = live ? Z[(long)m * Din + j] : 0.0; za[j] = 0.0; }
  const long p0 = (long)blockIdx.y * strip, p1 = min(P, p0 + strip);
  for (long p = p0; p < p1; ++p) {
    const double* __restrict__ x = Xin + (x_row0 + p) * Din;
    double d2 = 0.0;
    for (int j = 0; j < Dx; ++j) { const double d = z[j] - x[j]; d2 += d * d; }
    double gx, dfa, dfb, dh[7];
    mf_pair<true>(h, d2, z[Dx], x[Dx], gx, dfa, dfb, dh);
    const double w = live ? Kbar[p * Mp + m] : 0.0;
    for (int j = 0; j < Dx; ++j) za[j] -= w * gx * (z[j] - x[j]);
    za[Dx] += w * dfa;
#pragma unroll
    for (int k = 0; k < 7; ++k) ha[k] += w * dh[k];
  }
  if (live)
    for (int j = 0; j < Din; ++j) unsafeAtomicAdd(accZ + (long)m * Din + j, za[j]);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double v = ha[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (threadIdx.x == 0) unsafeAtomicAdd(accH + k, v);
  }
}
hipError_t mf_kuf_bwd_p(hipStream_t st, const double* Kbar, const double* Xin, long x_row0, long P, const double* Z,
                        const double* hyp, int M, int Mp, int Din, double* accZ, double* accH) {
  if (P == 0) return hipSuccess;
  long strips = (2048 + (M + 63) / 64 - 1) / ((M + 63) / 64);       // ~2048 workgroups in all
  if (strips > P) strips = P;
  const long strip = (P + strips - 1) / strips;
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((P + strip - 1) / strip));
  hipLaunchKernelGGL(mf_kuf_bwd_p_kernel, grid, dim3(64), 0, st, Kbar, Xin, x_row0, P, Z, hyp, M, Mp, Din, strip, accZ, accH);
  LAUNCH_CHECK();
}

// K_diag_p = vC (vP + vL f_p^2) + vI (+ white): hyper-parameter sums of sum_d vbar[p][d] * dK_diag/d hyp
__global__ __launch_bounds__(256) void mf_kdiag_bwd_kernel(const double* __restrict__ vbar, const double* __restrict__ Xin,
                                                           long x_row0, long P, int Din, int D,
                                                           const double* __restrict__ hyp, double* __restrict__ accH) {
  __shared__ double sh[8];
  double s0 = 0.0, s2 = 0.0;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    double vs = 0.0;
    for (int d = 0; d < D; ++d) vs += vbar[p * D + d];
    const double f = Xin[(x_row0 + p) * Din + Din - 1];
    s0 += vs;
    s2 += vs * f * f;
  }
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o); s2 += __shfl_down(s2, o); }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { sh[wv] = s0; sh[4 + wv] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s0 = sh[0] + sh[1] + sh[2] + sh[3];
    s2 = sh[4] + sh[5] + sh[6] + sh[7];
    const MfHyp h = mf_load(hyp);
    unsafeAtomicAdd(accH + 0, h.vP * s0 + h.vL * s2);
    unsafeAtomicAdd(accH + 2, h.vC * s0);
    unsafeAtomicAdd(accH + 4, h.vC * s2);
    unsafeAtomicAdd(accH + 5, s0);
  }
}
hipError_t mf_kdiag_bwd(hipStream_t st, const double* vbar, const double* Xin, long x_row0, long P, int Din, int D,
                        const double* hyp, double* accH) {
  if (P == 0) return hipSuccess;
  long blocks = (P + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mf_kdiag_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, vbar, Xin, x_row0, P, Din, D, hyp, accH);
  LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------- gradient assembly
// g[Z] = accZ, g[hyp] = accH  (the Kuf + K_diag parts; mf_kuu_bwd adds the Kuu part)
__global__ void mf_collect_kernel(const double* __restrict__ accZ, const double* __restrict__ accH, long nz,
                                  double* __restrict__ gZ, double* __restrict__ gH) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nz) gZ[i] = accZ[i];
  if (i < 7) gH[i] = accH[i];
}
hipError_t mf_collect(hipStream_t st, const double* accZ, const double* accH, int M, int Din, double* gZ, double* gH) {
  const long nz = (long)M * Din;
  hipLaunchKernelGGL(mf_collect_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, accZ, accH, nz, gZ, gH);
  LAUNCH_CHECK();
}

// through Kuu = K(Z,Z) + jitter I with dKuu = sym(S): one wave per inducing point m, lanes over n
__global__ __launch_bounds__(64) void mf_kuu_bwd_kernel(const double* __restrict__ S, const double* __restrict__ Z,
                                                        const double* __restrict__ hyp, int M, int Mp, int Din,
                                                        double* __restrict__ gZ, double* __restrict__ gH) {
  const int m = blockIdx.x, lane = threadIdx.x, Dx = Din - 1;
  const MfHyp h = mf_load(hyp);
  double z[kMfMaxD], za[kMfMaxD], ha[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < Din; ++j) { z[j] = Z[(long)m * Din + j]; za[j] = 0.0; }
  for (int n = lane; n < M; n += 64) {
    const double* __restrict__ zn = Z + (long)n * Din;
    double d2 = 0.0;
    for (int j = 0; j < Dx; ++j) { const double d = z[j] - zn[j]; d2 += d * d; }
    double gx, dfa, dfb, dh[7];
    mf_pair<true>(h, d2, z[Dx], zn[Dx], gx, dfa, dfb, dh);
    const double sym = 0.5 * (S[(long)m * Mp + n] + S[(long)n * Mp + m]);
    for (int j = 0; j < Dx; ++j) za[j] -= 2.0 * sym * gx * (z[j] - zn[j]);
    za[Dx] += 2.0 * sym * dfa;
#pragma unroll
    for (int k = 0; k < 7; ++k) ha[k] += sym * dh[k];
  }
  for (int j = 0; j < Din; ++j) {
    double v = za[j];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) gZ[(long)m * Din + j] += v;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double v = ha[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) unsafeAtomicAdd(gH + k, v);
  }
}
hipError_t mf_kuu_bwd(hipStream_t st, const double* S, const double* Z, const double* hyp, int M, int Mp, int Din, double* gZ,
                      double* gH) {
  hipLaunchKernelGGL(mf_kuu_bwd_kernel, dim3(M), dim3(64), 0, st, S, Z, hyp, M, Mp, Din, gZ, gH);
  LAUNCH_CHECK();
}

// d/d white.variance = trace(dKuu) + sum over points and outputs of vbar   (White enters Kuu and K_diag only)
__global__ __launch_bounds__(256) void white_grad_kernel(const double* __restrict__ S, int M, int Mp,
                                                         const double* __restrict__ vsum, double* __restrict__ out) {
  __shared__ double sh[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < M; i += blockDim.x) a += S[(long)i * Mp + i];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sh[0] + sh[1] + sh[2] + sh[3] + vsum[0];
}
hipError_t white_grad(hipStream_t st, const double* S, int M, int Mp, const double* vsum, double* out) {
  hipLaunchKernelGGL(white_grad_kernel, dim3(1), dim3(256), 0, st, S, M, Mp, vsum, out);
  LAUNCH_CHECK();
}

}  // namespace dgp
