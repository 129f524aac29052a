// Internal declarations shared by the HIP translation units of libdgp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f64.h"
#include "gemm_wide.h"

namespace dgp {

constexpr double kJitter = 1e-6;        // gpflow.default_jitter()  (layers.py:222,230; utils.py:41)

// Stationary kernels as functions of r2 = |(x - z)/l|^2 (kind = dgp_kernel_kind): value k and e = -2 dk/d(r2), the
// factor every input/inducing-point/lengthscale gradient carries (for the squared exponential e = k).
//   SquaredExponential  k = s2 exp(-r2/2)
//   Matern32            k = s2 (1 + sqrt3 r) exp(-sqrt3 r),             r = sqrt(max(r2, 1e-36))   (gpflow K_r)
//   Matern52            k = s2 (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)
// Below the clamp the gradient of r with respect to r2 is zero, as in the reference's autodiff.
// exp(x) for x <= 0 (every kernel exponent here is -r2/2 or -sqrt(c) r): Cody-Waite reduction by ln 2, degree-13 Taylor
// polynomial on |r| <= 0.347 (truncation 4e-18), v_ldexp_f64 for the scaling - within 1 ulp of libm's exp over [-746, 0]
// (checked against glibc on 2e7 arguments) at about half of its instructions: the Kuf assembly is exp-bound, not
// HBM-bound, with the library routine (0.78 ms per 10^6 x 256 values against 0.4 ms for writing them).
__device__ __forceinline__ double exp_nonpos(double x) {
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return x < -746.0 ? 0.0 : ldexp(p, (int)n);
}

__device__ __forceinline__ void stationary_k(int kind, double s2, double r2, double& k, double& e) {
  if (kind == 0) { k = s2 * exp_nonpos(-0.5 * r2); e = k; return; }
  const bool clamped = r2 < 1e-36;
  const double r = sqrt(clamped ? 1e-36 : r2);
  if (kind == 1) {
    const double a = 1.7320508075688772 * r, ex = exp_nonpos(-a);
    k = s2 * (1.0 + a) * ex;
    e = clamped ? 0.0 : 3.0 * s2 * ex;
  } else {
    const double a = 2.23606797749979 * r, ex = exp_nonpos(-a);
    k = s2 * (1.0 + a + (5.0 / 3.0) * r * r) * ex;
    e = clamped ? 0.0 : (5.0 / 3.0) * s2 * (1.0 + a) * ex;
  }
}
constexpr double kLikVarLower = 1e-6;   // gpflow Gaussian variance lower bound ("Softplus + Shift")

// ---------------------------------------------------------------- small matrices (smallmat.hip)
// All matrices are row-major [Mp x Mp] with leading dimension Mp, batched with stride Mp*Mp.
hipError_t rbf_kuu(hipStream_t st, int kind, const double* Z, const double* var, const double* ls, int M, int Mp, int Din,
                   double* Kuu /* + jitter*I, identity on the padding */, double* Euu /* e factors, or null */);
// A <- chol(A) (lower, upper zeroed) and X <- A^-1 of the factor; tmp = scratch [batch x Mp x Mp]; info: device flag set on non-PD
hipError_t potrf_inv(hipStream_t st, double* A, double* X, double* tmp, int Mp, int batch, int* info,
                     int n_act = 0 /* > 0: rows / columns >= n_act of A are an identity block (the padding of M up to Mp) */);
hipError_t trinv_lower(hipStream_t st, const double* L, double* X, double* tmp, int Mp, int batch);
// Lq[d] (padded, identity padding) <- tril(q_sqrt[d]) ; qmu_p (padded rows zero) <- q_mu
hipError_t pack_q(hipStream_t st, const double* q_sqrt, const double* q_mu, int M, int Mp, int D, double* Lq,
                  double* qmu_p);
// Wcat[k][d*Mp+n] <- Lq[d][k][n]   (white case: W_d = q_sqrt_d)
hipError_t lq_to_wcat(hipStream_t st, const double* Lq, int Mp, int D, double* Wcat);
hipError_t copy_mat(hipStream_t st, const double* src, double* dst, long n);
hipError_t copy_2d(hipStream_t st, const double* src, long lds, double* dst, long ldd, int nr, int nc);   // dst[r][c] = src[r][c]
// KL of one layer (layers.py:280-308) from W, u, diag(Lq), diag(Lu); adds into *kl_out (device)
hipError_t layer_kl(hipStream_t st, const double* Wcat, const double* u, const double* Lq, const double* Lu, int M,
                    int Mp, int D, int white, double* kl_out);
// elementwise pieces of the backward chain
hipError_t wbar_total(hipStream_t st, double* dWcat, const double* Wcat, double* du, const double* u, int M, int Mp,
                      int D);                                               // dW <- tril(dW) - W ; du <- du - u
hipError_t lqbar_finish(hipStream_t st, double* dLq, const double* Lq, int M, int Mp, int D);   // tril + 1/diag
hipError_t lubar_finish(hipStream_t st, double* dLu /* in: Q (+T2) */, const double* T2_or_null, const double* Lu,
                        int M, int Mp, int D, int white);                   // -tril(Q+T2) - D diag(1/Lu)
hipError_t phi_tril_halfdiag(hipStream_t st, double* T, int Mp, int batch);
hipError_t rbf_kuu_bwd(hipStream_t st, const double* S /* unsymmetrised dKuu */, const double* Kuu,
                       const double* Euu /* null: squared exponential, e = k */, const double* Z,
                       const double* var, const double* ls, int M, int Mp, int Din, double* dZ, double* dls,
                       double* dvar, const double* white = nullptr /* White variance on Kuu's diagonal, if any */);
hipError_t unpack_q_grads(hipStream_t st, const double* dLq, const double* dqmu_p, int M, int Mp, int D,
                          double* g_q_sqrt, double* g_q_mu);
hipError_t symmetrize_axpy(hipStream_t st, const double* Gm, const double* Pinv, double two_gamma, double* out,
                           int Mp, int batch);                              // out = Pinv + two_gamma * 0.5 (G + G^T)
hipError_t natgrad_mu(hipStream_t st, const double* Sn, const double* g_qmu_p, double gamma, double* qmu_p, int M,
                      int Mp, int D);                                       // mu_d -= gamma * Sn_d g_d
// S'cat[d] <- S'cat[d] - I on the valid M x M block; Gs <- symmetric completion of the lower-triangular G (batched)
// Z1 = [Z | 1] padded to Mp rows; and the Kuf part of dZ / dls / dvar from GX = g^T [X | 1] and sum_p x^2 rs
hipError_t make_z1(hipStream_t st, const double* Z, int M, int Mp, int Din, double* Z1);
hipError_t rbf_kuf_bwd_finish(hipStream_t st, const double* GX, const double* x2rs, const double* vsum, const double* Z,
                              const double* var, const double* ls, int M, int Din, double* dZ, double* dls, double* dvar,
                              const double* kdot /* null: sum k.dK = column sums of g (e = k); else that sum */);
// out[0] = sum_{i >= j} Lu[i][j] Q[i][j] = sum_p dk_p . k_p  (k = Lu c, Q = sum_p dk_p c_p^T): the variance gradient of
// the Kuf path for kernels whose e factor differs from k
hipError_t lower_dot(hipStream_t st, const double* Lu, const double* Q, int M, int Mp, double* out);
// exact GP regression (gpflow GPR, SO_BO.py:187-200): pieces beyond the shared Kuu / Cholesky / GEMM machinery
hipError_t add_diag(hipStream_t st, double* A, int M, int Mp, double v);
hipError_t gpr_scalars(hipStream_t st, const double* A /* L^-1 Y [N x Dy] */, const double* L, const double* S, int N, int Np,
                       int Dy, double* out /* [0] log marginal likelihood, [1] trace(S) */);
hipError_t gpr_dk(hipStream_t st, double* S /* alpha alpha^T -> d lml / dK */, const double* Kinv, int Dy, int Np);
hipError_t gpr_cbar(hipStream_t st, const double* mbar, const double* vbar, const double* A, const double* Ct, long P, int Np,
                    int Dy, double* Cbar);
hipError_t gpr_predict_var(hipStream_t st, const double* cnp, int nplane, long pstride, long N, int Dy, double kvar, double add,
                           double* var);
hipError_t wcat_transpose(hipStream_t st, const double* Wcat, int Mp, int D, double* WT);
hipError_t sub_identity(hipStream_t st, double* S, int M, int Mp, int batch);
hipError_t symmetrize_lower(hipStream_t st, double* G, int Mp, int batch);
hipError_t sub_scalars(hipStream_t st, const double* a, const double* b, double* out);   // out = a - b
// Q'[i][j] = sum_d (u[i][d] du[j][d] + T[d][j][i] - 2 G[d][i][j]), T[d] = dW_d W_d^T (optim.hip)
hipError_t qprime_from_sums(hipStream_t st, const double* T, const double* G, const double* u, const double* du, int Mp, int D, double* Q);
hipError_t sum_dprod(hipStream_t st, const double* T, const double* a, const double* b, int Mp, int D, double* out);   // out = a b^T + sum_d T[d] (optim.hip)
hipError_t zero_two(hipStream_t st, double* a, long na, double* b, long nb);                 // both ranges zeroed in one launch (optim.hip)
hipError_t rg_reduce(hipStream_t st, const double* slab, int nslab, int n, double* GX);       // GX[i] += sum of the slabs, in slab order (points.hip)
// gemm_gpanel.hip: R1 = g [Z | 1], GX += g^T [X | 1] for g = (Cbar Linv) .* E without storing g (Mp = 256, stationary kernels)
bool g_panel_ok(int Mp, int w1, long P);
hipError_t g_panel(hipStream_t st, const double* Cbar, const double* Linv, const double* E, long P, const double* Z1,
                   const double* X1, int w1, double* R1, double* GX, double* ws, int cu_count, int reserve_cus);
hipError_t acc_xfer(hipStream_t st, const long* tab, int nseg, double* square, double* packed, bool unpack);   // optim.hip: pack / unpack the transport form of the partial sums
hipError_t finish_tail_launch(hipStream_t st, const double* acc, double* scal, double* grad_last);   // grad_last = acc[1]; scal[1] = acc[0] - scal[0]
hipError_t store_q(hipStream_t st, const double* Lq, const double* qmu_p, int M, int Mp, int D, double* q_sqrt,
                   double* q_mu);

// ---------------------------------------------------------------- per-point kernels (points.hip)
struct ZSource {           // where the N(0,1) draws of one layer come from
  const double* zs;        // injected [S, Ntot, D] (device) or nullptr -> Philox
  uint64_t seed;
  const uint64_t* seed_dev = nullptr;   // when set, the seed is read from device memory (captured-graph training loop:
                                        // the launch arguments of a replayed graph cannot change)
  int layer;
  long n_global0;          // global index of local point 0 of the data set (Philox counter)
  long Ntot;               // row count of the injected array
};

hipError_t rbf_kuf(hipStream_t st, int kind, const double* Xin, long P, long x_row0, const double* Z, const double* var,
                   const double* ls, int M, int Mp, int Din, double* Kt, double* Et /* e factors, or null */);
// var/mean/sample stage of a layer (layers.py:249-278 + utils.py:41) from the GEMM epilogue partials.
// `dedup`: the P rows are the Nc data points of the first layer (identical for every sample s); F is always
// written for all S*Nc rows.
hipError_t finalize_layer(hipStream_t st, const double* cnp, const double* tnp, int nplane /* of cnp */, int nplane_t /* per d, of tnp */,
                          long pstride /* rows per partial plane */, const double* mean0,
                          const double* Xin, long x_row0, long P, long Nc, int S, int dedup, int Din, int D,
                          const double* kvar, int mean_kind, const double* meanW, const double* meanb, ZSource zsrc,
                          long n_chunk0, double* mean, double* var, double* F, int kernel_kind = 0,
                          const double* white = nullptr);   // kvar: kern.variance, or the 7 parameters of DGP_KERNEL_MF
// Gaussian variational expectations (gpflow Gaussian, via utils.py:89-93) + seeds of the backward pass
hipError_t gauss_lik(hipStream_t st, const double* mean, const double* var, const double* Y, long y_row0, long Nc,
                     int S, int dedup, int Dy, const double* lik_var, double* acc_elbo, double* acc_dlik,
                     double* mbar, double* vbar, double* acc_dkvar, double scale = 1.0);
// fold dF into (mbar, vbar) of the producing layer; sums over s when dedup
hipError_t fold_sample_grad(hipStream_t st, const double* Fbar, const double* var, long Nc, int S, int dedup, int D,
                            ZSource zsrc, long n_chunk0, double* mbar, double* vbar, double* acc_dkvar);
// seeds of a vector-Jacobian product through the last layer: cotangents of (F, Fmean, Fvar), each [S, Ntot, D] or null
hipError_t vjp_seed(hipStream_t st, const double* fbar, const double* meanbar, const double* varbar, const double* var,
                    long Nc, int S, int dedup, int D, ZSource zsrc, long Ntot, long n_chunk0, double* mbar, double* vbar,
                    double* acc_dkvar = nullptr);
hipError_t cbar_fix(hipStream_t st, double* Cbar, const double* mbar, const double* u, long P, int Mp, int D);
// fused per-point kernels of a layer with Mp = 64 (small_layer.hip): forward (Kuf, c, t_d, mean0, row norms), backward per
// point (dC, g, g^T [Z | 1], [X | 1]) and the backward's sums over the points (G_d, Q', du, g^T [X | 1])
bool small_layer_ok(int kind, int Mp, int D, int Din);
hipError_t small_layer_fwd(hipStream_t st, int kind, const double* Xin, long P, long x_row0, int Din, const double* Z,
                           const double* var, const double* ls, int M, int D, const double* LinvT, const double* Wcat,
                           const double* u, double* Kt, double* Et, double* Ct, double* Tt, double* cn, double* tn,
                           long pstride, double* mean0);
hipError_t small_layer_bwd(hipStream_t st, const double* Xin, long P, long x_row0, int Din, int D, const double* Linv,
                           const double* Scat, const double* u, const double* Z1, const double* Ct, const double* Tt,
                           const double* Ek, const double* vbar, const double* mbar, double* Cbar, double* Gt, double* R1,
                           double* X1);
hipError_t small_layer_sums(hipStream_t st, long P, int Din, int D, const double* Ct, const double* Cbar, const double* Gt,
                            const double* vbar, const double* mbar, const double* X1, double* accG, double* accQ, double* accdu,
                            double* accGX, double* ws /* >= small_layer_sums_ws_doubles() */);
long small_layer_sums_ws_doubles();
// one natural-gradient step of a layer's (q_mu, q_sqrt) in one launch (one workgroup per output), Mp = 64
hipError_t small_natgrad(hipStream_t st, double* Lq, const double* dLq, const double* dqmu_p, double* qmu_p, double* w0, double* w1,
                         double* w2, double* q_sqrt, double* q_mu, double gamma, int M, int D, int* info);
// the layer's gradient finish (finish_layer's chain) in one launch, Mp = 64, stationary kernels without a White term
hipError_t small_finish(hipStream_t st, const double* Gd, const double* du, const double* Qp, const double* GX, const double* x2,
                        const double* vsum, const double* Wcat, const double* u, const double* Linv, const double* Lu,
                        const double* Lq, const double* Kuu, const double* Euu, const double* Z, const double* var,
                        const double* ls, int M, int D, int Din, int white, double* dLq, double* dqmu_p, double* gZ, double* gls,
                        double* gvar, double* gqsqrt, double* gqmu, double* kdot);
// the layer's small-matrix preparation (prep_layer's chain) in one launch, Mp = 64
hipError_t small_prep(hipStream_t st, const double* q_sqrt, const double* q_mu, const double* Z, const double* var, const double* ls,
                      int M, int D, int Din, int kind, int white, int train, double* Lq, double* qmu_p, double* Kuu, double* Euu,
                      double* Lu, double* Linv, double* LinvT, double* Wcat, double* u, double* Scat /* null: not wanted */,
                      double* Z1 /* null: not wanted */, double* kl_out, int* info);
hipError_t make_x1(hipStream_t st, const double* Xin, long x_row0, long P, int Din, double* X1);     // [X | 1]
// R1 = g [Z | 1] and (GX != nullptr) GX += g^T [X | 1] in one pass over g: Mp = 256, w1 = D_in + 1 <= 9 (points.hip)
bool rbf_bwd_contract_ok(int Mp, int w1, long P);
long rbf_bwd_contract_ws_doubles();
hipError_t rbf_bwd_contract(hipStream_t st, const double* G, long P, const double* Z1, const double* X1, int w1, double* R1,
                            double* GX, double* ws);
hipError_t xbar_finish(hipStream_t st, const double* R1, const double* X1, long P, const double* ls, int Din, int D,
                       int mean_kind, const double* meanW, const double* mbar, int want_xbar, double* xbar,
                       double* acc_x2rs);
hipError_t expand_rows(hipStream_t st, const double* src, long Nc, int S, int D, int dedup, double* dst, long Ntot,
                       long n0);
hipError_t lik_predict_var(hipStream_t st, double* var, long n, const double* lik_var);
// full-covariance pieces (one sample at a time, small N): layers.py:77-80,265-268; utils.py:43-51
hipError_t fc_mean(hipStream_t st, const double* mean0, const double* X, long N, int Din, int D, int mean_kind,
                   const double* meanW, const double* meanb, double* mean);
hipError_t fc_export_var(hipStream_t st, const double* V /* [D][Np][Np], jitter on the diagonal */, long N, int Np, int D,
                         double* var_out /* [N][N][D] */);
hipError_t fc_sample(hipStream_t st, const double* L /* [D][Np][Np] */, const double* mean, long N, int Np, int D,
                     ZSource zsrc, int s, double* F);
hipError_t launch_normals(hipStream_t st, ZSource z, int S, long N, int D, double* out);
hipError_t launch_mfma_peak(hipStream_t st, int blocks, int iters, double* sink);

// ---------------------------------------------------------------- multi-fidelity kernel, White variance (mfkern.hip)
// hyp = [corr.variance, corr.lengthscales, prev.variance, prev.lengthscales, linear.variance, in.variance, in.lengthscales]
hipError_t mf_kuu(hipStream_t st, const double* Z, const double* hyp, int M, int Mp, int Din, double* Kuu);
hipError_t mf_kuf(hipStream_t st, const double* Xin, long P, long x_row0, const double* Z, const double* hyp, int M, int Mp,
                  int Din, double* Kt);
hipError_t mf_kuf_bwd_x(hipStream_t st, const double* Kbar, const double* Xin, long x_row0, long P, const double* Z,
                        const double* hyp, int M, int Mp, int Din, const double* vbar, int D, double* xbar);
hipError_t mf_kuf_bwd_p(hipStream_t st, const double* Kbar, const double* Xin, long x_row0, long P, const double* Z,
                        const double* hyp, int M, int Mp, int Din, double* accZ, double* accH);
hipError_t mf_kdiag_bwd(hipStream_t st, const double* vbar, const double* Xin, long x_row0, long P, int Din, int D,
                        const double* hyp, double* accH);
hipError_t mf_collect(hipStream_t st, const double* accZ, const double* accH, int M, int Din, double* gZ, double* gH);
hipError_t mf_kuu_bwd(hipStream_t st, const double* S, const double* Z, const double* hyp, int M, int Mp, int Din, double* gZ,
                      double* gH);
hipError_t add_diag_dev(hipStream_t st, double* A, int M, int Mp, const double* v);          // A[i][i] += v[0], i < M
hipError_t white_grad(hipStream_t st, const double* S, int M, int Mp, const double* vsum, double* out);

// ---------------------------------------------------------------- optimiser kernels (optim.hip)
enum Transform : int { TR_IDENTITY = 0, TR_SOFTPLUS = 1, TR_SOFTPLUS_SHIFT = 2, TR_TRIL = 3 };
struct ParamSeg { long off, n; int transform; int trainable; int rows; };   // rows: M for TR_TRIL blocks
// lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t): computed by the caller (t_dev == nullptr, `lr` is lr_t) or by the kernel from
// the step count in device memory (captured-graph training loop)
hipError_t adam_apply(hipStream_t st, double* params, const double* grad_elbo, double* m, double* v,
                      const ParamSeg* segs_dev, int nseg, long total, double lr, double beta1, double beta2,
                      double eps, const double* t_dev = nullptr);
// device-side bookkeeping of the captured training iteration: it[0] = Philox seed (bits), it[1] = Adam step count,
// it[2] = index into the ELBO log
hipError_t iter_bump(hipStream_t st, double* it, int seed_inc, int t_inc);
hipError_t iter_log_elbo(hipStream_t st, double* it, const double* scal, double* elbo_log, int cap);

}  // namespace dgp
