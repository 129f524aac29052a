// libdgp_hip.so: exact GP regression entry points of include/dgp_abi.h (dgp_gpr_*), on the machinery of dgp_ctx.h.
#include "dgp_ctx.h"

extern "C" {
// ----------------------------------------------------------------------------------- exact GP regression
// gpflow.models.GPR as SO_BO builds it for num_layers == 0 (SO_BO.py:187-200): the same kernel-matrix, Cholesky and
// GEMM machinery as an SVGP layer whose inducing inputs are the N training inputs.  Stateless: every call uploads
// (X, Y) and the hyper-parameters; N <= 1024.
}  // extern "C"

namespace {
struct GprFit {
  int Np = 0;
  double *Xd = nullptr, *Yp = nullptr, *pd = nullptr, *K0 = nullptr, *E0 = nullptr, *L = nullptr, *Linv = nullptr, *tmp = nullptr,
         *A = nullptr;
  ~GprFit() {
    for (double** p : {&Xd, &Yp, &pd, &K0, &E0, &L, &Linv, &tmp, &A}) dev_free(*p);
  }
};

// K0 = K(X,X) + 1e-6 I (shared kernel, jitter removed again below), L = chol(K(X,X) + noise I), Linv, A = L^-1 Y
int gpr_fit(dgp_ctx* ctx, GprFit& f, int kind, const double* X, const double* Y, long N, int D, int Dy, double variance,
            const double* ls, double noise) {
  if (!X || !Y || !ls || N <= 0 || N > 4096 || D <= 0 || D > 64 || Dy <= 0 || kind < DGP_KERNEL_RBF || kind > DGP_KERNEL_MATERN52)
    return fail(ctx, DGP_ERR_INVALID, "gpr: bad arguments (1 <= N <= 4096, 1 <= D <= 64)");
  HIPCHK(hipSetDevice(ctx->device));
  const int Np = (int)round_up(N, 64);
  const long NN = (long)Np * Np;
  f.Np = Np;
  RET(dev_alloc(ctx, &f.Xd, (size_t)N * D)); RET(dev_alloc(ctx, &f.Yp, (size_t)Np * Dy)); RET(dev_alloc(ctx, &f.pd, (size_t)1 + D));
  RET(dev_alloc(ctx, &f.K0, (size_t)NN)); RET(dev_alloc(ctx, &f.L, (size_t)NN)); RET(dev_alloc(ctx, &f.Linv, (size_t)NN));
  RET(dev_alloc(ctx, &f.tmp, (size_t)NN)); RET(dev_alloc(ctx, &f.A, (size_t)Np * Dy));
  if (kind != DGP_KERNEL_RBF) RET(dev_alloc(ctx, &f.E0, (size_t)NN));
  std::vector<double> hp(1 + D);
  hp[0] = variance;
  for (int j = 0; j < D; ++j) hp[1 + j] = ls[j];
  HIPCHK(hipMemcpyAsync(f.Xd, X, (size_t)N * D * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemsetAsync(f.Yp, 0, (size_t)Np * Dy * 8, ctx->st));
  HIPCHK(hipMemcpyAsync(f.Yp, Y, (size_t)N * Dy * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(f.pd, hp.data(), hp.size() * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));                       // hp is a local
  HIPCHK(rbf_kuu(ctx->st, kind, f.Xd, f.pd, f.pd + 1, (int)N, Np, D, f.K0, f.E0));
  HIPCHK(copy_mat(ctx->st, f.K0, f.L, NN));
  HIPCHK(add_diag(ctx->st, f.L, (int)N, Np, noise - kJitter));
  HIPCHK(potrf_inv(ctx->st, f.L, f.Linv, f.tmp, Np, 1, ctx->info));
  RET(G(ctx, 2, GEMM_NN, Np, Dy, Np, f.Linv, Np, f.Yp, Dy, f.A, Dy, 1.0, 0));
  return DGP_OK;
}
}  // namespace

extern "C" {

int dgp_gpr_lml(dgp_ctx* ctx, int32_t kernel_kind, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy,
                double variance, const double* lengthscales, double noise_variance, double* lml_out, double* grad_out) {
  if (!ctx || !lml_out) return DGP_ERR_INVALID;
  GprFit f;
  RET(gpr_fit(ctx, f, kernel_kind, X, Y, N, D, Dy, variance, lengthscales, noise_variance));
  const int Np = f.Np;
  const long NN = (long)Np * Np;
  double *alpha = nullptr, *Kinv = nullptr, *S = nullptr, *gz = nullptr, *out = nullptr;
  struct Free { std::vector<double**> ps; ~Free() { for (auto p : ps) dev_free(*p); } } guard;
  auto get = [&](double** p, long n) -> int { guard.ps.push_back(p); return dev_alloc(ctx, p, (size_t)n); };
  RET(get(&alpha, (long)Np * Dy)); RET(get(&Kinv, NN)); RET(get(&S, NN)); RET(get(&gz, N * D + D + 1)); RET(get(&out, 2));
  RET(G(ctx, 2, GEMM_TN, Np, Dy, Np, f.Linv, Np, f.A, Dy, alpha, Dy, 1.0, 0));            // alpha = (K + s2 I)^-1 Y
  RET(G(ctx, 2, GEMM_TN, Np, Np, Np, f.Linv, Np, f.Linv, Np, Kinv, Np, 1.0, 0));
  RET(G(ctx, 2, GEMM_NT, Np, Np, Dy, alpha, Dy, alpha, Dy, S, Np, 1.0, 0));
  HIPCHK(gpr_dk(ctx->st, S, Kinv, Dy, Np));                                                 // d lml / dK
  HIPCHK(hipMemsetAsync(gz, 0, (size_t)(N * D + D + 1) * 8, ctx->st));
  double *dls = gz + N * D, *dvar = dls + D;
  HIPCHK(rbf_kuu_bwd(ctx->st, S, f.K0, f.E0, f.Xd, f.pd, f.pd + 1, (int)N, Np, D, gz, dls, dvar));
  HIPCHK(gpr_scalars(ctx->st, f.A, f.L, S, (int)N, Np, Dy, out));
  double h[2];
  std::vector<double> hg(D + 1);
  HIPCHK(hipMemcpyAsync(h, out, 16, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(hg.data(), dls, (size_t)(D + 1) * 8, hipMemcpyDeviceToHost, ctx->st));
  RET(check_flags(ctx));
  *lml_out = h[0];
  if (grad_out) {
    grad_out[0] = hg[D];
    for (int j = 0; j < D; ++j) grad_out[1 + j] = hg[j];
    grad_out[1 + D] = h[1];
  }
  return DGP_OK;
}

int dgp_gpr_predict(dgp_ctx* ctx, int32_t kernel_kind, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy,
                    double variance, const double* lengthscales, double noise_variance, const double* Xnew, int64_t Nn,
                    int32_t add_noise, double* mean_out, double* var_out) {
  if (!ctx || !Xnew || Nn <= 0 || !mean_out || !var_out) return DGP_ERR_INVALID;
  GprFit f;
  RET(gpr_fit(ctx, f, kernel_kind, X, Y, N, D, Dy, variance, lengthscales, noise_variance));
  const int Np = f.Np, nplane = Np / 32;
  const long Pm = pad_rows(Nn);
  double *Xn = nullptr, *Kt = nullptr, *Ct = nullptr, *cnp = nullptr, *mean = nullptr, *var = nullptr;
  struct Free { std::vector<double**> ps; ~Free() { for (auto p : ps) dev_free(*p); } } guard;
  auto get = [&](double** p, long n) -> int { guard.ps.push_back(p); return dev_alloc(ctx, p, (size_t)n); };
  RET(get(&Xn, Nn * D)); RET(get(&Kt, Pm * Np)); RET(get(&Ct, Pm * Np)); RET(get(&cnp, Pm * nplane)); RET(get(&mean, Pm * Dy));
  RET(get(&var, Nn * Dy));
  HIPCHK(hipMemcpyAsync(Xn, Xnew, (size_t)Nn * D * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemsetAsync(Kt, 0, (size_t)Pm * Np * 8, ctx->st));
  HIPCHK(rbf_kuf(ctx->st, kernel_kind, Xn, Nn, 0, f.Xd, f.pd, f.pd + 1, (int)N, Np, D, Kt, nullptr));
  {  // c = L^-1 k(X, x*) and |c|^2, exactly the first product of an SVGP layer
    GemmArgs a = mk(Pm, Np, Np, Kt, Np, f.Linv, Np, Ct, Np);
    a.tri = TRI_B_UPPER; a.triblk = Np; a.epi = 2; a.rowsq = cnp; a.rowsq_ld = Pm;
    RET(GX(ctx, 0, GEMM_NT, a));
  }
  RET(GX(ctx, 0, GEMM_NN, mk(Pm, Dy, Np, Ct, Np, f.A, Dy, mean, Dy)));
  HIPCHK(gpr_predict_var(ctx->st, cnp, nplane, Pm, Nn, Dy, variance, add_noise ? noise_variance : 0.0, var));
  HIPCHK(hipMemcpyAsync(mean_out, mean, (size_t)Nn * Dy * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(var_out, var, (size_t)Nn * Dy * 8, hipMemcpyDeviceToHost, ctx->st));
  return check_flags(ctx);
}

int dgp_gpr_predict_vjp(dgp_ctx* ctx, int32_t kernel_kind, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy,
                        double variance, const double* lengthscales, double noise_variance, const double* Xnew, int64_t Nn,
                        const double* mean_bar, const double* var_bar, double* xbar_out) {
  if (!ctx || !Xnew || Nn <= 0 || !mean_bar || !var_bar || !xbar_out) return DGP_ERR_INVALID;
  GprFit f;
  RET(gpr_fit(ctx, f, kernel_kind, X, Y, N, D, Dy, variance, lengthscales, noise_variance));
  const int Np = f.Np, w1 = D + 1;
  const long Pm = pad_rows(Nn);
  double *Xn = nullptr, *Kt = nullptr, *Et = nullptr, *Ct = nullptr, *Cb = nullptr, *Kb = nullptr, *Gt = nullptr, *mb = nullptr,
         *vb = nullptr, *Z1 = nullptr, *X1 = nullptr, *R1 = nullptr, *xb = nullptr;
  struct Free { std::vector<double**> ps; ~Free() { for (auto p : ps) dev_free(*p); } } guard;
  auto get = [&](double** p, long n) -> int { guard.ps.push_back(p); return dev_alloc(ctx, p, (size_t)n); };
  RET(get(&Xn, Nn * D)); RET(get(&Kt, Pm * Np)); RET(get(&Ct, Pm * Np)); RET(get(&Cb, Pm * Np)); RET(get(&Kb, Pm * Np));
  RET(get(&Gt, Pm * Np)); RET(get(&mb, Pm * Dy)); RET(get(&vb, Pm * Dy)); RET(get(&Z1, (long)Np * w1)); RET(get(&X1, Pm * w1));
  RET(get(&R1, Pm * w1)); RET(get(&xb, Nn * D));
  if (kernel_kind != DGP_KERNEL_RBF) RET(get(&Et, Pm * Np));
  HIPCHK(hipMemcpyAsync(Xn, Xnew, (size_t)Nn * D * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemsetAsync(mb, 0, (size_t)Pm * Dy * 8, ctx->st));
  HIPCHK(hipMemsetAsync(vb, 0, (size_t)Pm * Dy * 8, ctx->st));
  HIPCHK(hipMemcpyAsync(mb, mean_bar, (size_t)Nn * Dy * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(vb, var_bar, (size_t)Nn * Dy * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemsetAsync(Kt, 0, (size_t)Pm * Np * 8, ctx->st));
  if (Et) HIPCHK(hipMemsetAsync(Et, 0, (size_t)Pm * Np * 8, ctx->st));
  HIPCHK(rbf_kuf(ctx->st, kernel_kind, Xn, Nn, 0, f.Xd, f.pd, f.pd + 1, (int)N, Np, D, Kt, Et));
  {
    GemmArgs a = mk(Pm, Np, Np, Kt, Np, f.Linv, Np, Ct, Np);
    a.tri = TRI_B_UPPER; a.triblk = Np;
    RET(GX(ctx, 0, GEMM_NT, a));
  }
  HIPCHK(gpr_cbar(ctx->st, mb, vb, f.A, Ct, Pm, Np, Dy, Cb));
  {  // dK = dC L^-1, g = dK .* e  (the RBF backward of an SVGP layer, with the training inputs as inducing inputs)
    GemmArgs a = mk(Pm, Np, Np, Cb, Np, f.Linv, Np, Kb, Np);
    a.tri = TRI_B_LOWER; a.triblk = Np; a.emul = Et ? Et : Kt; a.C2 = Gt;
    RET(GX(ctx, 0, GEMM_NN, a));
  }
  HIPCHK(make_z1(ctx->st, f.Xd, (int)N, Np, D, Z1));
  HIPCHK(make_x1(ctx->st, Xn, 0, Nn, D, X1));
  RET(GX(ctx, 0, GEMM_NN, mk(Pm, w1, Np, Gt, Np, Z1, w1, R1, w1)));
  HIPCHK(xbar_finish(ctx->st, R1, X1, Nn, f.pd + 1, D, Dy, 0, nullptr, mb, 1, xb, nullptr));
  HIPCHK(hipMemcpyAsync(xbar_out, xb, (size_t)Nn * D * 8, hipMemcpyDeviceToHost, ctx->st));
  return check_flags(ctx);
}


}  // extern "C"
