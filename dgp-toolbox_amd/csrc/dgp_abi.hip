// libdgp_hip.so: the C-ABI of the SVGP-layer ELBO path (include/dgp_abi.h) on top of dgp_ctx.h.
#include "dgp_ctx.h"
#include <dlfcn.h>
#include <link.h>
#include <limits.h>

struct ncclUniqueIdBytes { char internal[128]; };   // rccl.h: ncclUniqueId (passed by value to ncclCommInitRank)

// =================================================================================== C-ABI
extern "C" {

int dgp_create(int device, void* hip_stream, dgp_ctx** out) {
  if (!out) return DGP_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DGP_ERR_NO_DEVICE;
  dgp_ctx* ctx = new dgp_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return DGP_ERR_NO_DEVICE; }
  if (hip_stream) ctx->st = reinterpret_cast<hipStream_t>(hip_stream);
  else {
    if (hipStreamCreateWithFlags(&ctx->st, hipStreamNonBlocking) != hipSuccess) { delete ctx; return DGP_ERR_HIP; }
    ctx->own_stream = true;
  }
  if (hipMalloc(reinterpret_cast<void**>(&ctx->scal), 4 * sizeof(double)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&ctx->info), sizeof(int)) != hipSuccess) {
    delete ctx;
    return DGP_ERR_HIP;
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
  }
  (void)hipMemset(ctx->info, 0, sizeof(int));
  (void)hipMemset(ctx->scal, 0, 4 * sizeof(double));
  {
    const char* e = getenv("DGP_SIDE_STREAMS");
    ctx->use_side = !(e && e[0] == '0');
    const char* t = getenv("DGP_STORE_T");
    ctx->store_t = !(t && t[0] == '0');
    const char* ch = getenv("DGP_CHAIN");
    ctx->chain_on = ch && ch[0] == '1';
    const char* dg = getenv("DGP_DU_IN_GRAM");
    ctx->du_in_gram = !(dg && dg[0] == '0');
    const char* qg = getenv("DGP_Q_FROM_G");
    ctx->q_from_g = !(qg && qg[0] == '0');
    const char* bt = getenv("DGP_BLOCKED_T");
    ctx->blocked_t = !(bt && bt[0] == '0');
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < dgp_ctx::kSide && ok; ++i)
      ok = hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < dgp_ctx::kMaxEv && ok; ++i)
      ok = hipEventCreateWithFlags(&ctx->ev_prep[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_layer[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_red[i], hipEventDisableTiming) == hipSuccess;
    {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (ok && hipStreamCreateWithPriority(&ctx->bst, hipStreamNonBlocking, least) == hipSuccess &&
          hipEventCreateWithFlags(&ctx->ev_b[0], hipEventDisableTiming) == hipSuccess &&
          hipEventCreateWithFlags(&ctx->ev_b[1], hipEventDisableTiming) == hipSuccess) {
        const char* ob = getenv("DGP_OVERLAP_B");
        ctx->overlap_b = ob && ob[0] == '1';
        if (const char* r = getenv("DGP_OVERLAP_RESERVE")) ctx->overlap_reserve_wide = atoi(r);
        if (const char* r = getenv("DGP_OVERLAP_RESERVE_NARROW")) ctx->overlap_reserve_narrow = atoi(r);
        if (const char* r = getenv("DGP_OVERLAP_MIN_WORK")) ctx->overlap_min_work = atof(r);
      } else {
        ctx->overlap_b = false;
        (void)hipGetLastError();
      }
    }
    // one flag for all of them: a partial failure leaves every side path off (dgp_destroy releases what exists)
    ctx->events_ok = ok;
    if (!ok) { ctx->use_side = false; (void)hipGetLastError(); }
  }
  *out = ctx;
  return DGP_OK;
}

void dgp_destroy(dgp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->st);
  free_model(ctx);
  dev_free(ctx->X); dev_free(ctx->Y); dev_free(ctx->scal); dev_free(ctx->info); dev_free(ctx->Xnew);
  if (ctx->ws) (void)hipFree(ctx->ws);
  for (auto e : ctx->prof.ev) (void)hipEventDestroy(e);
  for (auto e : ctx->prof.marks) (void)hipEventDestroy(e);
  for (int i = 0; i < dgp_ctx::kSide; ++i) {
    if (ctx->side[i]) { (void)hipStreamSynchronize(ctx->side[i]); (void)hipStreamDestroy(ctx->side[i]); }
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->bst) { (void)hipStreamSynchronize(ctx->bst); (void)hipStreamDestroy(ctx->bst); }
  for (auto& e : ctx->ev_b) if (e) (void)hipEventDestroy(e);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  (void)dgp_comm_destroy(ctx);
  drop_graph(ctx);
  dev_free(ctx->it_dev); dev_free(ctx->elbo_log);
  for (int i = 0; i < dgp_ctx::kMaxEv; ++i) {
    if (ctx->ev_prep[i]) (void)hipEventDestroy(ctx->ev_prep[i]);
    if (ctx->ev_layer[i]) (void)hipEventDestroy(ctx->ev_layer[i]);
    if (ctx->ev_red[i]) (void)hipEventDestroy(ctx->ev_red[i]);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->st);
  delete ctx;
}

const char* dgp_last_error(const dgp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dgp_sync(dgp_ctx* ctx) {
  if (!ctx) return DGP_ERR_INVALID;
  return check_flags(ctx);
}

int dgp_device_info(dgp_ctx* ctx, char* name_out, int name_len, int* cu_count, int64_t* hbm_bytes) {
  if (!ctx) return DGP_ERR_INVALID;
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, ctx->device));
  if (name_out && name_len > 0) snprintf(name_out, name_len, "%s (%s)", p.name, p.gcnArchName);
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return DGP_OK;
}

int dgp_model_set(dgp_ctx* ctx, int n_layers, const dgp_layer_desc* layers, const double* flat_params, int64_t n_params,
                  const double* mean_params, int64_t n_mean_params) {
  if (!ctx || n_layers <= 0 || !layers || !flat_params) return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->st));
  free_model(ctx);
  long off = 0, moff = 0, aoff = 4;
  int Mpmax = 0, Dmax = 0;
  for (int l = 0; l < n_layers; ++l) {
    const dgp_layer_desc& d = layers[l];
    if (d.D_in <= 0 || d.D_in > 64 || d.D_out <= 0 || d.M <= 0 || d.M > 4096 || d.D_out > 65535 || d.kernel_kind < DGP_KERNEL_RBF || d.kernel_kind > DGP_KERNEL_MF ||
        d.mean_kind < 0 || d.mean_kind > 2 || d.kernel_white < 0 || d.kernel_white > 1)
      return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: unsupported layer description (M <= 4096, D_in <= 64)");
    if (d.kernel_kind == DGP_KERNEL_MF && (d.D_in < 2 || d.D_in > 33 || d.mean_kind != DGP_MEAN_ZERO))
      return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: the multi-fidelity kernel needs 2 <= D_in <= 33 and a zero mean function");
    if (l > 0 && d.D_in != layers[l - 1].D_out) return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: layer dims do not chain");
    if (d.mean_kind == DGP_MEAN_IDENTITY && d.D_in != d.D_out)
      return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: identity mean needs D_in == D_out");
    Layer y{};
    y.d = d;
    y.Mp = (int)round_up(d.M, 64);   // tiles of 64 columns never straddle an output block
    y.off_Z = off; off += (long)d.M * d.D_in;
    y.off_var = off; off += 1;
    y.off_ls = off; off += (d.kernel_kind == DGP_KERNEL_MF) ? 6 : d.D_in;
    if (d.kernel_white) { y.off_white = off; off += 1; }
    y.off_qmu = off; off += (long)d.M * d.D_out;
    y.off_qsqrt = off; off += (long)d.D_out * d.M * d.M;
    const long MM = (long)y.Mp * y.Mp;
    y.acc_Q = aoff; aoff += MM;
    y.acc_G = aoff; aoff += MM * d.D_out;
    y.acc_du = aoff; aoff += round_up((long)y.Mp * d.D_out, 2);
    y.acc_GX = aoff; aoff += round_up((long)y.Mp * (d.D_in + 1), 2);
    y.acc_x2 = aoff; aoff += round_up(std::max(d.D_in, 8), 2);   // DGP_KERNEL_MF keeps its 7 parameter sums here
    y.acc_dvar = aoff; aoff += 2;
    Mpmax = std::max(Mpmax, y.Mp);
    Dmax = std::max(Dmax, d.D_out);
    ctx->L.push_back(y);
    if (d.mean_kind == DGP_MEAN_LINEAR) moff += (long)d.D_in * d.D_out + d.D_out;
  }
  off += 1;   // likelihood variance
  if (off != n_params) { free_model(ctx); return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: n_params does not match the layer list"); }
  if (moff != n_mean_params || (moff > 0 && !mean_params)) {
    free_model(ctx);
    return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: n_mean_params does not match the layer list");
  }
  ctx->n_params = off;
  ctx->n_acc = aoff;
  RET(dev_alloc(ctx, &ctx->params, off));
  RET(dev_alloc(ctx, &ctx->grad, off));
  RET(dev_alloc(ctx, &ctx->adam_m, off));
  RET(dev_alloc(ctx, &ctx->adam_v, off));
  RET(dev_alloc(ctx, &ctx->acc_own, aoff));
  ctx->acc = ctx->acc_own;
  HIPCHK(hipMemcpy(ctx->params, flat_params, off * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(ctx->adam_m, 0, off * 8));
  HIPCHK(hipMemset(ctx->adam_v, 0, off * 8));
  HIPCHK(hipMemset(ctx->grad, 0, off * 8));
  ctx->adam_t = 0;
  if (moff > 0) {
    RET(dev_alloc(ctx, &ctx->mean_params, moff));
    HIPCHK(hipMemcpy(ctx->mean_params, mean_params, moff * 8, hipMemcpyHostToDevice));
  }
  moff = 0;
  ctx->segs.clear();
  for (auto& y : ctx->L) {
    const long MM = (long)y.Mp * y.Mp;
    const int D = y.d.D_out;
    RET(dev_alloc(ctx, &y.Kuu, MM)); RET(dev_alloc(ctx, &y.Lu, MM)); RET(dev_alloc(ctx, &y.Linv, MM)); RET(dev_alloc(ctx, &y.LinvT, MM));
    if (y.Mp == 256 && y.d.D_out <= 8) RET(dev_alloc(ctx, &y.alpha, (size_t)y.Mp * y.d.D_out));
    if (y.Mp == 256 && !ctx->gram_ws) RET(dev_alloc(ctx, &ctx->gram_ws, gemm_gram_ws_bytes(ctx->cu_count) / 8));
    RET(dev_alloc(ctx, &y.Lq, MM * D)); RET(dev_alloc(ctx, &y.qmu_p, (long)y.Mp * D));
    RET(dev_alloc(ctx, &y.Wcat, MM * D)); RET(dev_alloc(ctx, &y.u, (long)y.Mp * D)); RET(dev_alloc(ctx, &y.Scat, MM * D)); RET(dev_alloc(ctx, &y.Z1, (long)y.Mp * (y.d.D_in + 1)));
    if (y.d.kernel_kind == DGP_KERNEL_MATERN32 || y.d.kernel_kind == DGP_KERNEL_MATERN52) { RET(dev_alloc(ctx, &y.Euu, MM)); RET(dev_alloc(ctx, &y.kdot, 1)); }
    RET(dev_alloc(ctx, &y.dLq, MM * D)); RET(dev_alloc(ctx, &y.dqmu_p, (long)y.Mp * D));
    if (y.d.mean_kind == DGP_MEAN_LINEAR) {
      y.meanW = ctx->mean_params + moff; moff += (long)y.d.D_in * D;
      y.meanb = ctx->mean_params + moff; moff += D;
    }
    ctx->segs.push_back({y.off_Z, (long)y.d.M * y.d.D_in, TR_IDENTITY, 1, 0});
    ctx->segs.push_back({y.off_var, 1, TR_SOFTPLUS, 1, 0});
    ctx->segs.push_back({y.off_ls, (y.d.kernel_kind == DGP_KERNEL_MF) ? 6L : (long)y.d.D_in, TR_SOFTPLUS, 1, 0});
    if (y.off_white >= 0) ctx->segs.push_back({y.off_white, 1, TR_SOFTPLUS, 1, 0});
    ctx->segs.push_back({y.off_qmu, (long)y.d.M * D, TR_IDENTITY, 1, 0});
    ctx->segs.push_back({y.off_qsqrt, (long)D * y.d.M * y.d.M, TR_TRIL, 1, y.d.M});
  }
  ctx->segs.push_back({off - 1, 1, TR_SOFTPLUS_SHIFT, 1, 0});
  RET(dev_alloc(ctx, &ctx->segs_dev, ctx->segs.size()));
  const size_t each = (size_t)Mpmax * Mpmax * Dmax;
  for (auto& set : ctx->smset) for (auto& s : set) RET(dev_alloc(ctx, &s, each));
  return DGP_OK;
}

int64_t dgp_param_count(const dgp_ctx* ctx) { return ctx ? ctx->n_params : 0; }

int dgp_params_get(dgp_ctx* ctx, double* flat_out) {
  if (!ctx || !ctx->params || !flat_out) return fail(ctx, DGP_ERR_INVALID, "dgp_params_get: no model");
  HIPCHK(hipMemcpyAsync(flat_out, ctx->params, ctx->n_params * 8, hipMemcpyDeviceToHost, ctx->st));
  return check_flags(ctx);
}

int dgp_params_set(dgp_ctx* ctx, const double* flat_in) {
  if (!ctx || !ctx->params || !flat_in) return fail(ctx, DGP_ERR_INVALID, "dgp_params_set: no model");
  HIPCHK(hipMemcpyAsync(ctx->params, flat_in, ctx->n_params * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->grad_ready = false;
  ctx->prep_level = 0;
  return DGP_OK;
}

int dgp_data_set(dgp_ctx* ctx, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy, int64_t n_global_offset) {
  if (ctx) drop_graph(ctx);
  if (!ctx || !X || !Y || N <= 0 || D <= 0 || Dy <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_data_set: bad arguments");
  HIPCHK(hipStreamSynchronize(ctx->st));
  dev_free(ctx->X); dev_free(ctx->Y);
  RET(dev_alloc(ctx, &ctx->X, (size_t)N * D));
  RET(dev_alloc(ctx, &ctx->Y, (size_t)N * Dy));
  HIPCHK(hipMemcpy(ctx->X, X, (size_t)N * D * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->Y, Y, (size_t)N * Dy * 8, hipMemcpyHostToDevice));
  ctx->N = N; ctx->D = D; ctx->Dy = Dy; ctx->n_goff = n_global_offset;
  ctx->batch_lo = ctx->batch_n = 0; ctx->data_scale = 1.0;          // a new data set starts with the full-batch bound
  return DGP_OK;
}

int dgp_set_workspace_limit(dgp_ctx* ctx, int64_t bytes) {
  if (!ctx || bytes < (1 << 20)) return fail(ctx, DGP_ERR_INVALID, "dgp_set_workspace_limit: at least 1 MiB");
  ctx->ws_limit = bytes;
  return DGP_OK;
}

static int check_ready(dgp_ctx* ctx, bool need_data) {
  if (!ctx) return DGP_ERR_INVALID;
  if (ctx->L.empty()) return fail(ctx, DGP_ERR_INVALID, "no model: call dgp_model_set first");
  if (need_data) {
    if (!ctx->X) return fail(ctx, DGP_ERR_INVALID, "no data: call dgp_data_set first");
    if (ctx->D != ctx->L[0].d.D_in || ctx->Dy != ctx->L.back().d.D_out)
      return fail(ctx, DGP_ERR_INVALID, "data dimensions do not match the model");
  }
  return DGP_OK;
}

int dgp_batch_set(dgp_ctx* ctx, int64_t start, int64_t count, double scale) {
  if (!ctx || start < 0 || count < 0 || !(scale > 0.0)) return fail(ctx, DGP_ERR_INVALID, "dgp_batch_set: bad arguments");
  if (ctx->X && start + count > ctx->N) return fail(ctx, DGP_ERR_INVALID, "dgp_batch_set: window exceeds the resident points");
  ctx->batch_lo = count ? start : 0;
  ctx->batch_n = count;
  ctx->data_scale = scale;
  return DGP_OK;
}

int dgp_elbo(dgp_ctx* ctx, int32_t S, uint64_t seed, const double* const* zs, double* data_term, double* kl) {
  RET(check_ready(ctx, true));
  if (S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_elbo: S must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  if (zs) RET(upload_zs(ctx, zs, S, ctx->N));
  RET(prep(ctx));
  long Nc = 0;
  const long lo = ctx->batch_n ? ctx->batch_lo : 0, hi = ctx->batch_n ? ctx->batch_lo + ctx->batch_n : ctx->N;
  RET(ensure_ws(ctx, hi - lo, S, false, &Nc));
  Layer& last = ctx->L.back();
  const bool dedup_last = ctx->L.size() == 1;
  for (long n0 = lo; n0 < hi; n0 += Nc) {
    const long nc = std::min(Nc, hi - n0);
    RET(forward_chunk(ctx, ctx->X, ctx->N, n0, nc, S, seed, zs != nullptr, ctx->n_goff));
    ProfScope ps(ctx, 1, 0, 0);
    HIPCHK(gauss_lik(ctx->st, last.mean, last.var, ctx->Y, n0, nc, S, dedup_last ? 1 : 0, ctx->Dy,
                     P(ctx, ctx->n_params - 1), ctx->scal + 2, nullptr, nullptr, nullptr, nullptr, ctx->data_scale));
  }
  double h[4];
  HIPCHK(hipMemcpyAsync(h, ctx->scal, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->st));
  RET(check_flags(ctx));
  if (data_term) *data_term = h[2];
  if (kl) *kl = h[0];
  if (!std::isfinite(h[2]) || !std::isfinite(h[0])) return fail(ctx, DGP_ERR_NONFINITE, "ELBO is not finite");
  return DGP_OK;
}

int dgp_propagate(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed, const double* const* zs,
                  double* const* Fs, double* const* Fmeans, double* const* Fvars, int32_t add_lik_var) {
  RET(check_ready(ctx, false));
  if (!Xnew || Nn <= 0 || S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_propagate: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  const int nl = (int)ctx->L.size();
  const int Din0 = ctx->L[0].d.D_in;
  RET(grow(ctx, &ctx->Xnew, &ctx->Xnew_cap, (size_t)Nn * Din0));
  HIPCHK(hipMemcpyAsync(ctx->Xnew, Xnew, (size_t)Nn * Din0 * 8, hipMemcpyHostToDevice, ctx->st));
  if (zs) RET(upload_zs(ctx, zs, S, Nn));
  double* const* outs[3] = {Fs, Fmeans, Fvars};
  for (int k = 0; k < 3; ++k) {
    if ((int)ctx->out_dev[k].size() < nl) { ctx->out_dev[k].resize(nl, nullptr); ctx->out_cap[k].resize(nl, 0); }
    for (int l = 0; l < nl; ++l)
      if (outs[k] && outs[k][l]) RET(grow(ctx, &ctx->out_dev[k][l], &ctx->out_cap[k][l], (size_t)S * Nn * ctx->L[l].d.D_out));
  }
  RET(prep(ctx));
  long Nc = 0;
  RET(ensure_ws(ctx, Nn, S, false, &Nc));
  for (long n0 = 0; n0 < Nn; n0 += Nc) {
    const long nc = std::min(Nc, (long)Nn - n0);
    RET(forward_chunk(ctx, ctx->Xnew, Nn, n0, nc, S, seed, zs != nullptr, 0));
    for (int l = 0; l < nl; ++l) {
      Layer& y = ctx->L[l];
      const int D = y.d.D_out;
      const int dd = (l == 0) ? 1 : 0;
      ProfScope ps(ctx, 1, 0, 0);
      if (l == nl - 1 && add_lik_var)
        HIPCHK(lik_predict_var(ctx->st, y.var, (dd ? nc : (long)S * nc) * D, P(ctx, ctx->n_params - 1)));
      if (Fs && Fs[l]) HIPCHK(expand_rows(ctx->st, y.F, nc, S, D, 0, ctx->out_dev[0][l], Nn, n0));
      if (Fmeans && Fmeans[l]) HIPCHK(expand_rows(ctx->st, y.mean, nc, S, D, dd, ctx->out_dev[1][l], Nn, n0));
      if (Fvars && Fvars[l]) HIPCHK(expand_rows(ctx->st, y.var, nc, S, D, dd, ctx->out_dev[2][l], Nn, n0));
    }
  }
  for (int k = 0; k < 3; ++k)
    for (int l = 0; l < nl; ++l)
      if (outs[k] && outs[k][l])
        HIPCHK(hipMemcpyAsync(outs[k][l], ctx->out_dev[k][l], (size_t)S * Nn * ctx->L[l].d.D_out * 8,
                              hipMemcpyDeviceToHost, ctx->st));
  return check_flags(ctx);
}

int dgp_propagate_full_cov(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed,
                           const double* const* zs, double* const* Fs, double* const* Fmeans, double* const* Fvars) {
  RET(check_ready(ctx, false));
  if (!Xnew || Nn <= 0 || S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_propagate_full_cov: bad arguments");
  if (Nn > 4096) return fail(ctx, DGP_ERR_INVALID, "dgp_propagate_full_cov: at most 4096 points (N x N covariances per sample)");
  HIPCHK(hipSetDevice(ctx->device));
  const int nl = (int)ctx->L.size();
  const long N = Nn, Pm = pad_rows(N);
  const int Np = (int)round_up(N, 64);
  if (zs) RET(upload_zs(ctx, zs, S, N));
  RET(prep(ctx));
  int Dmax = 1, Mpmax = 64, Dinmax = 1;
  for (auto& y : ctx->L) { Dmax = std::max(Dmax, y.d.D_out); Mpmax = std::max(Mpmax, y.Mp); Dinmax = std::max(Dinmax, y.d.D_in); }
  // scratch for one sample of one layer (small-N path: allocated per call)
  double *Kt = nullptr, *Ct = nullptr, *T = nullptr, *mean0 = nullptr, *mean = nullptr, *Kff = nullptr, *V = nullptr,
         *Vinv = nullptr, *tmp = nullptr, *var_dev = nullptr, *Fa = nullptr, *Fb = nullptr, *X0 = nullptr;
  const long NN = (long)Np * Np;
  struct Free {
    std::vector<double**> ps;
    ~Free() { for (auto p : ps) dev_free(*p); }
  } guard;
  auto get = [&](double** p, long n) -> int { guard.ps.push_back(p); return dev_alloc(ctx, p, (size_t)n); };
  RET(get(&Kt, Pm * Mpmax)); RET(get(&Ct, Pm * Mpmax)); RET(get(&T, Pm * (long)Dmax * Mpmax));
  RET(get(&mean0, Pm * Dmax)); RET(get(&mean, N * Dmax)); RET(get(&Kff, NN)); RET(get(&V, NN * Dmax));
  RET(get(&Vinv, NN * Dmax)); RET(get(&tmp, NN * Dmax)); RET(get(&var_dev, N * N * Dmax));
  RET(get(&Fa, (long)S * N * std::max(Dmax, Dinmax))); RET(get(&Fb, (long)S * N * std::max(Dmax, Dinmax)));
  RET(get(&X0, N * ctx->L[0].d.D_in));
  HIPCHK(hipMemcpyAsync(X0, Xnew, (size_t)N * ctx->L[0].d.D_in * 8, hipMemcpyHostToDevice, ctx->st));
  double *Fin = Fa, *Fout = Fb;
  for (int l = 0; l < nl; ++l) {
    Layer& y = ctx->L[l];
    const int M = y.d.M, Mp = y.Mp, D = y.d.D_out, Din = y.d.D_in;
    (void)M;
    const long DM = (long)D * Mp;
    const ZSource zsrc = zsrc_of(ctx, l, zs != nullptr, seed, 0, N);
    for (int s = 0; s < S; ++s) {
      const double* Xs = (l == 0) ? X0 : Fin + (long)s * N * Din;
      HIPCHK(hipMemsetAsync(Kt, 0, (size_t)Pm * Mp * 8, ctx->st));            // pad rows of every product stay zero
      if (y.d.kernel_kind == DGP_KERNEL_MF)           // the composite kernel on [x, f] (mfkern.hip)
        HIPCHK(mf_kuf(ctx->st, Xs, N, 0, P(ctx, y.off_Z), P(ctx, y.off_var), y.d.M, Mp, Din, Kt));
      else
        HIPCHK(rbf_kuf(ctx->st, y.d.kernel_kind, Xs, N, 0, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), y.d.M, Mp, Din,
                       Kt, nullptr));
      {
        GemmArgs a = mk(Pm, Mp, Mp, Kt, Mp, y.Linv, Mp, Ct, Mp);
        a.tri = TRI_B_UPPER; a.triblk = Mp;
        RET(GX(ctx, 0, GEMM_NT, a));
      }
      {
        GemmArgs a = mk(Pm, DM, Mp, Ct, Mp, y.Wcat, DM, T, DM);
        a.tri = TRI_B_LOWER; a.triblk = Mp;
        RET(GX(ctx, 0, GEMM_NN, a));
      }
      RET(GX(ctx, 0, GEMM_NN, mk(Pm, D, Mp, Ct, Mp, y.u, D, mean0, D)));
      HIPCHK(fc_mean(ctx->st, mean0, Xs, N, Din, D, y.d.mean_kind, y.meanW, y.meanb, mean));
      // V_d = K(X_s, X_s) + jitter I - C C^T + T_d T_d^T   (layers.py:265-268 in whitened form; jitter: utils.py:47)
      //   K(X_s, X_s) of a kernel with a White term carries its variance on the diagonal (kern.K(X), layers.py:265)
      if (y.d.kernel_kind == DGP_KERNEL_MF)
        HIPCHK(mf_kuu(ctx->st, Xs, P(ctx, y.off_var), (int)N, Np, Din, Kff));
      else
        HIPCHK(rbf_kuu(ctx->st, y.d.kernel_kind, Xs, P(ctx, y.off_var), P(ctx, y.off_ls), (int)N, Np, Din, Kff, nullptr));
      if (y.off_white >= 0) HIPCHK(add_diag_dev(ctx->st, Kff, (int)N, Np, P(ctx, y.off_white)));
      for (int d = 0; d < D; ++d) HIPCHK(copy_mat(ctx->st, Kff, V + (long)d * NN, NN));
      {
        GemmArgs a = mk(Np, Np, Mp, Ct, Mp, Ct, Mp, V, Np, -1.0, 1);
        a.batch = D; a.sA = 0; a.sB = 0; a.sC = NN;
        RET(GX(ctx, 0, GEMM_NT, a));
      }
      {
        GemmArgs a = mk(Np, Np, Mp, T, DM, T, DM, V, Np, 1.0, 1);
        a.batch = D; a.sA = Mp; a.sB = Mp; a.sC = NN;
        RET(GX(ctx, 0, GEMM_NT, a));
      }
      if (Fvars && Fvars[l]) {
        HIPCHK(fc_export_var(ctx->st, V, N, Np, D, var_dev));
        HIPCHK(hipMemcpyAsync(Fvars[l] + (long)s * N * N * D, var_dev, (size_t)N * N * D * 8, hipMemcpyDeviceToHost, ctx->st));
      }
      if (Fmeans && Fmeans[l])
        HIPCHK(hipMemcpyAsync(Fmeans[l] + (long)s * N * D, mean, (size_t)N * D * 8, hipMemcpyDeviceToHost, ctx->st));
      HIPCHK(potrf_inv(ctx->st, V, Vinv, tmp, Np, D, ctx->info));
      HIPCHK(fc_sample(ctx->st, V, mean, N, Np, D, zsrc, s, Fout + (long)s * N * D));
      if (Fvars && Fvars[l]) HIPCHK(hipStreamSynchronize(ctx->st));           // var_dev is reused by the next sample
    }
    if (Fs && Fs[l])
      HIPCHK(hipMemcpyAsync(Fs[l], Fout, (size_t)S * N * D * 8, hipMemcpyDeviceToHost, ctx->st));
    std::swap(Fin, Fout);
  }
  return check_flags(ctx);
}

static int vjp_body(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed, const double* const* zs,
                    const double* f_bar, const double* mean_bar, const double* var_bar, double* xbar_out, bool params,
                    bool reset) {
  RET(check_ready(ctx, false));
  if (!Xnew || (!xbar_out && !params) || Nn <= 0 || S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_propagate_vjp: bad arguments");
  if (!f_bar && !mean_bar && !var_bar) return fail(ctx, DGP_ERR_INVALID, "dgp_propagate_vjp: no cotangent given");
  HIPCHK(hipSetDevice(ctx->device));
  const int nl = (int)ctx->L.size();
  const int Din0 = ctx->L[0].d.D_in;
  Layer& last = ctx->L.back();
  const int DL = last.d.D_out;
  RET(grow(ctx, &ctx->Xnew, &ctx->Xnew_cap, (size_t)Nn * Din0));
  HIPCHK(hipMemcpyAsync(ctx->Xnew, Xnew, (size_t)Nn * Din0 * 8, hipMemcpyHostToDevice, ctx->st));
  if (zs) RET(upload_zs(ctx, zs, S, Nn));
  // the output staging buffers of dgp_propagate double as staging for the cotangents ([S, Nn, D_L] each)
  const double* host_bar[3] = {f_bar, mean_bar, var_bar};
  const double* dev_bar[3] = {nullptr, nullptr, nullptr};
  for (int k = 0; k < 3; ++k) {
    if ((int)ctx->out_dev[k].size() < nl) { ctx->out_dev[k].resize(nl, nullptr); ctx->out_cap[k].resize(nl, 0); }
    if (!host_bar[k]) continue;
    const size_t n = (size_t)S * Nn * DL;
    RET(grow(ctx, &ctx->out_dev[k][nl - 1], &ctx->out_cap[k][nl - 1], n));
    HIPCHK(hipMemcpyAsync(ctx->out_dev[k][nl - 1], host_bar[k], n * 8, hipMemcpyHostToDevice, ctx->st));
    dev_bar[k] = ctx->out_dev[k][nl - 1];
  }
  RET(prep(ctx, true));
  if (params) {
    ctx->grad_ready = false;
    if (reset) HIPCHK(hipMemsetAsync(ctx->acc, 0, ctx->n_acc * 8, ctx->st));
  }
  long Nc = 0;
  RET(ensure_ws(ctx, Nn, S, true, &Nc));
  const bool dedup_last = nl == 1;
  for (long n0 = 0; n0 < Nn; n0 += Nc) {
    const long nc = std::min(Nc, (long)Nn - n0);
    RET(forward_chunk(ctx, ctx->Xnew, Nn, n0, nc, S, seed, zs != nullptr, 0));
    {
      ProfScope ps(ctx, 1, 0, 0);
      HIPCHK(vjp_seed(ctx->st, dev_bar[0], dev_bar[1], dev_bar[2], last.var, nc, S, dedup_last ? 1 : 0, DL,
                      zsrc_of(ctx, nl - 1, zs != nullptr, seed, 0, Nn), Nn, n0, last.mbar, last.vbar,
                      params ? ctx->acc + last.acc_dvar : nullptr));
    }
    RET(backward_chunk(ctx, n0, nc, S, seed, zs != nullptr, BwdOpts{ctx->Xnew, (long)Nn, 0, params, true}));
    if (xbar_out)
      HIPCHK(hipMemcpyAsync(xbar_out + n0 * Din0, ctx->xbar, (size_t)nc * Din0 * 8, hipMemcpyDeviceToHost, ctx->st));
  }
  return check_flags(ctx);
}

int dgp_propagate_vjp(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed, const double* const* zs,
                      const double* f_bar, const double* mean_bar, const double* var_bar, double* xbar_out) {
  if (!xbar_out) return fail(ctx, DGP_ERR_INVALID, "dgp_propagate_vjp: bad arguments");
  return vjp_body(ctx, Xnew, Nn, S, seed, zs, f_bar, mean_bar, var_bar, xbar_out, false, false);
}

int dgp_vjp_accumulate(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed, const double* const* zs,
                       const double* f_bar, const double* mean_bar, const double* var_bar, double* xbar_out,
                       int32_t reset) {
  return vjp_body(ctx, Xnew, Nn, S, seed, zs, f_bar, mean_bar, var_bar, xbar_out, true, reset != 0);
}

int dgp_grad_partial(dgp_ctx* ctx, int32_t S, uint64_t seed, const double* const* zs) {
  RET(check_ready(ctx, true));
  if (S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_grad_partial: S must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->grad_ready = false;
  if (zs) RET(upload_zs(ctx, zs, S, ctx->N));
  SideJoin join{ctx, ctx->st, ctx->use_side && ctx->events_ok};
  RET(prep(ctx, true, true, true));      // the upper layers' chains run under the first layer's forward pass; the partial sums start at zero
  long Nc = 0;
  const long lo = ctx->batch_n ? ctx->batch_lo : 0, hi = ctx->batch_n ? ctx->batch_lo + ctx->batch_n : ctx->N;
  RET(ensure_ws(ctx, hi - lo, S, true, &Nc));
  Layer& last = ctx->L.back();
  const bool dedup_last = ctx->L.size() == 1;
  for (long n0 = lo; n0 < hi; n0 += Nc) {
    const long nc = std::min(Nc, hi - n0);
    RET(forward_chunk(ctx, ctx->X, ctx->N, n0, nc, S, seed, zs != nullptr, ctx->n_goff));
    {
      ProfScope ps(ctx, 1, 0, 0);
      HIPCHK(gauss_lik(ctx->st, last.mean, last.var, ctx->Y, n0, nc, S, dedup_last ? 1 : 0, ctx->Dy,
                       P(ctx, ctx->n_params - 1), ctx->acc + 0, ctx->acc + 1, last.mbar, last.vbar,
                       ctx->acc + last.acc_dvar, ctx->data_scale));
    }
    RET(backward_chunk(ctx, n0, nc, S, seed, zs != nullptr, BwdOpts{ctx->X, ctx->N, ctx->n_goff, true, false}));
  }
  RET(prep_flush(ctx));            // (no-op unless no chunk ran)
  if (ctx->xfer_on) {              // a caller reduces the sums over ranks: hand them over in transport form
    ProfScope ps(ctx, 1, 0, (double)(ctx->n_xfer) * 16);
    HIPCHK(acc_xfer(ctx->st, ctx->xfer_tab, ctx->n_xfer_seg, ctx->acc, ctx->acc_xfer, false));
    ctx->xfer_pending = true;
  }
  return DGP_OK;
}

// The transport form of the partial sums (dgp_ctx.h: acc_xfer): segment table + library-owned buffer, built the first time a
// caller asks for the buffer.  Q' travels only when a layer still reduces it over the points (decided from what is fixed at
// dgp_model_set time: a layer for which it is assembled behind the reduction never needs it).
static int xfer_setup(dgp_ctx* ctx) {
  if (ctx->xfer_tab) return DGP_OK;
  std::vector<long> tab;
  long dst = 0;
  auto seg = [&](long kind, long src, long n) {
    tab.push_back(kind); tab.push_back(src); tab.push_back(dst); tab.push_back(n);
    dst += kind ? n * (n + 1) / 2 : n;
  };
  seg(0, 0, 4);                                                   // the scalars at the head (ELBO data term, d / d likelihood variance)
  const int nl = (int)ctx->L.size();
  for (int l = 0; l < nl; ++l) {
    const Layer& y = ctx->L[l];
    const long MM = (long)y.Mp * y.Mp;
    const bool q_after = ctx->q_from_g && !ctx->chain_on && y.d.kernel_kind != DGP_KERNEL_MF && !small_fused(ctx, y);
    if (!q_after) seg(0, y.acc_Q, MM);
    for (int d = 0; d < y.d.D_out; ++d) seg(1, y.acc_G + (long)d * MM, y.Mp);
    const long end = (l + 1 < nl) ? ctx->L[l + 1].acc_Q : ctx->n_acc;
    seg(0, y.acc_du, end - y.acc_du);
  }
  ctx->n_xfer = dst;
  ctx->n_xfer_seg = (int)(tab.size() / 4);
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&ctx->xfer_tab), tab.size() * sizeof(long)));
  HIPCHK(hipMemcpy(ctx->xfer_tab, tab.data(), tab.size() * sizeof(long), hipMemcpyHostToDevice));
  RET(dev_alloc(ctx, &ctx->acc_xfer_own, dst));
  HIPCHK(hipMemset(ctx->acc_xfer_own, 0, dst * 8));
  ctx->acc_xfer = ctx->acc_xfer_own;
  return DGP_OK;
}

int dgp_acc_info(dgp_ctx* ctx, void** device_ptr, int64_t* n_doubles) {
  if (!ctx || ctx->L.empty()) return fail(ctx, DGP_ERR_INVALID, "dgp_acc_info: no model");
  RET(xfer_setup(ctx));
  ctx->xfer_on = true;                 // from now on dgp_grad_partial ends by packing, dgp_grad_finish starts by unpacking
  if (device_ptr) *device_ptr = ctx->acc_xfer;
  if (n_doubles) *n_doubles = ctx->n_xfer;
  return DGP_OK;
}

int dgp_acc_bind(dgp_ctx* ctx, void* external_device_ptr) {
  if (!ctx || ctx->L.empty()) return fail(ctx, DGP_ERR_INVALID, "dgp_acc_bind: no model");
  RET(xfer_setup(ctx));
  ctx->xfer_on = true;
  ctx->acc_xfer = external_device_ptr ? reinterpret_cast<double*>(external_device_ptr) : ctx->acc_xfer_own;
  return DGP_OK;
}

// Small-matrix chain of ONE layer after its point sums are final (and summed over the ranks): Cholesky backward, KL
// gradient, kernel-hyper-parameter gradients -> that layer's entries of d ELBO / d params.  Launches on ctx->st with the
// scratch set ctx->sm (the caller selects a side stream + its scratch for concurrent layers).
static int finish_layer_launches(dgp_ctx* ctx, size_t li) {
  double* acc = ctx->acc;
  double* g = ctx->grad;
    Layer& y = ctx->L[li];
    const int M = y.d.M, Mp = y.Mp, D = y.d.D_out, Din = y.d.D_in;
    const long MM = (long)Mp * Mp, DM = (long)D * Mp;
    double *T1 = ctx->sm[0], *T2 = ctx->sm[1], *T3 = ctx->sm[2], *T4 = ctx->sm[3], *Sm = ctx->sm[4];
    double *Gd = acc + y.acc_G, *du = acc + y.acc_du, *Q = acc + y.acc_Q, *dW = ctx->sm[5];
    {  // dW_d = 2 G_d W_d  (G_d symmetric, accumulated as its lower triangle)
      HIPCHK(symmetrize_lower(ctx->st, Gd, Mp, D));
      GemmArgs a = mk(Mp, Mp, Mp, Gd, Mp, y.Wcat, DM, dW, DM, 2.0, 0);
      a.batch = D; a.sA = MM; a.sB = Mp; a.sC = Mp;
      RET(GX(ctx, 2, GEMM_NN, a));
    }
    if (q_from_sums(ctx, y)) {
      // Q' = u du^T + sum_d (W_d dW_d^T - 2 G_d) from the DATA terms of du and dW (before the KL terms join them below): the
      // reduction over the points that used to deliver it (a Gram launch reading Cbar and Ct: 1.05 ms per 10^6 points) is gone.
      // (y.dLq is free until the q_sqrt gradient is written into it further down: it holds the D products dW_d W_d^T meanwhile)
      RET(G(ctx, 2, GEMM_NT, Mp, Mp, Mp, dW, DM, y.Wcat, DM, y.dLq, Mp, 1.0, 0, D, Mp, Mp, MM));
      HIPCHK(qprime_from_sums(ctx->st, y.dLq, Gd, y.u, du, Mp, D, Q));
    }
    HIPCHK(wbar_total(ctx->st, dW, y.Wcat, du, y.u, M, Mp, D));
    if (y.d.kernel_kind != DGP_KERNEL_MF) {
      // Q' = Cbar^T C (assembled above, or its lower triangle reduced over the points): Q = dK^T C = Linv^T Q'.  Row i >= column j
      // of the product only meets Q'[k][j] with k >= i >= j (Linv is lower triangular): the lower triangle suffices
      RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Linv, Mp, Q, Mp, ctx->sm[6], Mp, 1.0, 0));
      HIPCHK(copy_mat(ctx->st, ctx->sm[6], Q, MM));
    }
    if (y.kdot) HIPCHK(lower_dot(ctx->st, y.Lu, Q, M, Mp, y.kdot));      // before Q is overwritten below
    if (!y.d.white) {
      // T1 = du u^T + sum_d dW_d W_d^T (K = D Mp against an Mp x Mp output)
      if (Mp <= 64) {      // small models: D accumulating one-workgroup products (recordable as chain steps, fixed order)
        RET(G(ctx, 2, GEMM_NT, Mp, Mp, D, du, D, y.u, D, T1, Mp, 1.0, 0));
        for (int d = 0; d < D; ++d)
          RET(G(ctx, 2, GEMM_NT, Mp, Mp, Mp, dW + (long)d * Mp, DM, y.Wcat + (long)d * Mp, DM, T1, Mp, 1.0, 1));
      } else {             // one batched Mp^3 product into the still unused dLq buffer, then the sum over d in order (optim.hip)
        RET(G(ctx, 2, GEMM_NT, Mp, Mp, Mp, dW, DM, y.Wcat, DM, y.dLq, Mp, 1.0, 0, D, Mp, Mp, MM));
        HIPCHK(sum_dprod(ctx->st, y.dLq, du, y.u, Mp, D, T1));
      }
      RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Linv, Mp, T1, Mp, T2, Mp, 1.0, 0));
      RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Linv, Mp, dW, DM, y.dLq, Mp, 1.0, 0, D, 0, Mp, MM));
      RET(G(ctx, 2, GEMM_TN, Mp, D, Mp, y.Linv, Mp, du, D, y.dqmu_p, D, 1.0, 0));
      HIPCHK(lubar_finish(ctx->st, Q, T2, y.Lu, M, Mp, D, 0));
    } else {
      // W_d = L_q,d, u = q_mu: gradients pass straight through
      for (int d = 0; d < D; ++d) HIPCHK(copy_2d(ctx->st, dW + (long)d * Mp, DM, y.dLq + (long)d * MM, Mp, Mp, Mp));
      HIPCHK(copy_mat(ctx->st, du, y.dqmu_p, (long)Mp * D));
      HIPCHK(lubar_finish(ctx->st, Q, nullptr, y.Lu, M, Mp, D, 1));
    }
    HIPCHK(lqbar_finish(ctx->st, y.dLq, y.Lq, M, Mp, D));
    // Cholesky backward: dKuu = sym(Lu^-T Phi(Lu^T dLu) Lu^-1)      
    RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Lu, Mp, Q, Mp, T3, Mp, 1.0, 0));
    HIPCHK(phi_tril_halfdiag(ctx->st, T3, Mp, 1));
    RET(G(ctx, 2, GEMM_NN, Mp, Mp, Mp, T3, Mp, y.Linv, Mp, T4, Mp, 1.0, 0));
    RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Linv, Mp, T4, Mp, Sm, Mp, 1.0, 0));
    if (y.d.kernel_kind == DGP_KERNEL_MF) {
      HIPCHK(mf_collect(ctx->st, acc + y.acc_GX, acc + y.acc_x2, M, Din, g + y.off_Z, g + y.off_var));
      HIPCHK(mf_kuu_bwd(ctx->st, Sm, P(ctx, y.off_Z), P(ctx, y.off_var), M, Mp, Din, g + y.off_Z, g + y.off_var));
    } else {
      HIPCHK(rbf_kuf_bwd_finish(ctx->st, acc + y.acc_GX, acc + y.acc_x2, acc + y.acc_dvar, P(ctx, y.off_Z), P(ctx, y.off_var),
                                P(ctx, y.off_ls), M, Din, g + y.off_Z, g + y.off_ls, g + y.off_var, y.kdot));
      HIPCHK(rbf_kuu_bwd(ctx->st, Sm, y.Kuu, y.Euu, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), M, Mp, Din,
                         g + y.off_Z, g + y.off_ls, g + y.off_var, y.off_white >= 0 ? P(ctx, y.off_white) : nullptr));
    }
    if (y.off_white >= 0) HIPCHK(white_grad(ctx->st, Sm, M, Mp, acc + y.acc_dvar, g + y.off_white));
    HIPCHK(unpack_q_grads(ctx->st, y.dLq, y.dqmu_p, M, Mp, D, g + y.off_qsqrt, g + y.off_qmu));
  return DGP_OK;
}

static int finish_layer(dgp_ctx* ctx, size_t li) {
  {
    Layer& y = ctx->L[li];
    if (small_fused(ctx, y) && y.off_white < 0 && !ctx->chain_on) {      // at most 64 inducing points: one launch (small_layer.hip)
      double* acc = ctx->acc;
      double* g = ctx->grad;
      ProfScope ps(ctx, 2, 0, 0);
      HIPCHK(small_finish(ctx->st, acc + y.acc_G, acc + y.acc_du, acc + y.acc_Q, acc + y.acc_GX, acc + y.acc_x2, acc + y.acc_dvar,
                          y.Wcat, y.u, y.Linv, y.Lu, y.Lq, y.Kuu, y.Euu, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), y.d.M,
                          y.d.D_out, y.d.D_in, y.d.white, y.dLq, y.dqmu_p, g + y.off_Z, g + y.off_ls, g + y.off_var, g + y.off_qsqrt,
                          g + y.off_qmu, y.kdot));
      return DGP_OK;
    }
  }
  return run_chain(ctx, CHAIN_FINISH, li, [&]() { return finish_layer_launches(ctx, li); });
}

static int finish_tail(dgp_ctx* ctx) {
  HIPCHK(finish_tail_launch(ctx->st, ctx->acc, ctx->scal, ctx->grad + ctx->n_params - 1));   // ELBO = data term - sum KL
  ctx->grad_ready = true;
  return DGP_OK;
}

int dgp_grad_finish(dgp_ctx* ctx, double* elbo_out) {
  RET(check_ready(ctx, false));
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->xfer_on && ctx->xfer_pending) {     // the (summed) transport buffer of the last dgp_grad_partial back into the working squares
    ctx->xfer_pending = false;
    ProfScope ps(ctx, 1, 0, (double)(ctx->n_xfer) * 16);
    HIPCHK(acc_xfer(ctx->st, ctx->xfer_tab, ctx->n_xfer_seg, ctx->acc, ctx->acc_xfer, true));
  }
  {
    LayerFork fork(ctx, (int)ctx->L.size());
    for (size_t li = 0; li < ctx->L.size(); ++li) {
      fork.use((int)li);
      RET(finish_layer(ctx, li));
    }
  }
  RET(finish_tail(ctx));
  if (elbo_out) return dgp_last_elbo(ctx, elbo_out);
  return DGP_OK;
}

// ------------------------------------------------------------------------------- RCCL (loaded at run time)
namespace {
struct NcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, ncclUniqueIdBytes, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
NcclApi g_nccl;
constexpr int kNcclFloat64 = 8, kNcclSum = 0;     // ncclDataType_t / ncclRedOp_t values of rccl.h

int nccl_load(dgp_ctx* ctx) {
  if (g_nccl.lib) return DGP_OK;
  // RCCL must sit on the SAME HIP runtime this library is bound to: a process that also holds torch may contain two ROCm
  // stacks (torch ships its own libamdhip64 / librccl with the system's sonames; which libamdhip64 this library got
  // depends on what was loaded first), and a communicator created by the other stack's RCCL fails with "unhandled cuda
  // error" on this library's streams and buffers.  So: the librccl next to the libamdhip64 that hipGetDeviceCount resolves to
  // (when torch was loaded first that is torch's copy, already in the process: same handle), else the soname.
  void* h = nullptr;
  {
    Dl_info di;
    if (dladdr(reinterpret_cast<void*>(static_cast<hipError_t (*)(int*)>(&hipGetDeviceCount)), &di) && di.dli_fname) {
      std::string dir(di.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        for (const char* nm : {"librccl.so.1", "librccl.so"}) {
          h = dlopen((dir + nm).c_str(), RTLD_NOW | RTLD_GLOBAL);
          if (h) break;
        }
      }
    }
  }
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(ctx, DGP_ERR_INVALID, "dgp_comm: librccl.so not found");
  NcclApi a;
  a.lib = h;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.GroupStart || !a.GroupEnd || !a.CommDestroy)
    return fail(ctx, DGP_ERR_INVALID, "dgp_comm: librccl.so lacks an expected symbol");
  g_nccl = a;
  return DGP_OK;
}
int nccl_chk(dgp_ctx* ctx, int rc, const char* what) {
  if (rc == 0) return DGP_OK;
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "RCCL error");
  return fail(ctx, DGP_ERR_HIP, buf);
}
}  // namespace

// ------------------------------------------------------------------------------- one ROCm stack per process
// A process that holds two copies of libamdhip64 (torch's wheel ships one with the system ROCm's soname and asks for it by
// file name) hands one stack's streams and buffers to the other's RCCL in dgp_comm_init ("unhandled cuda error") and aborts
// at exit (double free).  Nothing used to detect that state (NOTES 12.3); these do.
namespace {
struct RuntimeScan { std::vector<std::string> paths; };
int runtime_scan_cb(struct dl_phdr_info* info, size_t, void* data) {
  const char* nm = info->dlpi_name;
  if (!nm || !*nm) return 0;
  const char* base = strrchr(nm, '/');
  base = base ? base + 1 : nm;
  if (strncmp(base, "libamdhip64.so", 14) != 0) return 0;
  auto* sc = static_cast<RuntimeScan*>(data);
  char real[PATH_MAX];
  std::string p = realpath(nm, real) ? std::string(real) : std::string(nm);
  for (const std::string& q : sc->paths)
    if (q == p) return 0;
  sc->paths.push_back(p);
  return 0;
}
}  // namespace

int dgp_hip_runtimes(char* paths_out, int cap) {
  RuntimeScan sc;
  dl_iterate_phdr(runtime_scan_cb, &sc);
  if (paths_out && cap > 0) {
    std::string joined;
    for (size_t i = 0; i < sc.paths.size(); ++i) joined += (i ? "\n" : "") + sc.paths[i];
    snprintf(paths_out, (size_t)cap, "%s", joined.c_str());
  }
  return (int)sc.paths.size();
}

namespace {
int one_runtime_or_fail(dgp_ctx* ctx, const char* who) {
  char paths[2048];
  const int n = dgp_hip_runtimes(paths, (int)sizeof paths);
  if (n <= 1) return DGP_OK;
  for (char* c = paths; *c; ++c)
    if (*c == '\n') *c = ' ';
  char msg[2400];
  snprintf(msg, sizeof msg,
           "%s: %d copies of libamdhip64 are mapped in this process (%s): import torch BEFORE loading libdgp_hip.so so that "
           "both use one ROCm stack", who, n, paths);
  return fail(ctx, DGP_ERR_INVALID, msg);
}
}  // namespace

int dgp_comm_available(void) { return nccl_load(nullptr); }

int dgp_comm_unique_id(void* id128_out) {
  if (!id128_out) return DGP_ERR_INVALID;
  RET(nccl_load(nullptr));
  return g_nccl.GetUniqueId(id128_out) == 0 ? DGP_OK : DGP_ERR_HIP;
}

int dgp_comm_init(dgp_ctx* ctx, int32_t rank, int32_t world, const void* id128) {
  if (!ctx || world < 1 || rank < 0 || rank >= world) return fail(ctx, DGP_ERR_INVALID, "dgp_comm_init: bad rank / world");
  RET(dgp_comm_destroy(ctx));
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  if (world == 1 && !id128) return DGP_OK;      // single process: nothing to reduce
  if (!id128) return fail(ctx, DGP_ERR_INVALID, "dgp_comm_init: no unique id");
  // (world == 1 WITH an id builds a one-rank communicator: the whole RCCL path can then be exercised on one GPU)
  RET(one_runtime_or_fail(ctx, "dgp_comm_init"));
  RET(nccl_load(ctx));
  HIPCHK(hipSetDevice(ctx->device));
  ncclUniqueIdBytes id;
  memcpy(&id, id128, sizeof id);
  void* comm = nullptr;
  RET(nccl_chk(ctx, g_nccl.CommInitRank(&comm, world, id, rank), "ncclCommInitRank"));
  ctx->nccl_comm = comm;
  // the collectives run beside the backward pass: the persistent kernels leave them a few CUs (gemm_f64.hip)
  const char* e = getenv("DGP_COMM_RESERVE_CUS");
  ctx->reserved_cus = e ? atoi(e) : 8;
  return DGP_OK;
}

int dgp_comm_destroy(dgp_ctx* ctx) {
  if (!ctx) return DGP_ERR_INVALID;
  if (ctx->nccl_comm && g_nccl.CommDestroy) {
    (void)hipStreamSynchronize(ctx->st);
    (void)g_nccl.CommDestroy(ctx->nccl_comm);
  }
  ctx->reserved_cus = 0;
  ctx->nccl_comm = nullptr;
  ctx->comm_world = 1;
  ctx->comm_rank = 0;
  return DGP_OK;
}

int dgp_comm_allreduce(dgp_ctx* ctx, void* device_ptr, int64_t n_doubles) {
  if (!ctx || !device_ptr || n_doubles < 0) return fail(ctx, DGP_ERR_INVALID, "dgp_comm_allreduce: bad argument");
  if (n_doubles == 0 || (ctx->comm_world <= 1 && !ctx->nccl_comm)) return DGP_OK;
  if (!ctx->nccl_comm) return fail(ctx, DGP_ERR_INVALID, "dgp_comm_allreduce: call dgp_comm_init first");
  return nccl_chk(ctx, g_nccl.AllReduce(device_ptr, device_ptr, (size_t)n_doubles, kNcclFloat64, kNcclSum, ctx->nccl_comm, ctx->st),
                  "ncclAllReduce");
}

// ------------------------------------------------------------------------------- overlapped gradient evaluation
namespace {
// backward_chunk hook (last chunk only): layer l's sums are final -> all-reduce its slice on the COMM stream (every
// collective of the communicator is enqueued on that one stream, in the same order on every rank), then the layer's
// small-matrix chain on one of the two chain streams, while the main stream runs the backward pass of the layers below
int after_layer_hook(dgp_ctx* ctx, int l, int phase) {
  const int nl = (int)ctx->L.size();
  Layer& y = ctx->L[l];
  // the layer's contiguous slice; Q' at its head is left out when it is assembled behind the reduction (all zeros until then)
  const long lo = q_from_sums(ctx, y) ? y.acc_G : y.acc_Q, hi = (l + 1 < nl) ? ctx->L[l + 1].acc_Q : ctx->n_acc;
  const bool comm = ctx->nccl_comm != nullptr;
  const bool side = ctx->use_side && ctx->events_ok && nl <= dgp_ctx::kMaxEv;
  hipStream_t main_st = ctx->st;
  if (phase == 0) {                             // layer l's sums are final behind this point of the main stream
    if (side) HIPCHK(hipEventRecord(ctx->ev_layer[l], main_st));
    return DGP_OK;
  }
  hipStream_t chain_st = main_st, comm_st = main_st;
  int w = 0;
  // the first layer is the last one the backward pass leaves: nothing follows on the main stream that its chain could run
  // beside, so it stays there (one cross-stream hand-over less at the tail of every iteration)
  const bool on_main = (l == 0);
  if (side) {
    if (!on_main) {
      w = 1 + ((nl - 1 - l) % 2);               // alternate the two chain streams, top layer first
      chain_st = ctx->side[w - 1];
      ctx->side_touched[w - 1] = true;
    }
    comm_st = ctx->side[dgp_ctx::kSide - 1];
  }
  if (comm) {
    if (side) { HIPCHK(hipStreamWaitEvent(comm_st, ctx->ev_layer[l], 0)); ctx->side_touched[dgp_ctx::kSide - 1] = true; }
    RET(nccl_chk(ctx, g_nccl.GroupStart(), "ncclGroupStart"));
    int rc = 0;
    if (l == nl - 1)      // the two scalars at the head of the buffer (ELBO data term, likelihood-variance gradient)
      rc = g_nccl.AllReduce(ctx->acc, ctx->acc, 2, kNcclFloat64, kNcclSum, ctx->nccl_comm, comm_st);
    if (rc == 0) rc = g_nccl.AllReduce(ctx->acc + lo, ctx->acc + lo, (size_t)(hi - lo), kNcclFloat64, kNcclSum, ctx->nccl_comm, comm_st);
    const int rc2 = g_nccl.GroupEnd();
    RET(nccl_chk(ctx, rc ? rc : rc2, "ncclAllReduce"));
    if (side) {
      HIPCHK(hipEventRecord(ctx->ev_red[l], comm_st));
      HIPCHK(hipStreamWaitEvent(chain_st, ctx->ev_red[l], 0));
    }
  } else if (side && !on_main) {
    HIPCHK(hipStreamWaitEvent(chain_st, ctx->ev_layer[l], 0));
  }
  ctx->st = chain_st;
  ctx->sm = ctx->smset[w];
  const int r = finish_layer(ctx, (size_t)l);
  ctx->st = main_st;
  ctx->sm = ctx->smset[0];
  return r;
}
}  // namespace

int dgp_grad_step(dgp_ctx* ctx, int32_t S, uint64_t seed, const double* const* zs, double* elbo_out) {
  RET(check_ready(ctx, true));
  if (S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_grad_step: S must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->grad_ready = false;
  if (zs) RET(upload_zs(ctx, zs, S, ctx->N));
  const int nl = (int)ctx->L.size();
  {
    SideJoin join{ctx, ctx->st, ctx->use_side && ctx->events_ok};
    RET(prep(ctx, true, true, true));      // (the partial sums start at zero)
    long Nc = 0;
    const long lo = ctx->batch_n ? ctx->batch_lo : 0, hi = ctx->batch_n ? ctx->batch_lo + ctx->batch_n : ctx->N;
    RET(ensure_ws(ctx, hi - lo, S, true, &Nc));
    Layer& last = ctx->L.back();
    const bool dedup_last = ctx->L.size() == 1;
    bool hooked = false;
    for (long n0 = lo; n0 < hi; n0 += Nc) {
      const long nc = std::min(Nc, hi - n0);
      RET(forward_chunk(ctx, ctx->X, ctx->N, n0, nc, S, seed, zs != nullptr, ctx->n_goff));
      {
        ProfScope ps(ctx, 1, 0, 0);
        HIPCHK(gauss_lik(ctx->st, last.mean, last.var, ctx->Y, n0, nc, S, dedup_last ? 1 : 0, ctx->Dy,
                         P(ctx, ctx->n_params - 1), ctx->acc + 0, ctx->acc + 1, last.mbar, last.vbar,
                         ctx->acc + last.acc_dvar, ctx->data_scale));
      }
      BwdOpts o{ctx->X, ctx->N, ctx->n_goff, true, false};
      if (n0 + nc >= hi) { o.after_layer = after_layer_hook; hooked = true; }     // sums are final in the last chunk only
      RET(backward_chunk(ctx, n0, nc, S, seed, zs != nullptr, o));
    }
    RET(prep_flush(ctx));   // (no-op unless no chunk ran: the upper layers' prep chains are still owed)
    if (!hooked) {          // no data points on this rank: the chains still have to run (KL part), after the all-reduces
      for (int l = nl - 1; l >= 0; --l) { RET(after_layer_hook(ctx, l, 0)); RET(after_layer_hook(ctx, l, 1)); }
    }
  }   // side streams joined
  RET(finish_tail(ctx));
  if (elbo_out) return dgp_last_elbo(ctx, elbo_out);
  return DGP_OK;
}

int dgp_last_elbo(dgp_ctx* ctx, double* elbo_out) {
  if (!ctx || !elbo_out) return DGP_ERR_INVALID;
  double h[4];
  HIPCHK(hipMemcpyAsync(h, ctx->scal, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->st));
  RET(check_flags(ctx));
  *elbo_out = h[1];
  return DGP_OK;
}

int dgp_grad_get(dgp_ctx* ctx, double* flat_grad_out) {
  if (!ctx || !flat_grad_out || !ctx->grad_ready) return fail(ctx, DGP_ERR_INVALID, "dgp_grad_get: no gradient (call dgp_grad_finish)");
  HIPCHK(hipMemcpyAsync(flat_grad_out, ctx->grad, ctx->n_params * 8, hipMemcpyDeviceToHost, ctx->st));
  return check_flags(ctx);
}

int dgp_adam_reset(dgp_ctx* ctx) {
  if (!ctx || !ctx->params) return fail(ctx, DGP_ERR_INVALID, "dgp_adam_reset: no model");
  HIPCHK(hipMemsetAsync(ctx->adam_m, 0, ctx->n_params * 8, ctx->st));
  HIPCHK(hipMemsetAsync(ctx->adam_v, 0, ctx->n_params * 8, ctx->st));
  ctx->adam_t = 0;
  return DGP_OK;
}

int dgp_adam_step(dgp_ctx* ctx, double lr, double beta_1, double beta_2, double epsilon, const uint8_t* trainable) {
  if (!ctx || !ctx->grad_ready) return fail(ctx, DGP_ERR_INVALID, "dgp_adam_step: no gradient (call dgp_grad_finish)");
  bool changed = !ctx->segs_uploaded;
  for (size_t i = 0; i < ctx->segs.size(); ++i) {
    const int t = trainable ? (trainable[i] ? 1 : 0) : 1;
    if (ctx->segs[i].trainable != t) { ctx->segs[i].trainable = t; changed = true; }
  }
  if (changed) {   // rare: only when gpflow.set_trainable-style flags change
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipMemcpy(ctx->segs_dev, ctx->segs.data(), ctx->segs.size() * sizeof(ParamSeg), hipMemcpyHostToDevice));
    ctx->segs_uploaded = true;
  }
  ctx->prep_level = 0;
  ctx->adam_t += 1;
  if (ctx->capturing) {          // inside the captured iteration: the step count lives in device memory
    HIPCHK(iter_bump(ctx->st, ctx->it_dev, 0, 1));
    HIPCHK(adam_apply(ctx->st, ctx->params, ctx->grad, ctx->adam_m, ctx->adam_v, ctx->segs_dev, (int)ctx->segs.size(),
                      ctx->n_params, lr, beta_1, beta_2, epsilon, ctx->it_dev + 1));
    return DGP_OK;
  }
  const double t = (double)ctx->adam_t;
  const double lr_t = lr * std::sqrt(1.0 - std::pow(beta_2, t)) / (1.0 - std::pow(beta_1, t));
  ProfScope ps(ctx, 3, 0, (double)ctx->n_params * 48);
  HIPCHK(adam_apply(ctx->st, ctx->params, ctx->grad, ctx->adam_m, ctx->adam_v, ctx->segs_dev, (int)ctx->segs.size(),
                    ctx->n_params, lr_t, beta_1, beta_2, epsilon));
  return DGP_OK;
}

int dgp_natgrad_step(dgp_ctx* ctx, double gamma, const uint8_t* layer_mask) {
  if (!ctx || !ctx->grad_ready) return fail(ctx, DGP_ERR_INVALID, "dgp_natgrad_step: no gradient (call dgp_grad_finish)");
  ctx->prep_level = 0;
  LayerFork fork(ctx, (int)ctx->L.size());
  for (size_t l = 0; l < ctx->L.size(); ++l) {
    if (layer_mask && !layer_mask[l]) continue;
    Layer& y = ctx->L[l];
    fork.use((int)l);
    const int M = y.d.M, Mp = y.Mp, D = y.d.D_out;
    const long MM = (long)Mp * Mp;
    if (small_fused(ctx, y) && !ctx->capturing) {        // at most 64 inducing points: one launch (small_layer.hip)
      HIPCHK(small_natgrad(ctx->st, y.Lq, y.dLq, y.dqmu_p, y.qmu_p, ctx->sm[0], ctx->sm[5], ctx->sm[6], ctx->params + y.off_qsqrt,
                           ctx->params + y.off_qmu, gamma, M, D, ctx->info));
      continue;
    }
    double *Li = ctx->sm[0], *T = ctx->sm[1], *T1 = ctx->sm[2], *Gm = ctx->sm[3], *Pinv = ctx->sm[4], *Pn = ctx->sm[5],
           *Ri = ctx->sm[6], *Sn = ctx->sm[7];
    HIPCHK(trinv_lower(ctx->st, y.Lq, Li, ctx->sm[9], Mp, D));
    RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Lq, Mp, y.dLq, Mp, T, Mp, 1.0, 0, D, MM, MM, MM));
    HIPCHK(phi_tril_halfdiag(ctx->st, T, Mp, D));
    RET(G(ctx, 2, GEMM_NN, Mp, Mp, Mp, T, Mp, Li, Mp, T1, Mp, 1.0, 0, D, MM, MM, MM));
    RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, Li, Mp, T1, Mp, Gm, Mp, 1.0, 0, D, MM, MM, MM));       // d ELBO / d Sigma (unsym.)
    RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, Li, Mp, Li, Mp, Pinv, Mp, 1.0, 0, D, MM, MM, MM));     // Sigma^-1
    HIPCHK(symmetrize_axpy(ctx->st, Gm, Pinv, -2.0 * gamma, Pn, Mp, D));                       // loss = -ELBO
    HIPCHK(potrf_inv(ctx->st, Pn, Ri, ctx->sm[9], Mp, D, ctx->info));
    RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, Ri, Mp, Ri, Mp, Sn, Mp, 1.0, 0, D, MM, MM, MM));       // Sigma'
    HIPCHK(natgrad_mu(ctx->st, Sn, y.dqmu_p, gamma, y.qmu_p, M, Mp, D));
    HIPCHK(copy_mat(ctx->st, Sn, y.Lq, MM * D));
    HIPCHK(potrf_inv(ctx->st, y.Lq, ctx->sm[8], ctx->sm[9], Mp, D, ctx->info));
    HIPCHK(store_q(ctx->st, y.Lq, y.qmu_p, M, Mp, D, ctx->params + y.off_qsqrt, ctx->params + y.off_qmu));
  }
  return DGP_OK;
}

// One loop body of DGP.optimize_adam (dgp.py:271-276) or of part 2 of DGP.optimize_nat_adam (dgp.py:326-345).
static int train_iteration(dgp_ctx* ctx, int32_t S, uint64_t seed, double lr, double b1, double b2, double eps,
                           const uint8_t* trainable, double gamma, const uint8_t* layer_mask) {
  RET(dgp_grad_step(ctx, S, seed, nullptr, nullptr));
  if (ctx->capturing) HIPCHK(iter_bump(ctx->st, ctx->it_dev, 1, 0));             // next evaluation: next seed
  RET(dgp_adam_step(ctx, lr, b1, b2, eps, trainable));
  if (ctx->capturing) HIPCHK(iter_log_elbo(ctx->st, ctx->it_dev, ctx->scal, ctx->elbo_log, dgp_ctx::kLogCap));
  if (gamma > 0.0) {
    RET(dgp_grad_step(ctx, S, seed + 1, nullptr, nullptr));
    if (ctx->capturing) HIPCHK(iter_bump(ctx->st, ctx->it_dev, 1, 0));
    RET(dgp_natgrad_step(ctx, gamma, layer_mask));
  }
  return DGP_OK;
}

int dgp_adam_iterations(dgp_ctx* ctx, int32_t n_iter, int32_t S, uint64_t seed0, double lr, double beta_1, double beta_2,
                        double epsilon, const uint8_t* trainable, double gamma, const uint8_t* layer_mask,
                        int32_t use_graph, double* elbo_out) {
  RET(check_ready(ctx, true));
  if (n_iter < 0 || S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_adam_iterations: bad n_iter / S");
  if (n_iter == 0) return DGP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const int per = gamma > 0.0 ? 2 : 1;                      // ELBO evaluations (= seeds) per iteration
  const int nseg = (int)ctx->segs.size(), nl = (int)ctx->L.size();
  // launch-bound models (the Bayesian-optimisation workloads: tens of inducing points, ~120 launches of a few
  // microseconds per iteration): capture ONE iteration into a hipGraph and replay it.
  // Measured on MI355X / ROCm 7.2 (tools/small_model_bench.py): the replayed graph is SLOWER than the call-by-call
  // launches (config 1: 0.81 against 0.75 ms per iteration; notebook model 0.90 against 0.77) -- the replay runs the
  // captured side-stream branches one after the other, and these iterations are bound by the dependent chain of ~85
  // kernels of 4-15 us, not by the host's launch rate.  That was round 2.  With the fused small-layer kernels (round 3: ~21
  // launches per iteration) the replay wins where the iteration has no side-stream branches to speak of - one hidden layer:
  // config 1 0.381 against 0.404 ms, nat-adam 0.96 against 1.05 - and still loses with two hidden layers (notebook model 0.48
  // against 0.43: three chains that the replay runs one after the other).  -1 ("by size") therefore means: models of at most two
  // SVGP layers whose layers all run on the fused small-layer kernels; 1 or DGP_GRAPH=1 asks for the graph, 0 / DGP_GRAPH=0 refuses.
  bool graph = use_graph > 0;
  if (use_graph < 0 && nl <= 2) {
    graph = true;
    for (int l = 0; l < nl; ++l) graph = graph && small_fused(ctx, ctx->L[l]) && ctx->L[l].off_white < 0;
  }
  {
    static int env = -2;
    if (env == -2) { const char* e = getenv("DGP_GRAPH"); env = e ? atoi(e) : -1; }
    if (env == 0) graph = false;
    if (env > 0 && use_graph != 0) graph = true;
  }
  if (ctx->prof.on || ctx->nccl_comm || ctx->comm_world > 1 || n_iter < 3 || !ctx->use_side) graph = false;
  std::vector<double> elbos((size_t)n_iter, 0.0);
  // one iteration launched call by call; the ELBO the iteration would print is read back only when asked for
  auto eager_iter = [&](int i) -> int {
    const uint64_t sd = seed0 + (uint64_t)per * i;
    RET(dgp_grad_step(ctx, S, sd, nullptr, nullptr));
    RET(dgp_adam_step(ctx, lr, beta_1, beta_2, epsilon, trainable));
    if (elbo_out) RET(dgp_last_elbo(ctx, &elbos[i]));
    if (per == 2) {
      RET(dgp_grad_step(ctx, S, sd + 1, nullptr, nullptr));
      RET(dgp_natgrad_step(ctx, gamma, layer_mask));
    }
    return DGP_OK;
  };
  int done = 0;
  if (graph) {
    if (!ctx->it_dev) { RET(dev_alloc(ctx, &ctx->it_dev, 4)); RET(dev_alloc(ctx, &ctx->elbo_log, dgp_ctx::kLogCap)); }
    char keybuf[512];
    snprintf(keybuf, sizeof keybuf, "S%d N%ld b%ld+%ld s%.17g lr%.17g b1%.17g b2%.17g e%.17g g%.17g ws%p acc%p X%p ", S, ctx->N,
             ctx->batch_lo, ctx->batch_n, ctx->data_scale, lr, beta_1, beta_2, epsilon, gamma, (void*)ctx->ws, (void*)ctx->acc,
             (void*)ctx->X);
    std::string key = keybuf;
    for (int i = 0; i < nseg; ++i) key += (trainable ? (trainable[i] ? '1' : '0') : '1');
    key += ' ';
    for (int l = 0; l < nl; ++l) key += (layer_mask ? (layer_mask[l] ? '1' : '0') : '1');
    if (!ctx->graph_exec || ctx->graph_key != key) {
      drop_graph(ctx);
      // one eager iteration first: allocations (workspace, scratch), the upload of the trainable flags and whatever else
      // must not happen inside a capture; it is iteration 0 of this call
      RET(eager_iter(0));
      done = 1;
      // (the workspace may just have been allocated: the key is built from the pointers as they are now)
      snprintf(keybuf, sizeof keybuf, "S%d N%ld b%ld+%ld s%.17g lr%.17g b1%.17g b2%.17g e%.17g g%.17g ws%p acc%p X%p ", S, ctx->N,
               ctx->batch_lo, ctx->batch_n, ctx->data_scale, lr, beta_1, beta_2, epsilon, gamma, (void*)ctx->ws, (void*)ctx->acc,
               (void*)ctx->X);
      key = keybuf;
      for (int i = 0; i < nseg; ++i) key += (trainable ? (trainable[i] ? '1' : '0') : '1');
      key += ' ';
      for (int l = 0; l < nl; ++l) key += (layer_mask ? (layer_mask[l] ? '1' : '0') : '1');
      HIPCHK(hipStreamSynchronize(ctx->st));
      const long adam_t0 = ctx->adam_t;
      ctx->capturing = true;
      hipError_t e = hipStreamBeginCapture(ctx->st, hipStreamCaptureModeRelaxed);
      int rc = DGP_OK;
      if (e == hipSuccess) {
        rc = train_iteration(ctx, S, 0, lr, beta_1, beta_2, epsilon, trainable, gamma, layer_mask);
        hipGraph_t gph = nullptr;
        e = hipStreamEndCapture(ctx->st, &gph);
        ctx->graph = gph;
      }
      ctx->capturing = false;
      ctx->adam_t = adam_t0;               // the captured launches did not run
      ctx->prep_level = 0;
      if (e != hipSuccess || rc != DGP_OK || !ctx->graph ||
          hipGraphInstantiate(&ctx->graph_exec, ctx->graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        drop_graph(ctx);
        graph = false;                     // capture refused: the call finishes call by call
      } else {
        ctx->graph_key = key;
      }
    }
  }
  if (!graph) {
    for (int i = done; i < n_iter; ++i) RET(eager_iter(i));
    if (elbo_out) memcpy(elbo_out, elbos.data(), sizeof(double) * n_iter);
    return DGP_OK;
  }
  // device-side iteration state: seed of the next evaluation, Adam step count so far, index into the ELBO log
  for (int base = done; base < n_iter; base += dgp_ctx::kLogCap) {
    const int cnt = std::min(n_iter - base, (int)dgp_ctx::kLogCap);
    double st[4] = {0.0, (double)ctx->adam_t, 0.0, 0.0};
    const uint64_t sd = seed0 + (uint64_t)per * base;
    memcpy(&st[0], &sd, 8);
    HIPCHK(hipMemcpyAsync(ctx->it_dev, st, sizeof st, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));       // (st lives on this stack frame)
    for (int i = 0; i < cnt; ++i) HIPCHK(hipGraphLaunch(ctx->graph_exec, ctx->st));
    ctx->adam_t += cnt;
    if (elbo_out) HIPCHK(hipMemcpyAsync(elbos.data() + base, ctx->elbo_log, sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->st));
    RET(check_flags(ctx));                       // synchronises; a failed Cholesky inside the replays surfaces here
  }
  ctx->prep_level = 0;
  ctx->grad_ready = true;
  if (elbo_out) memcpy(elbo_out, elbos.data(), sizeof(double) * n_iter);
  return DGP_OK;
}

int dgp_prof_enable(dgp_ctx* ctx, int32_t on) {
  if (!ctx) return DGP_ERR_INVALID;
  RET(prof_drain(ctx));
  Prof& p = ctx->prof;
  p.on = on != 0;
  p.mask = (on & 0x100) ? (unsigned)(on & 0xF) : 0xFu;
  p.depth = 0;
  for (int c = 0; c < kNCat; ++c) { p.ms[c] = p.flops[c] = p.bytes[c] = 0; p.launches[c] = 0; }
  return DGP_OK;
}

int dgp_prof_read(dgp_ctx* ctx, int32_t n_cat, double* ms_out, int64_t* launches_out, double* alg_flops_out,
                  double* alg_bytes_out) {
  if (!ctx) return DGP_ERR_INVALID;
  RET(prof_drain(ctx));
  Prof& p = ctx->prof;
  for (int c = 0; c < n_cat && c < kNCat; ++c) {
    if (ms_out) ms_out[c] = p.ms[c];
    if (launches_out) launches_out[c] = p.launches[c];
    if (alg_flops_out) alg_flops_out[c] = p.flops[c];
    if (alg_bytes_out) alg_bytes_out[c] = p.bytes[c];
    p.ms[c] = p.flops[c] = p.bytes[c] = 0;
    p.launches[c] = 0;
  }
  return DGP_OK;
}

int dgp_prof_mark(dgp_ctx* ctx) {
  if (!ctx) return DGP_ERR_INVALID;
  Prof& p = ctx->prof;
  if (p.marks_used == p.marks.size()) {
    if (p.marks.size() >= 65536) return fail(ctx, DGP_ERR_INVALID, "dgp_prof_mark: too many marks (read them first)");
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    p.marks.push_back(e);
  }
  HIPCHK(hipEventRecord(p.marks[p.marks_used++], ctx->st));
  return DGP_OK;
}

int dgp_prof_marks_read(dgp_ctx* ctx, int32_t n_max, double* ms_between_out, int32_t* n_out) {
  if (!ctx || n_max < 0 || (n_max > 0 && !ms_between_out)) return fail(ctx, DGP_ERR_INVALID, "dgp_prof_marks_read: bad arguments");
  Prof& p = ctx->prof;
  int n = 0;
  if (p.marks_used > 0) HIPCHK(hipEventSynchronize(p.marks[p.marks_used - 1]));
  for (size_t i = 0; i + 1 < p.marks_used && n < n_max; ++i, ++n) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, p.marks[i], p.marks[i + 1]));
    ms_between_out[n] = ms;
  }
  p.marks_used = 0;
  if (n_out) *n_out = n;
  return DGP_OK;
}

// ----------------------------------------------------------------------------------- unit-level hooks
int dgp_dev_gemm(dgp_ctx* ctx, int32_t op, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B,
                 int64_t ldb, double* C, int64_t ldc, double alpha, int32_t beta, int32_t splits, int32_t tri,
                 int64_t triblk, int32_t repeats, double* ms_per_call) {
  if (!ctx || !A || !B || !C || op < 0 || op > 2) return fail(ctx, DGP_ERR_INVALID, "dgp_dev_gemm: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  const long ar = (op == GEMM_TN) ? K : M, br = (op == GEMM_NT) ? N : K;
  double *dA, *dB, *dC;
  const bool same = (A == B) && ar == br && lda == ldb;     // one operand passed twice stays ONE device buffer (Gram form)
  RET(dev_alloc(ctx, &dA, (size_t)ar * lda));
  if (same) dB = dA; else RET(dev_alloc(ctx, &dB, (size_t)br * ldb));
  RET(dev_alloc(ctx, &dC, (size_t)M * ldc));
  // reductions over the points with a 256 x 256 lower-triangular output: the Gram kernel's scratch (gemm_gram.h)
  if (op == GEMM_TN && M == 256 && N == 256 && tri == TRI_OUT_LOWER && !ctx->gram_ws)
    RET(dev_alloc(ctx, &ctx->gram_ws, gemm_gram_ws_bytes(ctx->cu_count) / 8));
  HIPCHK(hipMemcpy(dA, A, (size_t)ar * lda * 8, hipMemcpyHostToDevice));
  if (!same) HIPCHK(hipMemcpy(dB, B, (size_t)br * ldb * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dC, C, (size_t)M * ldc * 8, hipMemcpyHostToDevice));
  GemmArgs ga = mk(M, N, K, dA, lda, dB, ldb, dC, ldc, alpha, beta);
  ga.splits = splits; ga.tri = tri; ga.triblk = triblk;
  if (ctx->gram_ws) { ga.gram_ws = ctx->gram_ws; ga.gram_ws_bytes = gemm_gram_ws_bytes(ctx->cu_count); }
  int rc = GX(ctx, 0, (GemmOp)op, ga);
  if (rc == DGP_OK) {
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipMemcpy(C, dC, (size_t)M * ldc * 8, hipMemcpyDeviceToHost));
    if (repeats > 0 && ms_per_call) {
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipEventRecord(e0, ctx->st));
      for (int r = 0; r < repeats && rc == DGP_OK; ++r)
        rc = GX(ctx, 0, (GemmOp)op, ga);
      HIPCHK(hipEventRecord(e1, ctx->st));
      HIPCHK(hipEventSynchronize(e1));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e0, e1));
      *ms_per_call = ms / repeats;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
  }
  dev_free(dA); if (!same) dev_free(dB);
  dev_free(dC);
  return rc;
}

int dgp_dev_gram(dgp_ctx* ctx, const double* C, const double* s, int64_t Pn, int32_t D, double* G, const double* mb, double* du) {
  if (!ctx || !C || !G || Pn <= 0 || D < 1 || D > 64 || (!s && D != 1) || ((mb != nullptr) != (du != nullptr)) || (mb && !s))
    return fail(ctx, DGP_ERR_INVALID, "dgp_dev_gram: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  const int Mp = 256;
  const long MM = (long)Mp * Mp;
  double *dC = nullptr, *ds = nullptr, *dG = nullptr, *dmb = nullptr, *ddu = nullptr;
  struct Free { double **a, **b, **c, **d, **e; ~Free() { dev_free(*a); dev_free(*b); dev_free(*c); dev_free(*d); dev_free(*e); } } freer{&dC, &ds, &dG, &dmb, &ddu};
  RET(dev_alloc(ctx, &dC, (size_t)Pn * Mp));
  RET(dev_alloc(ctx, &dG, (size_t)D * MM));
  if (s) RET(dev_alloc(ctx, &ds, (size_t)Pn * D));
  if (mb) { RET(dev_alloc(ctx, &dmb, (size_t)Pn * D)); RET(dev_alloc(ctx, &ddu, (size_t)Mp * D)); }
  if (!ctx->gram_ws) RET(dev_alloc(ctx, &ctx->gram_ws, gemm_gram_ws_bytes(ctx->cu_count) / 8));
  HIPCHK(hipMemcpy(dC, C, (size_t)Pn * Mp * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dG, G, (size_t)D * MM * 8, hipMemcpyHostToDevice));
  if (s) HIPCHK(hipMemcpy(ds, s, (size_t)Pn * D * 8, hipMemcpyHostToDevice));
  if (mb) {
    HIPCHK(hipMemcpy(dmb, mb, (size_t)Pn * D * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ddu, du, (size_t)Mp * D * 8, hipMemcpyHostToDevice));
  }
  // the call backward_chunk makes (dgp_ctx.h: args_G; du = Ct^T mbar rides on it when the Gram kernel runs it)
  GemmArgs a = s ? args_G(ctx, Pn, Mp, D, dC, ds, dG) : mk(Mp, Mp, Pn, dC, Mp, dC, Mp, dG, Mp, 1.0, 1);
  if (!s) {
    a.tri = TRI_OUT_LOWER; a.triblk = Mp; a.splits = pick_splits_tiles(ctx, lower_tiles(Mp), Pn, (long)Mp * 8);
    a.gram_ws = ctx->gram_ws; a.gram_ws_bytes = gemm_gram_ws_bytes(ctx->cu_count);
  }
  if (mb) {
    if (gemm_engine_of(GEMM_TN, a) == 4) { a.gram_mb = dmb; a.gram_du = ddu; }
    else RET(GX(ctx, 0, GEMM_TN, args_du(ctx, Pn, Mp, D, dC, dmb, ddu)));
  }
  RET(GX(ctx, 0, GEMM_TN, a));
  HIPCHK(hipStreamSynchronize(ctx->st));
  HIPCHK(hipMemcpy(G, dG, (size_t)D * MM * 8, hipMemcpyDeviceToHost));
  if (mb) HIPCHK(hipMemcpy(du, ddu, (size_t)Mp * D * 8, hipMemcpyDeviceToHost));
  return DGP_OK;
}

int dgp_dev_layer_products(dgp_ctx* ctx, int64_t Pn, int32_t Mp, int32_t D, const double* Kt, const double* Linv,
                           const double* Wcat, const double* u, const double* vbar, const double* mbar, double* Ct,
                           double* cn, double* T, double* tn, double* mean0, double* Cbar, double* g, double* du,
                           double* Gd, int32_t* engines) {
  if (!ctx || Pn <= 0 || Mp <= 0 || Mp % 64 != 0 || D < 1 || D > 64 || !Kt || !Linv || !Wcat || !u || !vbar || !mbar || !Ct || !cn ||
      !T || !tn || !mean0 || !Cbar || !g || !du || !Gd)
    return fail(ctx, DGP_ERR_INVALID, "dgp_dev_layer_products: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  const long P = Pn, Pm = pad_rows(P), MM = (long)Mp * Mp, DM = (long)D * Mp;
  const int maxpl = Mp / 32;
  std::vector<double*> bufs;
  auto take = [&](double** p, size_t n, bool zero) -> int {
    RET(dev_alloc(ctx, p, n));
    bufs.push_back(*p);
    if (zero) HIPCHK(hipMemsetAsync(*p, 0, n * 8, ctx->st));
    return DGP_OK;
  };
  struct Free { std::vector<double*>& b; ~Free() { for (double* q : b) (void)hipFree(q); } } freer{bufs};
  double *dKt, *dLinv, *dLinvT, *dW, *dS, *du_, *dvb, *dmb, *dCt, *cnp, *tnp, *dT, *dm0, *dCb, *dKb, *dG, *ddu, *dGd, *dal;
  RET(take(&dal, (size_t)Mp * D, true));
  if (Mp == 256 && !ctx->gram_ws) RET(dev_alloc(ctx, &ctx->gram_ws, gemm_gram_ws_bytes(ctx->cu_count) / 8));
  RET(take(&dGd, (size_t)D * MM, true));
  RET(take(&dKt, (size_t)Pm * Mp, true)); RET(take(&dLinv, MM, false)); RET(take(&dLinvT, MM, false));
  RET(take(&dW, (size_t)Mp * DM, false)); RET(take(&dS, (size_t)Mp * DM, false)); RET(take(&du_, (size_t)Mp * D, false));
  RET(take(&dvb, (size_t)Pm * D, true)); RET(take(&dmb, (size_t)Pm * D, true));
  RET(take(&dCt, (size_t)Pm * Mp, true)); RET(take(&cnp, (size_t)Pm * maxpl, true)); RET(take(&tnp, (size_t)Pm * maxpl * D, true));
  RET(take(&dT, (size_t)Pm * DM, true)); RET(take(&dm0, (size_t)Pm * D, true)); RET(take(&dCb, (size_t)Pm * Mp, true));
  RET(take(&dKb, (size_t)Pm * Mp, true)); RET(take(&dG, (size_t)Pm * Mp, true)); RET(take(&ddu, (size_t)Mp * D, true));
  HIPCHK(hipMemcpyAsync(dKt, Kt, (size_t)P * Mp * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dLinv, Linv, MM * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dW, Wcat, (size_t)Mp * DM * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(du_, u, (size_t)Mp * D * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dvb, vbar, (size_t)P * D * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dmb, mbar, (size_t)P * D * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(wcat_transpose(ctx->st, dLinv, Mp, 1, dLinvT));         // as prep() does
  HIPCHK(wcat_transpose(ctx->st, dW, Mp, D, dS));                // Scat = W_d^T stacked (stored-T form)
  // forward (forward_chunk)
  const bool with_alpha = Mp == 256 && D <= 8;          // as dgp_set_layers allocates it
  if (with_alpha) RET(G(ctx, 2, GEMM_NN, Mp, D, Mp, dLinvT, Mp, du_, D, dal, D, 1.0, 0));      // alpha = LinvT u, as prep() does
  GemmArgs aC = args_Ct(Pm, Mp, dKt, dLinvT, dCt, cnp, with_alpha ? dal : nullptr, dm0, D);
  GemmArgs aT = args_T(Pm, Mp, D, dCt, dW, dT, tnp, t_blocked(ctx, Pm, Mp, D));
  int nplane = 0, nplane_t = 0;
  bool mean_done = false;
  RET(launch_Ct_T(ctx, aC, aT, dLinv, P, &nplane, &nplane_t, &mean_done));
  if (!mean_done) RET(GX(ctx, 0, GEMM_NN, args_mean0(Pm, Mp, D, dCt, du_, dm0)));
  // backward (backward_chunk, stored-T form)
  GemmArgs aB = args_Cbar(Pm, Mp, D, dT, dS, dCb, dvb, dCt, dmb, du_, t_blocked(ctx, Pm, Mp, D));
  RET(GX(ctx, 0, GEMM_NN, aB));
  GemmArgs aG = args_g(Pm, Mp, dCb, dLinv, dKb, dKt, dG);
  RET(GX(ctx, 0, GEMM_NN, aG));
  // du rides on G_d's launch when that one goes to the Gram kernel (backward_chunk)
  GemmArgs aU = args_du(ctx, P, Mp, D, dCt, dmb, ddu);
  GemmArgs aGd = args_G(ctx, P, Mp, D, dCt, dvb, dGd);
  const bool du_in_gram = ctx->du_in_gram && gemm_engine_of(GEMM_TN, aGd) == 4;
  if (du_in_gram) { aGd.gram_mb = dmb; aGd.gram_du = ddu; }
  else RET(GX(ctx, 0, GEMM_TN, aU));
  RET(GX(ctx, 0, GEMM_TN, aGd));
  if (engines) {
    GemmArgs c2 = aC, t2 = aT;
    int pc, pt;
    plan_Ct_T(ctx, c2, t2, dLinv, &pc, &pt);
    engines[0] = gemm_engine_of(c2.no_wide ? GEMM_NT : GEMM_NN, c2);
    engines[1] = gemm_engine_of(GEMM_NN, t2);
    engines[2] = gemm_engine_of(GEMM_NN, aB);
    engines[3] = gemm_engine_of(GEMM_NN, aG);
    engines[5] = gemm_engine_of(GEMM_TN, aGd);
    engines[6] = mean_done ? 1 : 0;
    engines[4] = du_in_gram ? engines[5] : gemm_engine_of(GEMM_TN, aU);
  }
  // results: row-norm planes summed as finalize_layer sums them, T taken out of the engine's blocked layout
  std::vector<double> hpl((size_t)Pm * maxpl * D), hT((size_t)Pm * DM);
  HIPCHK(hipMemcpyAsync(Ct, dCt, (size_t)P * Mp * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(mean0, dm0, (size_t)P * D * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(Cbar, dCb, (size_t)P * Mp * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(g, dG, (size_t)P * Mp * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(du, ddu, (size_t)Mp * D * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(Gd, dGd, (size_t)D * MM * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(hT.data(), dT, (size_t)Pm * DM * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(hpl.data(), cnp, (size_t)Pm * nplane * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  for (long p = 0; p < P; ++p) {
    double s = 0.0;
    for (int q = 0; q < nplane; ++q) s += hpl[(size_t)q * Pm + p];
    cn[p] = s;
  }
  HIPCHK(hipMemcpy(hpl.data(), tnp, (size_t)Pm * nplane_t * D * 8, hipMemcpyDeviceToHost));
  for (long p = 0; p < P; ++p)
    for (int d = 0; d < D; ++d) {
      double s = 0.0;
      for (int q = 0; q < nplane_t; ++q) s += hpl[((size_t)d * nplane_t + q) * Pm + p];
      tn[p * D + d] = s;
    }
  if (aT.c_blocked) {      // gemm_f64.h: panels of 128 rows, inside a panel blocks of 16 columns, each block 128 x 16 row-major
    for (long p = 0; p < P; ++p) {
      const double* panel = hT.data() + (p / 128) * 128 * DM;
      const long r = p % 128;
      for (long n = 0; n < DM; ++n) T[p * DM + n] = panel[(n / 16) * (128 * 16) + r * 16 + (n % 16)];
    }
  } else {
    memcpy(T, hT.data(), (size_t)P * DM * 8);
  }
  return DGP_OK;
}

int dgp_dev_rbf_contract(dgp_ctx* ctx, const double* G, const double* Z1, const double* X1, int64_t Pn, int32_t Mp, int32_t w1,
                         double* R1, double* GX, int32_t* fused) {
  if (!ctx || !G || !Z1 || !X1 || !R1 || !GX || Pn <= 0 || Mp <= 0 || Mp % 64 != 0 || w1 < 1 || w1 > 64)
    return fail(ctx, DGP_ERR_INVALID, "dgp_dev_rbf_contract: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  const long P = Pn, Pm = pad_rows(P);
  double *dG = nullptr, *dZ = nullptr, *dX = nullptr, *dR = nullptr, *dGX = nullptr;
  struct Free { double **a, **b, **c, **d, **e; ~Free() { dev_free(*a); dev_free(*b); dev_free(*c); dev_free(*d); dev_free(*e); } } freer{&dG, &dZ, &dX, &dR, &dGX};
  RET(dev_alloc(ctx, &dG, (size_t)Pm * Mp)); RET(dev_alloc(ctx, &dZ, (size_t)Mp * w1)); RET(dev_alloc(ctx, &dX, (size_t)Pm * w1));
  RET(dev_alloc(ctx, &dR, (size_t)Pm * w1)); RET(dev_alloc(ctx, &dGX, (size_t)Mp * w1));
  HIPCHK(hipMemsetAsync(dG, 0, (size_t)Pm * Mp * 8, ctx->st));
  HIPCHK(hipMemsetAsync(dX, 0, (size_t)Pm * w1 * 8, ctx->st));
  HIPCHK(hipMemcpyAsync(dG, G, (size_t)P * Mp * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dZ, Z1, (size_t)Mp * w1 * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dX, X1, (size_t)P * w1 * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dGX, GX, (size_t)Mp * w1 * 8, hipMemcpyHostToDevice, ctx->st));
  int f = 0;
  RET(launch_rbf_contract(ctx, dG, P, Pm, Mp, dZ, dX, w1, dR, dGX, &f));      // the call backward_chunk makes
  if (fused) *fused = f;
  HIPCHK(hipMemcpyAsync(R1, dR, (size_t)P * w1 * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(GX, dGX, (size_t)Mp * w1 * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  return DGP_OK;
}

int dgp_dev_g_panel(dgp_ctx* ctx, const double* Cbar, const double* Linv, const double* E, const double* Z1, const double* X1,
                    int64_t Pn, int32_t w1, double* R1, double* GX, int32_t* used) {
  if (!ctx || !Cbar || !Linv || !E || !Z1 || !X1 || !R1 || !GX || Pn <= 0 || w1 < 2 || w1 > 9)
    return fail(ctx, DGP_ERR_INVALID, "dgp_dev_g_panel: bad arguments (Mp = 256, 2 <= w1 <= 9)");
  HIPCHK(hipSetDevice(ctx->device));
  const long P = Pn, Mp = 256;
  const bool ok = g_panel_ok((int)Mp, w1, P);
  if (used) *used = ok ? 1 : 0;
  if (!ok) return DGP_OK;                       // (the backward pass would take the stored-g path: dgp_dev_layer_products + dgp_dev_rbf_contract)
  double *dC = nullptr, *dL = nullptr, *dE = nullptr, *dZ = nullptr, *dX = nullptr, *dR = nullptr, *dGX = nullptr;
  struct Free { double** p[7]; ~Free() { for (auto q : p) dev_free(*q); } } freer{{&dC, &dL, &dE, &dZ, &dX, &dR, &dGX}};
  // exactly P rows: the kernel must not touch anything behind them (rows of the last panel past P read as zeros)
  RET(dev_alloc(ctx, &dC, (size_t)P * Mp)); RET(dev_alloc(ctx, &dL, (size_t)Mp * Mp)); RET(dev_alloc(ctx, &dE, (size_t)P * Mp));
  RET(dev_alloc(ctx, &dZ, (size_t)Mp * w1)); RET(dev_alloc(ctx, &dX, (size_t)P * w1)); RET(dev_alloc(ctx, &dR, (size_t)P * w1));
  RET(dev_alloc(ctx, &dGX, (size_t)Mp * w1));
  HIPCHK(hipMemcpyAsync(dC, Cbar, (size_t)P * Mp * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dL, Linv, (size_t)Mp * Mp * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dE, E, (size_t)P * Mp * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dZ, Z1, (size_t)Mp * w1 * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dX, X1, (size_t)P * w1 * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(dGX, GX, (size_t)Mp * w1 * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemsetAsync(dR, 0xff, (size_t)P * w1 * 8, ctx->st));
  if (!ctx->rg_ws) RET(dev_alloc(ctx, &ctx->rg_ws, (size_t)rbf_bwd_contract_ws_doubles()));
  HIPCHK(g_panel(ctx->st, dC, dL, dE, P, dZ, dX, w1, dR, dGX, ctx->rg_ws, ctx->cu_count, ctx->reserved_cus));      // the call backward_chunk makes
  HIPCHK(hipMemcpyAsync(R1, dR, (size_t)P * w1 * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(GX, dGX, (size_t)Mp * w1 * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  return DGP_OK;
}

int dgp_dev_chol(dgp_ctx* ctx, double* A, int32_t M, int32_t batch) {
  if (!ctx || !A || M <= 0 || M % 16 != 0 || batch <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_dev_chol: M must be a multiple of 16");
  double* d;
  const size_t n = (size_t)batch * M * M;
  RET(dev_alloc(ctx, &d, n));
  HIPCHK(hipMemcpy(d, A, n * 8, hipMemcpyHostToDevice));
  double *x, *t;
  RET(dev_alloc(ctx, &x, n)); RET(dev_alloc(ctx, &t, n));
  HIPCHK(potrf_inv(ctx->st, d, x, t, M, batch, ctx->info));
  HIPCHK(hipStreamSynchronize(ctx->st));
  HIPCHK(hipMemcpy(A, d, n * 8, hipMemcpyDeviceToHost));
  dev_free(d); dev_free(x); dev_free(t);
  return check_flags(ctx);
}

int dgp_dev_trinv(dgp_ctx* ctx, const double* L, double* X, int32_t M, int32_t batch) {
  if (!ctx || !L || !X || M <= 0 || batch <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_dev_trinv: bad arguments");
  if (M % 16 != 0) return fail(ctx, DGP_ERR_INVALID, "dgp_dev_trinv: M must be a multiple of 16");
  double *dl, *dx, *dt;
  const size_t n = (size_t)batch * M * M;
  RET(dev_alloc(ctx, &dl, n)); RET(dev_alloc(ctx, &dx, n)); RET(dev_alloc(ctx, &dt, n));
  HIPCHK(hipMemcpy(dl, L, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dx, 0, n * 8));
  HIPCHK(trinv_lower(ctx->st, dl, dx, dt, M, batch));
  HIPCHK(hipStreamSynchronize(ctx->st));
  HIPCHK(hipMemcpy(X, dx, n * 8, hipMemcpyDeviceToHost));
  dev_free(dl); dev_free(dx); dev_free(dt);
  return DGP_OK;
}

}  // extern "C"


extern "C" {

int dgp_dev_normals(dgp_ctx* ctx, uint64_t seed, int32_t layer, int32_t S, int64_t n0, int64_t N, int32_t D, double* z_out) {
  if (!ctx || !z_out || S <= 0 || N <= 0 || D <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_dev_normals: bad arguments");
  double* d;
  const size_t n = (size_t)S * N * D;
  RET(dev_alloc(ctx, &d, n));
  ZSource z;
  z.zs = nullptr; z.seed = seed; z.layer = layer; z.n_global0 = n0; z.Ntot = N;
  HIPCHK(launch_normals(ctx->st, z, S, (long)N, D, d));
  HIPCHK(hipStreamSynchronize(ctx->st));
  HIPCHK(hipMemcpy(z_out, d, n * 8, hipMemcpyDeviceToHost));
  dev_free(d);
  return DGP_OK;
}

int dgp_dev_mfma_peak(dgp_ctx* ctx, int32_t iters, double* tflops_out) {
  if (!ctx || iters <= 0 || !tflops_out) return fail(ctx, DGP_ERR_INVALID, "dgp_dev_mfma_peak: bad arguments");
  int cus = 256;
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, ctx->device));
  cus = p.multiProcessorCount;
  double* sink;
  const int blocks = cus * 4;
  RET(dev_alloc(ctx, &sink, (size_t)blocks * 256));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(launch_mfma_peak(ctx->st, blocks, iters, sink));   // warm-up
  HIPCHK(hipEventRecord(e0, ctx->st));
  HIPCHK(launch_mfma_peak(ctx->st, blocks, iters, sink));
  HIPCHK(hipEventRecord(e1, ctx->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  // per wave: iters * 8 MFMAs of 16x16x4 (2*16*16*4 flops each); 4 waves per block
  const double flops = (double)blocks * 4.0 * iters * 8.0 * 2048.0;
  *tflops_out = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  dev_free(sink);
  return DGP_OK;
}

}  // extern "C"
