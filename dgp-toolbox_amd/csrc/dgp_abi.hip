// libdgp_hip.so: context, orchestration of the SVGP-layer ELBO path and the C-ABI (include/dgp_abi.h).
//
// Reference path being replaced (all executed by TensorFlow/GPflow there):
//   DGP_Base.propagate / ELBO            dgp_dace/models/dgp.py:34-109
//   SVGP_Layer.conditional_ND / KL        dgp_dace/utils/layers.py:227-308
//   Layer.sample_from_conditional         dgp_dace/utils/layers.py:87-130, utils.py:22-51
//   DGP.optimize_adam / optimize_nat_adam dgp_dace/models/dgp.py:255-345 (loop bodies)
//
// Whitened formulation used on the device (identical in exact arithmetic to layers.py:243-276):
//   Lu = chol(K(Z,Z) + 1e-6 I),  c_p = Lu^-1 k(Z, x_p),  W_d = Lu^-1 L_q,d (non-white) or L_q,d (white),
//   u = Lu^-1 q_mu (non-white) or q_mu,   mean = c^T u + mf(x),   var = k_diag - |c|^2 + |W_d^T c|^2.
// Data layout in HBM: every per-point intermediate is point-major ([points][M] / [points][D][M]), so
// a tile of points is one contiguous slab; a chunk of data points (all S samples of each point) goes
// through forward-all-layers then backward-all-layers with its intermediates resident in HBM.
#include "../../include/dgp_abi.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dgp_internal.h"

using namespace dgp;

namespace {

struct Layer {
  dgp_layer_desc d;
  int Mp;
  long off_Z, off_var, off_ls, off_qmu, off_qsqrt;   // offsets in the flat parameter vector
  double *meanW = nullptr, *meanb = nullptr;
  double *Kuu, *Lu, *Linv, *Lq, *qmu_p, *Wcat, *u;   // derived small matrices (padded to Mp)
  double *Scat;                                      // [D*Mp x Mp] (backward only): W_d^T stacked when t_d is kept from the
                                                     // forward pass (ctx->store_t), else S'_d = W_d W_d^T - I stacked
  double *Tt = nullptr;                              // [points][D*Mp]: t_d = W_d^T c (training chunks, ctx->store_t)
  double *Et = nullptr;                              // [points][Mp]: -2 dk/d(r2) of Kuf (training chunks, Matern kernels)
  double *Euu = nullptr;                             // [Mp x Mp]: the same for Kuu (Matern kernels)
  double *kdot = nullptr;                            // device scalar: sum_p dk_p . k_p (Matern kernels)
  double *Z1;                                        // [Mp x (D_in+1)] = [Z | 1]
  double *dLq, *dqmu_p;                              // d ELBO / d (Lq, q_mu) of the last grad_finish
  long acc_Q, acc_G, acc_du, acc_GX, acc_x2, acc_dvar;
  double *Kt, *Ct, *cnp, *tnp, *mean0, *mean, *var, *F, *mbar, *vbar;   // chunk workspace
};

constexpr int kNCat = 4;
struct Prof {
  bool on = false;
  int depth = 0;
  std::vector<hipEvent_t> ev;
  std::vector<int> cat;
  size_t used = 0;
  double ms[kNCat] = {0, 0, 0, 0}, flops[kNCat] = {0, 0, 0, 0}, bytes[kNCat] = {0, 0, 0, 0};
  long launches[kNCat] = {0, 0, 0, 0};
};

}  // namespace

struct dgp_ctx {
  int device = 0;
  int cu_count = 0;
  hipStream_t st = nullptr;
  bool own_stream = false;
  std::string err;
  std::vector<Layer> L;
  long n_params = 0;
  double *params = nullptr, *grad = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  long adam_t = 0;
  std::vector<ParamSeg> segs;
  ParamSeg* segs_dev = nullptr;
  double *mean_params = nullptr;
  double *X = nullptr, *Y = nullptr;
  long N = 0;
  int D = 0, Dy = 0;
  long n_goff = 0;
  double *acc = nullptr, *acc_own = nullptr;
  long n_acc = 0;
  double* scal = nullptr;   // device scalars: [0] sum KL, [1] ELBO of last grad_finish, [2] scratch data term
  int* info = nullptr;
  char* ws = nullptr;
  size_t ws_cap = 0;
  long ws_limit = 96L << 30;
  double *Cbar = nullptr, *Kbar = nullptr, *xbar = nullptr, *Gt = nullptr, *X1 = nullptr, *R1 = nullptr;
  // small-matrix scratch, one set per side stream: the per-layer chains (prep, gradient finish, natural-gradient
  // step) are independent of each other and run concurrently, one layer per side stream (LayerFork below)
  static constexpr int kSide = 3;
  double* smset[kSide][10] = {{nullptr}};
  double** sm = smset[0];
  hipStream_t side[kSide] = {nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[kSide] = {nullptr};
  bool use_side = true;
  // Option (DGP_STORE_T=1): keep t_d = W_d^T c from the forward pass (16 KB/point/layer at D=8, M=256), so that dC
  // needs only the triangular products W_d t_d instead of the dense S'_d c.  Measured at config 2: the dC product
  // 19.2 -> 15.7 ms (it now streams 16 GB, re-read per column tile), the T-product 12.0 -> 15.1 ms (the 16 GB store):
  // no net gain, +32 GB of HBM traffic per iteration -> off by default.
  bool store_t = false;
  std::vector<double*> zs_dev;
  std::vector<size_t> zs_cap;
  double* Xnew = nullptr;
  size_t Xnew_cap = 0;
  std::vector<double*> out_dev[3];
  std::vector<size_t> out_cap[3];
  bool grad_ready = false;
  bool segs_uploaded = false;
  Prof prof;
};

namespace {

int fail(dgp_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof buf, "%s", what);
  if (ctx) ctx->err = buf;
  return code;
}

#define HIPCHK(x)                                                   \
  do {                                                              \
    hipError_t e_ = (x);                                            \
    if (e_ != hipSuccess) return fail(ctx, DGP_ERR_HIP, #x, e_);    \
  } while (0)
#define RET(x)                  \
  do {                          \
    int r_ = (x);               \
    if (r_ != DGP_OK) return r_; \
  } while (0)

inline long round_up(long x, long m) { return ((x + m - 1) / m) * m; }

// ------------------------------------------------------------------------------- profiling helpers
int prof_drain(dgp_ctx* ctx) {
  Prof& p = ctx->prof;
  if (p.used == 0) return DGP_OK;
  HIPCHK(hipStreamSynchronize(ctx->st));
  for (size_t i = 0; i < p.used; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
    p.ms[p.cat[i]] += ms;
  }
  p.used = 0;
  return DGP_OK;
}

struct ProfScope {
  dgp_ctx* ctx;
  bool active;
  size_t slot;
  ProfScope(dgp_ctx* c, int cat, double flops, double bytes) : ctx(c), active(false), slot(0) {
    Prof& p = c->prof;
    if (!p.on) return;
    p.flops[cat] += flops;
    p.bytes[cat] += bytes;
    if (p.depth++ > 0) return;          // nested scopes are covered by the outermost one
    active = true;
    if (p.used * 2 + 2 > p.ev.size()) {
      if (p.ev.size() < 2 * 8192) {
        const size_t old = p.ev.size();
        p.ev.resize(old + 1024);
        for (size_t i = old; i < p.ev.size(); ++i) (void)hipEventCreate(&p.ev[i]);
        p.cat.resize(p.ev.size() / 2);
      } else {
        prof_drain(c);
      }
    }
    slot = p.used++;
    p.cat[slot] = cat;
    p.launches[cat] += 1;
    (void)hipEventRecord(p.ev[2 * slot], c->st);
  }
  ~ProfScope() {
    Prof& p = ctx->prof;
    if (!p.on) return;
    if (active) (void)hipEventRecord(p.ev[2 * slot + 1], ctx->st);
    if (p.depth > 0) --p.depth;
  }
};

// ------------------------------------------------------------------------------- GEMM wrapper
int G(dgp_ctx* ctx, int cat, GemmOp op, long M, long N, long K, const double* A, long lda, const double* B, long ldb,
      double* C, long ldc, double alpha, int beta, int batch = 1, long sA = 0, long sB = 0, long sC = 0,
      int splits = 1, int tri = TRI_NONE, long triblk = 0, double flops = 0.0, double bytes = 0.0) {
  GemmArgs a;
  a.A = A; a.B = B; a.C = C;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.M = M; a.N = N; a.K = K;
  a.sA = sA; a.sB = sB; a.sC = sC;
  a.batch = batch; a.splits = splits; a.ksplit = 0;
  a.alpha = alpha; a.beta = beta; a.tri = tri; a.triblk = triblk;
  if (M <= 0 || N <= 0 || K <= 0) return DGP_OK;
  ProfScope ps(ctx, cat, flops, bytes);
  HIPCHK(gemm_f64(ctx->st, op, a));
  return DGP_OK;
}

int GX(dgp_ctx* ctx, int cat, GemmOp op, GemmArgs a, double flops = 0.0, double bytes = 0.0) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return DGP_OK;
  ProfScope ps(ctx, cat, flops, bytes);
  HIPCHK(gemm_f64(ctx->st, op, a));
  return DGP_OK;
}

GemmArgs mk(long M, long N, long K, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
            double alpha = 1.0, int beta = 0) {
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.sA = a.sB = a.sC = 0; a.batch = 1; a.splits = 1; a.ksplit = 0; a.alpha = alpha; a.beta = beta;
  a.tri = TRI_NONE; a.triblk = 0;
  return a;
}

// Runs independent per-layer chains of small kernels concurrently on three workers: the context's own stream and
// two side streams (the runtime gives this process few hardware queues: with three side streams two of them shared
// one, and raising GPU_MAX_HW_QUEUES slowed every launch).  Item i goes to worker i % 3 with that worker's scratch
// set; the destructor joins everything back into the context's stream (also on error returns).
struct LayerFork {
  dgp_ctx* ctx;
  hipStream_t main;
  bool on;
  ProfScope ps;
  int used = 0;
  LayerFork(dgp_ctx* c, int n_items) : ctx(c), main(c->st), on(c->use_side && n_items > 1), ps(c, 2, 0, 0) {
    if (!on) return;
    used = n_items - 1 < dgp_ctx::kSide - 1 ? n_items - 1 : dgp_ctx::kSide - 1;     // side streams in use
    (void)hipEventRecord(c->ev_fork, main);
    for (int i = 0; i < used; ++i) (void)hipStreamWaitEvent(c->side[i], c->ev_fork, 0);
  }
  void use(int i) {
    if (!on) return;
    const int w = i % dgp_ctx::kSide;
    ctx->st = w == 0 ? main : ctx->side[w - 1];
    ctx->sm = ctx->smset[w];
  }
  ~LayerFork() {
    if (!on) return;
    ctx->st = main;
    ctx->sm = ctx->smset[0];
    for (int i = 0; i < used; ++i) {
      (void)hipEventRecord(ctx->ev_join[i], ctx->side[i]);
      (void)hipStreamWaitEvent(main, ctx->ev_join[i], 0);
    }
  }
};

// Split-K factor for the reductions over points: all active workgroups of such a launch run equally long, so
// their number should fill whole rounds of the resident slots (2 workgroups per CU); `active_tiles` counts the
// output tiles that survive the triangular skip, times the batch.
int pick_splits_tiles(dgp_ctx* ctx, long active_tiles, long K, long row_bytes = 0, long min_rows = 2048) {
  // Chunk of points per split sized so that the streamed rows of one chunk (row_bytes each) stay in one XCD's
  // 4 MiB L2 while all members of the chunk read them (see gemm_f64.h: XCD-grouped split-K mapping); the number
  // of splits is a multiple of 8 (one group per XCD at a time).
  // The members advance through their chunk in step, so L2 only has to hold the window between the fastest and
  // the slowest member, not the whole chunk: a chunk of a few MiB keeps the hit rate while bounding the atomics.
  if (row_bytes <= 0) row_bytes = 2048;
  static long chunk_bytes = 0;
  if (chunk_bytes == 0) {
    const char* e = getenv("DGP_GRAM_CHUNK_BYTES");
    chunk_bytes = e ? atol(e) : (8L << 20);
    if (chunk_bytes < (1L << 16)) chunk_bytes = 1L << 16;
  }
  long chunk = chunk_bytes / row_bytes;
  chunk = (chunk / 16) * 16;
  if (chunk < 256) chunk = 256;
  long s = (K + chunk - 1) / chunk;
  // a launch of few members (skinny outputs, small shards of a multi-GPU run) still has to fill the chip: enough
  // splits for ~2 workgroups per CU, as long as a split keeps >= min_rows of K (the atomics of a split cost a
  // full tile of traffic: 2048 rows for 128x64 tiles, 256 for the skinny 128x16 ones)
  const long fill = (2L * (ctx->cu_count > 0 ? ctx->cu_count : 256) + active_tiles - 1) / (active_tiles > 0 ? active_tiles : 1);
  const long cap = K / min_rows;
  if (s < fill) s = fill < cap ? fill : cap;
  s = ((s + 7) / 8) * 8;
  if (s * 16 > K) s = 1;                               // tiny reductions: no split
  if (s > 32760) s = 32760;
  return (int)s;
}

// number of 128x64 output tiles of an [Mp x Mp] lower-triangular target that are not skipped
long lower_tiles(long Mp) {
  long n = 0;
  for (long m0 = 0; m0 < Mp; m0 += 128)
    for (long n0 = 0; n0 < Mp; n0 += 64)
      if (m0 + 127 >= n0) ++n;
  return n;
}

int pick_splits(dgp_ctx* ctx, long Mrows, long Ncols, long K) {
  const long tiles = ((Mrows + 127) / 128) * ((Ncols + 63) / 64);
  return pick_splits_tiles(ctx, tiles, K, Mrows * 8, Ncols <= 16 ? 256 : 2048);
}

// ------------------------------------------------------------------------------- memory helpers
template <typename T>
int dev_alloc(dgp_ctx* ctx, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return DGP_OK;
}
template <typename T>
void dev_free(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}
int grow(dgp_ctx* ctx, double** p, size_t* cap, size_t n) {
  if (*cap >= n && *p) return DGP_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  RET(dev_alloc(ctx, p, n));
  *cap = n;
  return DGP_OK;
}

void free_model(dgp_ctx* ctx) {
  for (auto& l : ctx->L) {
    dev_free(l.Kuu); dev_free(l.Lu); dev_free(l.Linv); dev_free(l.Lq); dev_free(l.qmu_p); dev_free(l.Wcat);
    dev_free(l.u); dev_free(l.Scat); dev_free(l.Z1); dev_free(l.Euu); dev_free(l.kdot); dev_free(l.dLq); dev_free(l.dqmu_p);
  }
  ctx->L.clear();
  dev_free(ctx->params); dev_free(ctx->grad); dev_free(ctx->adam_m); dev_free(ctx->adam_v);
  dev_free(ctx->segs_dev); dev_free(ctx->mean_params); dev_free(ctx->acc_own);
  ctx->acc = nullptr;
  for (auto& set : ctx->smset) for (auto& s : set) dev_free(s);
  for (auto& z : ctx->zs_dev) dev_free(z);
  ctx->zs_dev.clear(); ctx->zs_cap.clear();
  for (int k = 0; k < 3; ++k) {
    for (auto& o : ctx->out_dev[k]) dev_free(o);
    ctx->out_dev[k].clear(); ctx->out_cap[k].clear();
  }
  ctx->n_params = 0;
  ctx->grad_ready = false;
  ctx->segs_uploaded = false;
}

inline long pad_rows(long P) { return round_up(P, 128); }

// carve the chunk workspace; returns bytes needed.  base == nullptr: size query only
size_t carve(dgp_ctx* ctx, char* base, long Nc, int S, bool train) {
  size_t off = 0;
  auto take = [&](long nd) -> double* {
    double* p = base ? reinterpret_cast<double*>(base + off) : nullptr;
    off += (size_t)round_up(nd * 8, 256);
    return p;
  };
  const int nl = (int)ctx->L.size();
  // Row counts of everything a row-parallel GEMM reads or writes are rounded up to the GEMM row tile (pad_rows):
  // the products then run on whole tiles (no predicated edge strip: a 128-row strip of a K = D*Mp product is a
  // 0.25 ms serial tail at multi-GPU shard sizes).  The pad rows hold garbage that never enters a reduction
  // (those run over exactly Pl rows).
  long Pmax_Mp = 0, pl_max = 0, xb_max = 0, x1_max = 0;
  for (int l = 0; l < nl; ++l) {
    Layer& y = ctx->L[l];
    const long Pl = (l == 0) ? Nc : (long)S * Nc, Pm = pad_rows(Pl);
    Pmax_Mp = std::max(Pmax_Mp, Pm * y.Mp);
    pl_max = std::max(pl_max, Pm * (y.Mp / 32) * (1 + y.d.D_out));
    xb_max = std::max(xb_max, (long)S * Nc * y.d.D_in);
    x1_max = std::max(x1_max, Pm * (y.d.D_in + 1));
  }
  double *sKt = nullptr, *sCt = nullptr, *sPl = nullptr;
  if (!train) { sKt = take(Pmax_Mp); sCt = take(Pmax_Mp); }
  sPl = take(pl_max);                                      // row-norm partial planes: consumed within the layer
  for (int l = 0; l < nl; ++l) {
    Layer& y = ctx->L[l];
    const long Pl = (l == 0) ? Nc : (long)S * Nc, Pm = pad_rows(Pl);
    const long D = y.d.D_out;
    if (train) { y.Kt = take(Pm * y.Mp); y.Ct = take(Pm * y.Mp); }
    else { y.Kt = sKt; y.Ct = sCt; }
    y.cnp = sPl;
    y.tnp = sPl ? sPl + Pm * (y.Mp / 32) : nullptr;
    y.mean0 = take(Pm * D); y.mean = take(Pl * D); y.var = take(Pl * D);
    y.F = take((long)S * Nc * D);
    if (train) { y.mbar = take(Pm * D); y.vbar = take(Pm * D); }
    else { y.mbar = y.vbar = nullptr; }
    y.Tt = (train && ctx->store_t) ? take(Pm * D * y.Mp) : nullptr;
    y.Et = (train && y.d.kernel_kind != DGP_KERNEL_RBF) ? take(Pm * y.Mp) : nullptr;
  }
  if (train) {
    ctx->Cbar = take(Pmax_Mp); ctx->Kbar = take(Pmax_Mp); ctx->Gt = take(Pmax_Mp); ctx->xbar = take(xb_max);
    ctx->X1 = take(x1_max); ctx->R1 = take(x1_max);
  }
  return off;
}

int ensure_ws(dgp_ctx* ctx, long N, int S, bool train, long* Nc_out) {
  const size_t per1 = carve(ctx, nullptr, 1024, S, train) / 1024 + 1;
  long Nc = (long)((size_t)ctx->ws_limit / per1);
  if (Nc < 1) Nc = 1;
  if (Nc > N) Nc = N;
  size_t need = carve(ctx, nullptr, Nc, S, train);
  while (need > (size_t)ctx->ws_limit && Nc > 1) { Nc = Nc * 9 / 10; need = carve(ctx, nullptr, Nc, S, train); }
  if (need > ctx->ws_cap) {
    if (ctx->ws) (void)hipFree(ctx->ws);
    ctx->ws = nullptr; ctx->ws_cap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&ctx->ws), need));
    ctx->ws_cap = need;
  }
  carve(ctx, ctx->ws, Nc, S, train);
  *Nc_out = Nc;
  return DGP_OK;
}

int upload_zs(dgp_ctx* ctx, const double* const* zs, int S, long Ntot) {
  const int nl = (int)ctx->L.size();
  if ((int)ctx->zs_dev.size() < nl) { ctx->zs_dev.resize(nl, nullptr); ctx->zs_cap.resize(nl, 0); }
  for (int l = 0; l < nl; ++l) {
    if (!zs[l]) return fail(ctx, DGP_ERR_INVALID, "zs: NULL entry");
    const size_t n = (size_t)S * Ntot * ctx->L[l].d.D_out;
    RET(grow(ctx, &ctx->zs_dev[l], &ctx->zs_cap[l], n));
    HIPCHK(hipMemcpyAsync(ctx->zs_dev[l], zs[l], n * 8, hipMemcpyHostToDevice, ctx->st));
  }
  return DGP_OK;
}

inline const double* P(dgp_ctx* ctx, long off) { return ctx->params + off; }

// ------------------------------------------------------------------------------- prep: small matrices + KL
int prep(dgp_ctx* ctx, bool train = false) {
  HIPCHK(hipMemsetAsync(ctx->scal, 0, 4 * sizeof(double), ctx->st));
  LayerFork fork(ctx, (int)ctx->L.size());
  for (size_t li = 0; li < ctx->L.size(); ++li) {
    Layer& y = ctx->L[li];
    fork.use((int)li);
    const int M = y.d.M, Mp = y.Mp, D = y.d.D_out, Din = y.d.D_in;
    const long MM = (long)Mp * Mp;
    HIPCHK(pack_q(ctx->st, P(ctx, y.off_qsqrt), P(ctx, y.off_qmu), M, Mp, D, y.Lq, y.qmu_p));
    HIPCHK(rbf_kuu(ctx->st, y.d.kernel_kind, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), M, Mp, Din, y.Kuu,
                   train ? y.Euu : nullptr));
    HIPCHK(copy_mat(ctx->st, y.Kuu, y.Lu, MM));
    HIPCHK(potrf_inv(ctx->st, y.Lu, y.Linv, ctx->sm[9], Mp, 1, ctx->info));
    if (y.d.white) {
      HIPCHK(lq_to_wcat(ctx->st, y.Lq, Mp, D, y.Wcat));
      HIPCHK(copy_mat(ctx->st, y.qmu_p, y.u, (long)Mp * D));
    } else {
      RET(G(ctx, 2, GEMM_NN, Mp, Mp, Mp, y.Linv, Mp, y.Lq, Mp, y.Wcat, (long)D * Mp, 1.0, 0, D, 0, MM, Mp));
      RET(G(ctx, 2, GEMM_NN, Mp, D, Mp, y.Linv, Mp, y.qmu_p, D, y.u, D, 1.0, 0));
    }
    HIPCHK(layer_kl(ctx->st, y.Wcat, y.u, y.Lq, y.Lu, M, Mp, D, y.d.white, ctx->scal));
    if (train) {
      if (ctx->store_t) {
        HIPCHK(wcat_transpose(ctx->st, y.Wcat, Mp, D, y.Scat));
      } else {       // S'_d = W_d W_d^T - I  (symmetric), stacked [D*Mp x Mp]
        GemmArgs a = mk(Mp, Mp, Mp, y.Wcat, (long)D * Mp, y.Wcat, (long)D * Mp, y.Scat, Mp);
        a.batch = D; a.sA = Mp; a.sB = Mp; a.sC = MM;
        RET(GX(ctx, 2, GEMM_NT, a));
        HIPCHK(sub_identity(ctx->st, y.Scat, M, Mp, D));
      }
      HIPCHK(make_z1(ctx->st, P(ctx, y.off_Z), M, Mp, Din, y.Z1));
    }
  }
  return DGP_OK;
}

ZSource zsrc_of(dgp_ctx* ctx, int l, bool use_zs, uint64_t seed, long n_goff, long Ntot) {
  ZSource z;
  z.zs = use_zs ? ctx->zs_dev[l] : nullptr;
  z.seed = seed;
  z.layer = l;
  z.n_global0 = n_goff;
  z.Ntot = Ntot;
  return z;
}

// ------------------------------------------------------------------------------- forward over one chunk
int forward_chunk(dgp_ctx* ctx, const double* Xsrc, long Ntot, long n0, long Nc, int S, uint64_t seed, bool use_zs,
                  long n_goff) {
  const int nl = (int)ctx->L.size();
  for (int l = 0; l < nl; ++l) {
    Layer& y = ctx->L[l];
    const int M = y.d.M, Mp = y.Mp, D = y.d.D_out, Din = y.d.D_in;
    const bool dedup = (l == 0);
    const long Pl = dedup ? Nc : (long)S * Nc, Pm = pad_rows(Pl);
    const double* Xin = dedup ? Xsrc : ctx->L[l - 1].F;
    const long row0 = dedup ? n0 : 0;
    const int nplane = Mp / 32;
    {
      ProfScope ps(ctx, 1, 0, (double)Pl * (Mp + Din) * 8);
      HIPCHK(rbf_kuf(ctx->st, y.d.kernel_kind, Xin, Pl, row0, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), M, Mp, Din,
                     y.Kt, y.Et));
    }
    const double tri1 = (double)Pl * Mp * (Mp + 1.0);   // 2 * M(M+1)/2 flops per point
    {  // c = Lu^-1 k  and |c|^2 partials
      GemmArgs a = mk(Pm, Mp, Mp, y.Kt, Mp, y.Linv, Mp, y.Ct, Mp);
      a.tri = TRI_B_UPPER; a.triblk = Mp; a.epi = 2; a.rowsq = y.cnp; a.rowsq_ld = Pm;
      RET(GX(ctx, 0, GEMM_NT, a, tri1, (double)Pl * Mp * 16));
    }
    {  // t_d = W_d^T c: |t_d|^2 partials always leave the kernel; t_d itself only for the backward pass (store_t)
      GemmArgs a = mk(Pm, (long)D * Mp, Mp, y.Ct, Mp, y.Wcat, (long)D * Mp, y.Tt, (long)D * Mp);
      a.tri = TRI_B_LOWER; a.triblk = Mp; a.epi = y.Tt ? 2 : 1; a.rowsq = y.tnp; a.rowsq_ld = Pm;
      RET(GX(ctx, 0, GEMM_NN, a, tri1 * D, (double)Pl * Mp * 8 * (y.Tt ? 1 + D : 1)));
    }
    RET(GX(ctx, 0, GEMM_NN, mk(Pm, D, Mp, y.Ct, Mp, y.u, D, y.mean0, D), 2.0 * Pl * Mp * D, (double)Pl * Mp * 8));
    {
      ProfScope ps(ctx, 1, 0, (double)Pl * nplane * 8 * (1 + D));
      HIPCHK(finalize_layer(ctx->st, y.cnp, y.tnp, nplane, Pm, y.mean0, Xin, row0, Pl, Nc, S, dedup ? 1 : 0, Din, D,
                            P(ctx, y.off_var), y.d.mean_kind, y.meanW, y.meanb,
                            zsrc_of(ctx, l, use_zs, seed, n_goff, Ntot), n0, y.mean, y.var, y.F));
    }
  }
  return DGP_OK;
}

// ------------------------------------------------------------------------------- backward over one chunk
// `params`: accumulate the parameter-gradient partial sums (training).  `xgrad0`: also produce the gradient with
// respect to the first layer's inputs (ctx->xbar, [Nc x D_in0]) -- the vector-Jacobian product used on the
// acquisition side (reference: tf.GradientTape on x, Infill_criteria.py:79-85).
struct BwdOpts {
  const double* X;   // inputs of the first layer (device), all Ntot rows
  long Ntot;
  long n_goff;
  bool params;
  bool xgrad0;
};

int backward_chunk(dgp_ctx* ctx, long n0, long Nc, int S, uint64_t seed, bool use_zs, const BwdOpts& o) {
  const int nl = (int)ctx->L.size();
  double* acc = ctx->acc;
  for (int l = nl - 1; l >= 0; --l) {
    Layer& y = ctx->L[l];
    const int M = y.d.M, Mp = y.Mp, D = y.d.D_out, Din = y.d.D_in;
    const bool dedup = (l == 0);
    const long Pl = dedup ? Nc : (long)S * Nc, Pm = pad_rows(Pl);
    const double* Xin = dedup ? o.X : ctx->L[l - 1].F;
    const long row0 = dedup ? n0 : 0;
    const long DM = (long)D * Mp, MM = (long)Mp * Mp;
    const double tri1 = (double)Pl * Mp * (Mp + 1.0);
    if (y.Tt) {  // dC = sum_d 2 vbar_d (W_d t_d - c): [2 vbar .* T] * WTcat, W_d lower => k <= n per block; "- c" in the epilogue
      GemmArgs a = mk(Pm, Mp, DM, y.Tt, DM, y.Scat, Mp, ctx->Cbar, Mp, 2.0, 0);
      a.ascale = y.vbar; a.as_ld = D; a.a_kblk = Mp; a.ascale_mode = 1; a.a_wrap = 0;
      a.tri = TRI_B_UPPER; a.triblk = Mp;
      a.eadd = y.Ct; a.eadd_nsc = D;
      a.rowf = y.mbar; a.colf = y.u; a.rank = D;             // + mbar u^T in the epilogue
      RET(GX(ctx, 0, GEMM_NN, a, tri1 * D, (double)Pl * Mp * 8 * (2 + D)));
    } else {  // dC = sum_d 2 vbar_d .* (C S'_d)      (A operand scaled on the fly; K = D*Mp re-reads C per block)
      GemmArgs a = mk(Pm, Mp, DM, y.Ct, Mp, y.Scat, Mp, ctx->Cbar, Mp, 2.0, 0);
      a.ascale = y.vbar; a.as_ld = D; a.a_kblk = Mp; a.ascale_mode = 1; a.a_wrap = 2;
      a.rowf = y.mbar; a.colf = y.u; a.rank = D;             // + mbar u^T in the epilogue
      // algorithmic count: the D triangular products W_d t_d of SURVEY App. C (the dense S' form executes 2x that)
      RET(GX(ctx, 0, GEMM_NN, a, tri1 * D, (double)Pl * Mp * 16));
    }
    {
      GemmArgs a = mk(Pm, Mp, Mp, ctx->Cbar, Mp, y.Linv, Mp, ctx->Kbar, Mp);
      a.tri = TRI_B_LOWER; a.triblk = Mp;
      a.emul = y.Et ? y.Et : y.Kt; a.C2 = ctx->Gt;          // g = dK .* e (e = -2 dk/dr2; = k for the squared exponential)
      RET(GX(ctx, 0, GEMM_NN, a, tri1, (double)Pl * Mp * 32));
    }
    // reductions over the chunk's points (accumulate into the all-reduce buffer)
    if (o.params) {
    {  // G_d = sum_p vbar_pd c_p c_p^T   (lower triangle; dW_d = 2 G_d W_d after the all-reduce)
      GemmArgs a = mk(Mp, Mp, Pl, y.Ct, Mp, y.Ct, Mp, acc + y.acc_G, Mp, 1.0, 1);
      a.batch = D; a.sC = MM; a.tri = TRI_OUT_LOWER; a.triblk = Mp; a.splits = pick_splits_tiles(ctx, lower_tiles(Mp) * D, Pl, (long)Mp * 8);
      a.ascale = y.vbar; a.as_ld = D; a.ascale_mode = 2;
      RET(GX(ctx, 0, GEMM_TN, a, tri1 * D, (double)Pl * Mp * 8));
    }
    {
      GemmArgs a = mk(Mp, Mp, Pl, ctx->Kbar, Mp, y.Ct, Mp, acc + y.acc_Q, Mp, 1.0, 1);
      a.tri = TRI_OUT_LOWER; a.triblk = Mp; a.splits = pick_splits_tiles(ctx, lower_tiles(Mp), Pl, (long)Mp * 16);
      RET(GX(ctx, 0, GEMM_TN, a, tri1, (double)Pl * Mp * 16));
    }
    {
      GemmArgs a = mk(Mp, D, Pl, y.Ct, Mp, y.mbar, D, acc + y.acc_du, D, 1.0, 1);
      a.splits = pick_splits(ctx, Mp, D, Pl);
      RET(GX(ctx, 0, GEMM_TN, a, 2.0 * Pl * Mp * D, (double)Pl * Mp * 8));
    }
    }
    {  // RBF backward through Kuf: two skinny contractions of g with [Z | 1] and [X | 1]
      const int w1 = Din + 1;
      {
        ProfScope ps(ctx, 1, 0, (double)Pl * w1 * 16);
        HIPCHK(make_x1(ctx->st, Xin, row0, Pl, Din, ctx->X1));
      }
      RET(GX(ctx, 0, GEMM_NN, mk(Pm, w1, Mp, ctx->Gt, Mp, y.Z1, w1, ctx->R1, w1), 2.0 * Pl * Mp * w1, (double)Pl * Mp * 8));
      if (o.params) {
        GemmArgs a = mk(Mp, w1, Pl, ctx->Gt, Mp, ctx->X1, w1, acc + y.acc_GX, w1, 1.0, 1);
        a.splits = pick_splits(ctx, Mp, w1, Pl);
        RET(GX(ctx, 0, GEMM_TN, a, 2.0 * Pl * Mp * w1, (double)Pl * Mp * 8));
      }
      ProfScope ps(ctx, 1, 0, (double)Pl * w1 * 24);
      HIPCHK(xbar_finish(ctx->st, ctx->R1, ctx->X1, Pl, P(ctx, y.off_ls), Din, D, y.d.mean_kind, y.meanW, y.mbar,
                         (l > 0 || o.xgrad0) ? 1 : 0, ctx->xbar, o.params ? acc + y.acc_x2 : nullptr));
    }
    if (l > 0) {
      Layer& w = ctx->L[l - 1];
      ProfScope ps(ctx, 1, 0, (double)S * Nc * Din * 24);
      HIPCHK(fold_sample_grad(ctx->st, ctx->xbar, w.var, Nc, S, (l - 1 == 0) ? 1 : 0, w.d.D_out,
                              zsrc_of(ctx, l - 1, use_zs, seed, o.n_goff, o.Ntot), n0, w.mbar, w.vbar,
                              o.params ? acc + w.acc_dvar : nullptr));
    }
  }
  return DGP_OK;
}

int check_flags(dgp_ctx* ctx) {
  int h = 0;
  HIPCHK(hipMemcpyAsync(&h, ctx->info, sizeof(int), hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  if (h) {
    HIPCHK(hipMemsetAsync(ctx->info, 0, sizeof(int), ctx->st));
    return fail(ctx, DGP_ERR_NOT_PD, "Cholesky: matrix is not positive definite");
  }
  return DGP_OK;
}

}  // namespace

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
    ctx->store_t = t && t[0] == '1';
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < dgp_ctx::kSide && ok; ++i)
      ok = hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) ctx->use_side = false;
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
  for (int i = 0; i < dgp_ctx::kSide; ++i) {
    if (ctx->side[i]) { (void)hipStreamSynchronize(ctx->side[i]); (void)hipStreamDestroy(ctx->side[i]); }
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
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
    if (d.D_in <= 0 || d.D_in > 64 || d.D_out <= 0 || d.M <= 0 || d.M > 1024 || d.D_out > 65535 || d.kernel_kind < DGP_KERNEL_RBF || d.kernel_kind > DGP_KERNEL_MATERN52 ||
        d.mean_kind < 0 || d.mean_kind > 2)
      return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: unsupported layer description (RBF kernel, M <= 1024, D_in <= 64)");
    if (l > 0 && d.D_in != layers[l - 1].D_out) return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: layer dims do not chain");
    if (d.mean_kind == DGP_MEAN_IDENTITY && d.D_in != d.D_out)
      return fail(ctx, DGP_ERR_INVALID, "dgp_model_set: identity mean needs D_in == D_out");
    Layer y{};
    y.d = d;
    y.Mp = (int)round_up(d.M, 64);   // tiles of 64 columns never straddle an output block
    y.off_Z = off; off += (long)d.M * d.D_in;
    y.off_var = off; off += 1;
    y.off_ls = off; off += d.D_in;
    y.off_qmu = off; off += (long)d.M * d.D_out;
    y.off_qsqrt = off; off += (long)d.D_out * d.M * d.M;
    const long MM = (long)y.Mp * y.Mp;
    y.acc_Q = aoff; aoff += MM;
    y.acc_G = aoff; aoff += MM * d.D_out;
    y.acc_du = aoff; aoff += round_up((long)y.Mp * d.D_out, 2);
    y.acc_GX = aoff; aoff += round_up((long)y.Mp * (d.D_in + 1), 2);
    y.acc_x2 = aoff; aoff += round_up(d.D_in, 2);
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
    RET(dev_alloc(ctx, &y.Kuu, MM)); RET(dev_alloc(ctx, &y.Lu, MM)); RET(dev_alloc(ctx, &y.Linv, MM));
    RET(dev_alloc(ctx, &y.Lq, MM * D)); RET(dev_alloc(ctx, &y.qmu_p, (long)y.Mp * D));
    RET(dev_alloc(ctx, &y.Wcat, MM * D)); RET(dev_alloc(ctx, &y.u, (long)y.Mp * D)); RET(dev_alloc(ctx, &y.Scat, MM * D)); RET(dev_alloc(ctx, &y.Z1, (long)y.Mp * (y.d.D_in + 1)));
    if (y.d.kernel_kind != DGP_KERNEL_RBF) { RET(dev_alloc(ctx, &y.Euu, MM)); RET(dev_alloc(ctx, &y.kdot, 1)); }
    RET(dev_alloc(ctx, &y.dLq, MM * D)); RET(dev_alloc(ctx, &y.dqmu_p, (long)y.Mp * D));
    if (y.d.mean_kind == DGP_MEAN_LINEAR) {
      y.meanW = ctx->mean_params + moff; moff += (long)y.d.D_in * D;
      y.meanb = ctx->mean_params + moff; moff += D;
    }
    ctx->segs.push_back({y.off_Z, (long)y.d.M * y.d.D_in, TR_IDENTITY, 1, 0});
    ctx->segs.push_back({y.off_var, 1, TR_SOFTPLUS, 1, 0});
    ctx->segs.push_back({y.off_ls, y.d.D_in, TR_SOFTPLUS, 1, 0});
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
  return DGP_OK;
}

int dgp_data_set(dgp_ctx* ctx, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy, int64_t n_global_offset) {
  if (!ctx || !X || !Y || N <= 0 || D <= 0 || Dy <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_data_set: bad arguments");
  HIPCHK(hipStreamSynchronize(ctx->st));
  dev_free(ctx->X); dev_free(ctx->Y);
  RET(dev_alloc(ctx, &ctx->X, (size_t)N * D));
  RET(dev_alloc(ctx, &ctx->Y, (size_t)N * Dy));
  HIPCHK(hipMemcpy(ctx->X, X, (size_t)N * D * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->Y, Y, (size_t)N * Dy * 8, hipMemcpyHostToDevice));
  ctx->N = N; ctx->D = D; ctx->Dy = Dy; ctx->n_goff = n_global_offset;
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

int dgp_elbo(dgp_ctx* ctx, int32_t S, uint64_t seed, const double* const* zs, double* data_term, double* kl) {
  RET(check_ready(ctx, true));
  if (S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_elbo: S must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  if (zs) RET(upload_zs(ctx, zs, S, ctx->N));
  RET(prep(ctx));
  long Nc = 0;
  RET(ensure_ws(ctx, ctx->N, S, false, &Nc));
  Layer& last = ctx->L.back();
  const bool dedup_last = ctx->L.size() == 1;
  for (long n0 = 0; n0 < ctx->N; n0 += Nc) {
    const long nc = std::min(Nc, ctx->N - n0);
    RET(forward_chunk(ctx, ctx->X, ctx->N, n0, nc, S, seed, zs != nullptr, ctx->n_goff));
    ProfScope ps(ctx, 1, 0, 0);
    HIPCHK(gauss_lik(ctx->st, last.mean, last.var, ctx->Y, n0, nc, S, dedup_last ? 1 : 0, ctx->Dy,
                     P(ctx, ctx->n_params - 1), ctx->scal + 2, nullptr, nullptr, nullptr, nullptr));
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
  if (Nn > 1024) return fail(ctx, DGP_ERR_INVALID, "dgp_propagate_full_cov: at most 1024 points (N x N covariances per sample)");
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
      HIPCHK(rbf_kuu(ctx->st, y.d.kernel_kind, Xs, P(ctx, y.off_var), P(ctx, y.off_ls), (int)N, Np, Din, Kff, nullptr));
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

int dgp_propagate_vjp(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed, const double* const* zs,
                      const double* f_bar, const double* mean_bar, const double* var_bar, double* xbar_out) {
  RET(check_ready(ctx, false));
  if (!Xnew || !xbar_out || Nn <= 0 || S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_propagate_vjp: bad arguments");
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
  long Nc = 0;
  RET(ensure_ws(ctx, Nn, S, true, &Nc));
  const bool dedup_last = nl == 1;
  for (long n0 = 0; n0 < Nn; n0 += Nc) {
    const long nc = std::min(Nc, (long)Nn - n0);
    RET(forward_chunk(ctx, ctx->Xnew, Nn, n0, nc, S, seed, zs != nullptr, 0));
    {
      ProfScope ps(ctx, 1, 0, 0);
      HIPCHK(vjp_seed(ctx->st, dev_bar[0], dev_bar[1], dev_bar[2], last.var, nc, S, dedup_last ? 1 : 0, DL,
                      zsrc_of(ctx, nl - 1, zs != nullptr, seed, 0, Nn), Nn, n0, last.mbar, last.vbar));
    }
    RET(backward_chunk(ctx, n0, nc, S, seed, zs != nullptr, BwdOpts{ctx->Xnew, (long)Nn, 0, false, true}));
    HIPCHK(hipMemcpyAsync(xbar_out + n0 * Din0, ctx->xbar, (size_t)nc * Din0 * 8, hipMemcpyDeviceToHost, ctx->st));
  }
  return check_flags(ctx);
}

int dgp_grad_partial(dgp_ctx* ctx, int32_t S, uint64_t seed, const double* const* zs) {
  RET(check_ready(ctx, true));
  if (S <= 0) return fail(ctx, DGP_ERR_INVALID, "dgp_grad_partial: S must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->grad_ready = false;
  if (zs) RET(upload_zs(ctx, zs, S, ctx->N));
  RET(prep(ctx, true));
  HIPCHK(hipMemsetAsync(ctx->acc, 0, ctx->n_acc * 8, ctx->st));
  long Nc = 0;
  RET(ensure_ws(ctx, ctx->N, S, true, &Nc));
  Layer& last = ctx->L.back();
  const bool dedup_last = ctx->L.size() == 1;
  for (long n0 = 0; n0 < ctx->N; n0 += Nc) {
    const long nc = std::min(Nc, ctx->N - n0);
    RET(forward_chunk(ctx, ctx->X, ctx->N, n0, nc, S, seed, zs != nullptr, ctx->n_goff));
    {
      ProfScope ps(ctx, 1, 0, 0);
      HIPCHK(gauss_lik(ctx->st, last.mean, last.var, ctx->Y, n0, nc, S, dedup_last ? 1 : 0, ctx->Dy,
                       P(ctx, ctx->n_params - 1), ctx->acc + 0, ctx->acc + 1, last.mbar, last.vbar,
                       ctx->acc + last.acc_dvar));
    }
    RET(backward_chunk(ctx, n0, nc, S, seed, zs != nullptr, BwdOpts{ctx->X, ctx->N, ctx->n_goff, true, false}));
  }
  return DGP_OK;
}

int dgp_acc_info(dgp_ctx* ctx, void** device_ptr, int64_t* n_doubles) {
  if (!ctx || ctx->L.empty()) return fail(ctx, DGP_ERR_INVALID, "dgp_acc_info: no model");
  if (device_ptr) *device_ptr = ctx->acc;
  if (n_doubles) *n_doubles = ctx->n_acc;
  return DGP_OK;
}

int dgp_acc_bind(dgp_ctx* ctx, void* external_device_ptr) {
  if (!ctx || ctx->L.empty()) return fail(ctx, DGP_ERR_INVALID, "dgp_acc_bind: no model");
  ctx->acc = external_device_ptr ? reinterpret_cast<double*>(external_device_ptr) : ctx->acc_own;
  return DGP_OK;
}

int dgp_grad_finish(dgp_ctx* ctx, double* elbo_out) {
  RET(check_ready(ctx, false));
  HIPCHK(hipSetDevice(ctx->device));
  double* acc = ctx->acc;
  double* g = ctx->grad;
  {
  LayerFork fork(ctx, (int)ctx->L.size());
  for (size_t li = 0; li < ctx->L.size(); ++li) {
    Layer& y = ctx->L[li];
    fork.use((int)li);
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
    HIPCHK(wbar_total(ctx->st, dW, y.Wcat, du, y.u, M, Mp, D));
    if (y.kdot) HIPCHK(lower_dot(ctx->st, y.Lu, Q, M, Mp, y.kdot));      // before Q is overwritten below
    if (!y.d.white) {
      RET(G(ctx, 2, GEMM_NT, Mp, Mp, D, du, D, y.u, D, T1, Mp, 1.0, 0));
      // K = D*Mp against an Mp x Mp output: split the reduction so that more than a handful of workgroups run
      RET(G(ctx, 2, GEMM_NT, Mp, Mp, DM, dW, DM, y.Wcat, DM, T1, Mp, 1.0, 1, 1, 0, 0, 0, D > 1 ? D : 1));
      RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Linv, Mp, T1, Mp, T2, Mp, 1.0, 0));
      RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Linv, Mp, dW, DM, y.dLq, Mp, 1.0, 0, D, 0, Mp, MM));
      RET(G(ctx, 2, GEMM_TN, Mp, D, Mp, y.Linv, Mp, du, D, y.dqmu_p, D, 1.0, 0));
      HIPCHK(lubar_finish(ctx->st, Q, T2, y.Lu, M, Mp, D, 0));
    } else {
      // W_d = L_q,d, u = q_mu: gradients pass straight through
      for (int d = 0; d < D; ++d)
        HIPCHK(hipMemcpy2DAsync(y.dLq + (long)d * MM, Mp * 8, dW + (long)d * Mp, DM * 8, Mp * 8, Mp,
                                hipMemcpyDeviceToDevice, ctx->st));
      HIPCHK(copy_mat(ctx->st, du, y.dqmu_p, (long)Mp * D));
      HIPCHK(lubar_finish(ctx->st, Q, nullptr, y.Lu, M, Mp, D, 1));
    }
    HIPCHK(lqbar_finish(ctx->st, y.dLq, y.Lq, M, Mp, D));
    // Cholesky backward: dKuu = sym(Lu^-T Phi(Lu^T dLu) Lu^-1)      
    RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Lu, Mp, Q, Mp, T3, Mp, 1.0, 0));
    HIPCHK(phi_tril_halfdiag(ctx->st, T3, Mp, 1));
    RET(G(ctx, 2, GEMM_NN, Mp, Mp, Mp, T3, Mp, y.Linv, Mp, T4, Mp, 1.0, 0));
    RET(G(ctx, 2, GEMM_TN, Mp, Mp, Mp, y.Linv, Mp, T4, Mp, Sm, Mp, 1.0, 0));
    HIPCHK(rbf_kuf_bwd_finish(ctx->st, acc + y.acc_GX, acc + y.acc_x2, acc + y.acc_dvar, P(ctx, y.off_Z), P(ctx, y.off_var),
                              P(ctx, y.off_ls), M, Din, g + y.off_Z, g + y.off_ls, g + y.off_var, y.kdot));
    HIPCHK(rbf_kuu_bwd(ctx->st, Sm, y.Kuu, y.Euu, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), M, Mp, Din,
                       g + y.off_Z, g + y.off_ls, g + y.off_var));
    HIPCHK(unpack_q_grads(ctx->st, y.dLq, y.dqmu_p, M, Mp, D, g + y.off_qsqrt, g + y.off_qmu));
  }
  }
  HIPCHK(copy_mat(ctx->st, acc + 1, g + ctx->n_params - 1, 1));
  HIPCHK(sub_scalars(ctx->st, acc + 0, ctx->scal + 0, ctx->scal + 1));   // ELBO = data term - sum KL
  ctx->grad_ready = true;
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
  ctx->adam_t += 1;
  const double t = (double)ctx->adam_t;
  const double lr_t = lr * std::sqrt(1.0 - std::pow(beta_2, t)) / (1.0 - std::pow(beta_1, t));
  ProfScope ps(ctx, 3, 0, (double)ctx->n_params * 48);
  HIPCHK(adam_apply(ctx->st, ctx->params, ctx->grad, ctx->adam_m, ctx->adam_v, ctx->segs_dev, (int)ctx->segs.size(),
                    ctx->n_params, lr_t, beta_1, beta_2, epsilon));
  return DGP_OK;
}

int dgp_natgrad_step(dgp_ctx* ctx, double gamma, const uint8_t* layer_mask) {
  if (!ctx || !ctx->grad_ready) return fail(ctx, DGP_ERR_INVALID, "dgp_natgrad_step: no gradient (call dgp_grad_finish)");
  LayerFork fork(ctx, (int)ctx->L.size());
  for (size_t l = 0; l < ctx->L.size(); ++l) {
    if (layer_mask && !layer_mask[l]) continue;
    Layer& y = ctx->L[l];
    fork.use((int)l);
    const int M = y.d.M, Mp = y.Mp, D = y.d.D_out;
    const long MM = (long)Mp * Mp;
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

int dgp_prof_enable(dgp_ctx* ctx, int32_t on) {
  if (!ctx) return DGP_ERR_INVALID;
  RET(prof_drain(ctx));
  Prof& p = ctx->prof;
  p.on = on != 0;
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
  if (!X || !Y || !ls || N <= 0 || N > 1024 || D <= 0 || D > 64 || Dy <= 0 || kind < DGP_KERNEL_RBF || kind > DGP_KERNEL_MATERN52)
    return fail(ctx, DGP_ERR_INVALID, "gpr: bad arguments (1 <= N <= 1024, 1 <= D <= 64)");
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

// ----------------------------------------------------------------------------------- unit-level hooks
int dgp_dev_gemm(dgp_ctx* ctx, int32_t op, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B,
                 int64_t ldb, double* C, int64_t ldc, double alpha, int32_t beta, int32_t splits, int32_t tri,
                 int64_t triblk, int32_t repeats, double* ms_per_call) {
  if (!ctx || !A || !B || !C || op < 0 || op > 2) return fail(ctx, DGP_ERR_INVALID, "dgp_dev_gemm: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  const long ar = (op == GEMM_TN) ? K : M, br = (op == GEMM_NT) ? N : K;
  double *dA, *dB, *dC;
  RET(dev_alloc(ctx, &dA, (size_t)ar * lda));
  RET(dev_alloc(ctx, &dB, (size_t)br * ldb));
  RET(dev_alloc(ctx, &dC, (size_t)M * ldc));
  HIPCHK(hipMemcpy(dA, A, (size_t)ar * lda * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dB, B, (size_t)br * ldb * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dC, C, (size_t)M * ldc * 8, hipMemcpyHostToDevice));
  int rc = G(ctx, 0, (GemmOp)op, M, N, K, dA, lda, dB, ldb, dC, ldc, alpha, beta, 1, 0, 0, 0, splits, tri, triblk);
  if (rc == DGP_OK) {
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipMemcpy(C, dC, (size_t)M * ldc * 8, hipMemcpyDeviceToHost));
    if (repeats > 0 && ms_per_call) {
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipEventRecord(e0, ctx->st));
      for (int r = 0; r < repeats && rc == DGP_OK; ++r)
        rc = G(ctx, 0, (GemmOp)op, M, N, K, dA, lda, dB, ldb, dC, ldc, alpha, beta, 1, 0, 0, 0, splits, tri, triblk);
      HIPCHK(hipEventRecord(e1, ctx->st));
      HIPCHK(hipEventSynchronize(e1));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e0, e1));
      *ms_per_call = ms / repeats;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
  }
  dev_free(dA); dev_free(dB); dev_free(dC);
  return rc;
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
