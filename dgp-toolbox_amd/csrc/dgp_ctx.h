// Internal to libdgp_hip.so: the context, the per-layer state and the orchestration helpers shared by the
// translation units that implement include/dgp_abi.h (dgp_abi.hip, dgp_gpr.hip).  Everything here has internal
// linkage (anonymous namespace): each unit compiles its own copy of what it uses.
#pragma once
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
#include "chain.h"
#include <map>

using namespace dgp;

namespace {

struct Layer {
  dgp_layer_desc d;
  int Mp;
  long off_Z, off_var, off_ls, off_qmu, off_qsqrt;   // offsets in the flat parameter vector (DGP_KERNEL_MF: its 7
                                                     // parameters start at off_var)
  long off_white = -1;                               // white.variance when the kernel carries a White term
  double *meanW = nullptr, *meanb = nullptr;
  double *Kuu, *Lu, *Linv, *Lq, *qmu_p, *Wcat, *u;   // derived small matrices (padded to Mp)
  double *LinvT = nullptr;                           // Linv^T: the B operand of c = Lu^-1 k as an NN product (wide-tile kernel)
  double *alpha = nullptr;                           // LinvT u [Mp x D] (Mp = 256, D <= 8): mean0 = Kt alpha inside the Ct launch (gemm_wide.h, mode 4)
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
  unsigned mask = 0xFu;               // categories whose scopes record events (dgp_prof_enable); the others only count flops / launches
  int depth = 0;
  std::vector<hipEvent_t> ev;
  std::vector<int> cat;
  size_t used = 0;
  double ms[kNCat] = {0, 0, 0, 0}, flops[kNCat] = {0, 0, 0, 0}, bytes[kNCat] = {0, 0, 0, 0};
  long launches[kNCat] = {0, 0, 0, 0};
  std::vector<hipEvent_t> marks;      // step boundaries recorded by dgp_prof_mark (timing events on the context's stream)
  size_t marks_used = 0;
};

}  // namespace

struct dgp_ctx {
  int device = 0;
  int cu_count = 0;
  int reserved_cus = 0;     // CUs the persistent kernels of THIS context leave to its collective (dgp_comm_init)
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
  long batch_lo = 0, batch_n = 0;   // window of the resident points the bound is evaluated on (batch_n == 0: all of them)
  double data_scale = 1.0;          // factor on the data term (N / batch size for a minibatch estimate)
  double *acc = nullptr, *acc_own = nullptr;
  // Transport form of the partial-sum buffer (what a multi-GPU host all-reduces: dgp_acc_info / dgp_acc_bind): the lower
  // triangles of G_d in rectangular packed form, Q' left out when it is assembled after the reduction, everything else
  // verbatim.  `acc` stays the working (square) buffer; dgp_grad_partial packs into `acc_xfer` at its end, dgp_grad_finish
  // unpacks at its start.  Active only once a caller asked for the buffer (single-process use never packs).
  double *acc_xfer = nullptr, *acc_xfer_own = nullptr;
  long n_xfer = 0;
  long* xfer_tab = nullptr;     // device: per segment {kind (0 copy, 1 lower triangle), src offset, dst offset, length or Mp}
  int n_xfer_seg = 0;
  bool xfer_on = false, xfer_pending = false;
  double *gram_ws = nullptr;                         // partial triangles of the weighted Gram kernel (layers with Mp = 256)
  double *rg_ws = nullptr;                           // per-workgroup GX partials of the one-pass R1 / GX kernel (points.hip: rbf_bwd_contract)
  double *sl_ws = nullptr;                           // per-block partial sums of the fused small-layer backward (small_layer.hip)
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
  // Keep t_d = W_d^T c from the forward pass (16 KB/point/layer at D=8, M=256) so that dC needs only the triangular
  // products W_d t_d instead of the dense S'_d c (DGP_STORE_T=0 selects the T-free form).  History at config 2: when
  // the triangular products ran at tile granularity this was a wash (dC 19.2 -> 15.7 ms against the T-product's
  // 12.0 -> 15.1 ms for the 16 GB store); with the half-MFMA diagonal k-tiles both products of this form got
  // cheaper and it wins everywhere measured (config 2: 69.9 -> 67.1 ms, config 4 minibatch 86.8 -> 77.6 ms, config 4
  // shard 1810 -> 1656 ms, N/8 shard -0.2 ms) at +32 GB of HBM traffic per iteration and 16 GB more workspace.
  bool store_t = true;
  bool blocked_t = true;   // t_d in the engine's blocked layout (gemm_f64.h: a_blocked / c_blocked); DGP_BLOCKED_T=0: row-major
  std::vector<double*> zs_dev;
  std::vector<size_t> zs_cap;
  double* Xnew = nullptr;
  size_t Xnew_cap = 0;
  std::vector<double*> out_dev[3];
  std::vector<size_t> out_cap[3];
  bool grad_ready = false;
  int prep_level = 0;      // what the small-matrix state on the device matches: 0 nothing, 1 forward pieces, 2 + backward pieces.
                           // Parameters only change through model_set / params_set / adam_step / natgrad_step, so repeated
                           // evaluations at fixed parameters (prediction sweeps of the acquisition side, the layer calls of
                           // the multi-fidelity graph) factorise Kuu once.
  bool segs_uploaded = false;
  Prof prof;
  // overlap machinery of dgp_grad_step: per-layer events (prep done / backward left the layer / finish chain done)
  static constexpr int kMaxEv = 16;
  hipEvent_t ev_prep[kMaxEv] = {nullptr}, ev_layer[kMaxEv] = {nullptr}, ev_red[kMaxEv] = {nullptr};
  bool events_ok = false;   // every stream and event above exists (tested wherever the side path is taken)
  bool prep_wait[kMaxEv] = {false};
  bool side_touched[kSide] = {false};
  // recorded small-kernel chains (chain.h): key = (chain kind, layer, scratch set) -> program in device memory; n < 0: this
  // chain contains a launch that cannot be recorded and runs launch by launch
  struct ChainProg { ChainOp* dev = nullptr; int n = 0; };
  std::map<long, ChainProg> chains;
  bool q_from_g = true;     // Q' = sum_p cbar_p c_p^T from G_d, du and W in the finish chain instead of a pass over the points (q_from_sums below); DGP_Q_FROM_G=0: the Gram launch
  bool du_in_gram = true;   // du = Ct^T mbar inside the Gram launch of G_d (gemm_gram.h, form DU); DGP_DU_IN_GRAM=0: a launch of its own
  bool chain_on = false;    // off: measured slower than launch by launch (NOTES.md, round 3); DGP_CHAIN=1 enables
  bool prep_deferred = false, prep_deferred_train = false;   // the upper layers' prep chains are still to be enqueued (prep_deferred_layers)
  // the backward pass's HBM-bound part on a low-priority stream beside its matrix-core-bound reductions (backward_chunk)
  hipStream_t bst = nullptr;
  hipEvent_t ev_b[2] = {nullptr, nullptr};
  bool overlap_b = false;     // measured slower (NOTES.md): the HBM-bound kernels need most of the chip's CUs for their bandwidth
  int overlap_reserve_wide = 32, overlap_reserve_narrow = 96;   // CUs left to B by the Gram kernel: layers with D >= 4 / fewer outputs
  double overlap_min_work = 2.0e7;                               // smallest layer (points x Mp) that is split   // side streams that got work in the current dgp_grad_step (joined at its end)
  // library-owned RCCL communicator (dgp_comm_init); the functions come from dlopen("librccl.so.1")
  // captured-graph training loop (dgp_adam_iterations): device-side iteration state, ELBO log, the instantiated graph
  long ws_key_N = -1, ws_key_Nc = 0, ws_key_limit = 0;   // last workspace request (ensure_ws)
  int ws_key_S = 0, ws_key_train = -1;
  const void* ws_key_model = nullptr;
  double* it_dev = nullptr;            // [0] seed bits, [1] Adam step count, [2] log index
  double* elbo_log = nullptr;
  static constexpr int kLogCap = 4096;
  bool capturing = false;              // kernels read the seed / Adam step from it_dev instead of launch arguments
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  std::string graph_key;
  void* nccl_lib = nullptr;
  void* nccl_comm = nullptr;
  int comm_rank = 0, comm_world = 1;
};

namespace {

int fail(dgp_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof buf, "%s", what);
  if (ctx) {
    ctx->err = buf;
    if (code != DGP_ERR_INVALID) ctx->prep_level = 0;   // device-side failure: the cached factorisation may be partial
  }
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
// Folds the recorded event pairs into the per-category sums and empties the ring.  The pairs were recorded on whatever stream the
// context was launching on (main stream, chain streams, comm stream): every pair's END event is waited for, not just the current
// stream, and the ring is emptied whatever happens - round 4 found the bug the hard way: with the ring full (bench.py --steps 300)
// a pair still running on a side stream made hipEventElapsedTime fail, the early return left `used` at the ring's size and the next
// scope wrote behind the arrays (segmentation fault).
int prof_drain(dgp_ctx* ctx) {
  Prof& p = ctx->prof;
  if (p.used == 0) return DGP_OK;
  int rc = DGP_OK;
  for (size_t i = 0; i < p.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(p.ev[2 * i + 1]) == hipSuccess && hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]) == hipSuccess)
      p.ms[p.cat[i]] += ms;
    else
      rc = fail(ctx, DGP_ERR_HIP, "prof_drain: an event pair could not be read (its time is missing from the profile)");
  }
  p.used = 0;
  return rc;
}

struct ProfScope {
  dgp_ctx* ctx;
  bool active, counted;
  size_t slot;
  ProfScope(dgp_ctx* c, int cat, double flops, double bytes) : ctx(c), active(false), counted(false), slot(0) {
    Prof& p = c->prof;
    if (!p.on || chain_recorder()) return;
    p.flops[cat] += flops;
    p.bytes[cat] += bytes;
    if (!((p.mask >> cat) & 1u)) {      // a category that is not being timed: no event pair
      if (p.depth == 0) p.launches[cat] += 1;
      return;
    }
    counted = true;
    if (p.depth++ > 0) return;          // nested scopes are covered by the outermost one
    active = true;
    if (p.used * 2 + 2 > p.ev.size()) {
      if (p.ev.size() < 2 * 8192) {
        const size_t old = p.ev.size();
        p.ev.resize(old + 1024);
        for (size_t i = old; i < p.ev.size(); ++i) (void)hipEventCreate(&p.ev[i]);
        p.cat.resize(p.ev.size() / 2);
      } else {
        (void)prof_drain(c);
      }
    }
    if (p.used * 2 + 2 > p.ev.size()) { active = false; return; }      // (cannot happen after a drain; never index past the ring)
    slot = p.used++;
    p.cat[slot] = cat;
    p.launches[cat] += 1;
    (void)hipEventRecord(p.ev[2 * slot], c->st);
  }
  ~ProfScope() {
    Prof& p = ctx->prof;
    if (!p.on || !counted || chain_recorder()) return;
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
  a.cu_count = ctx->cu_count; a.reserve_cus = ctx->reserved_cus;
  HIPCHK(gemm_f64(ctx->st, op, a));
  return DGP_OK;
}

int GX(dgp_ctx* ctx, int cat, GemmOp op, GemmArgs a, double flops = 0.0, double bytes = 0.0) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return DGP_OK;
  ProfScope ps(ctx, cat, flops, bytes);
  // (the Gram kernel's scratch - one buffer per context, not stream-safe - is attached only by the callers that reduce
  //  over the points on the context's main stream: backward_chunk's G_d / Q' and the unit hooks)
  a.cu_count = ctx->cu_count; a.reserve_cus = std::max(a.reserve_cus, ctx->reserved_cus);
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
  // small models (one or two output tiles: M <= 128 inducing points, the Bayesian-optimisation surrogates): the whole
  // reduction over the points ran on 8 workgroups and was half of the iteration (4 launches of 100-145 us at N = 1000,
  // S = 10); 128 rows per split: the k-tiles of a split are a dependent chain of global-load latencies, its atomics one
  // 128 x 64 tile (measured at config 1: 64 rows 0.70 ms, 128: 0.69, 256: 0.73, 512: 0.76, 1024: 1.00 per iteration)
  static long small_rows = 0;
  if (small_rows == 0) { const char* e = getenv("DGP_SMALL_SPLIT_ROWS"); small_rows = e ? atol(e) : 128; if (small_rows < 16) small_rows = 16; }
  if (active_tiles <= 16 && min_rows > small_rows) min_rows = small_rows;
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

void drop_graph(dgp_ctx* ctx) {
  if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
  if (ctx->graph) (void)hipGraphDestroy(ctx->graph);
  ctx->graph_exec = nullptr; ctx->graph = nullptr; ctx->graph_key.clear();
}

void drop_chains(dgp_ctx* ctx);
void free_model(dgp_ctx* ctx) {
  drop_graph(ctx);
  drop_chains(ctx);
  ctx->ws_key_N = -1;
  ctx->prep_level = 0;
  for (auto& l : ctx->L) {
    dev_free(l.Kuu); dev_free(l.Lu); dev_free(l.Linv); dev_free(l.LinvT); dev_free(l.alpha); dev_free(l.Lq); dev_free(l.qmu_p); dev_free(l.Wcat);
    dev_free(l.u); dev_free(l.Scat); dev_free(l.Z1); dev_free(l.Euu); dev_free(l.kdot); dev_free(l.dLq); dev_free(l.dqmu_p);
  }
  ctx->L.clear();
  dev_free(ctx->params); dev_free(ctx->grad); dev_free(ctx->adam_m); dev_free(ctx->adam_v);
  dev_free(ctx->segs_dev); dev_free(ctx->mean_params); dev_free(ctx->acc_own); dev_free(ctx->gram_ws); dev_free(ctx->sl_ws); dev_free(ctx->rg_ws);
  dev_free(ctx->acc_xfer_own);
  if (ctx->xfer_tab) { (void)hipFree(ctx->xfer_tab); ctx->xfer_tab = nullptr; }
  ctx->acc = nullptr;
  ctx->acc_xfer = nullptr; ctx->n_xfer = 0; ctx->n_xfer_seg = 0; ctx->xfer_on = false; ctx->xfer_pending = false;
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
    y.Et = (train && (y.d.kernel_kind == DGP_KERNEL_MATERN32 || y.d.kernel_kind == DGP_KERNEL_MATERN52)) ? take(Pm * y.Mp) : nullptr;
  }
  if (train) {
    ctx->Cbar = take(Pmax_Mp); ctx->Kbar = take(Pmax_Mp); ctx->Gt = take(Pmax_Mp); ctx->xbar = take(xb_max);
    ctx->X1 = take(x1_max); ctx->R1 = take(x1_max);
  }
  return off;
}

int ensure_ws(dgp_ctx* ctx, long N, int S, bool train, long* Nc_out) {
  // the chunk of data points is sized from min(user limit, ~80 % of what the device has free right now + what the
  // context already holds): two ranks on one GPU, a co-tenant or a smaller part then run in more chunks instead of
  // failing; a refused allocation is retried with a smaller chunk
  // the same request as last time: same carving, no queries (also what makes a captured iteration reproducible)
  if (ctx->ws && ctx->ws_key_N == N && ctx->ws_key_S == S && ctx->ws_key_train == (train ? 1 : 0) &&
      ctx->ws_key_limit == ctx->ws_limit && ctx->ws_key_model == (const void*)ctx->params) {
    carve(ctx, ctx->ws, ctx->ws_key_Nc, S, train);
    *Nc_out = ctx->ws_key_Nc;
    return DGP_OK;
  }
  size_t limit = (size_t)ctx->ws_limit;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t avail = (size_t)((free_b + ctx->ws_cap) * 0.8);
      if (avail < limit) limit = avail;
    } else {
      (void)hipGetLastError();
    }
    if (limit < (64u << 20)) limit = 64u << 20;
  }
  const size_t per1 = carve(ctx, nullptr, 1024, S, train) / 1024 + 1;
  long Nc = (long)(limit / per1);
  if (Nc < 1) Nc = 1;
  if (Nc > N) Nc = N;
  size_t need = carve(ctx, nullptr, Nc, S, train);
  while (need > limit && Nc > 1) { Nc = Nc * 9 / 10; need = carve(ctx, nullptr, Nc, S, train); }
  while (need > ctx->ws_cap) {
    if (ctx->ws) (void)hipFree(ctx->ws);
    ctx->ws = nullptr; ctx->ws_cap = 0;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->ws), need);
    if (e == hipSuccess) { ctx->ws_cap = need; break; }
    (void)hipGetLastError();
    ctx->ws = nullptr;
    if (Nc <= 1) return fail(ctx, DGP_ERR_HIP, "workspace allocation", e);
    Nc = Nc > 16 ? Nc / 2 : 1;
    need = carve(ctx, nullptr, Nc, S, train);
  }
  carve(ctx, ctx->ws, Nc, S, train);
  *Nc_out = Nc;
  ctx->ws_key_N = N; ctx->ws_key_S = S; ctx->ws_key_train = train ? 1 : 0; ctx->ws_key_limit = ctx->ws_limit;
  ctx->ws_key_model = (const void*)ctx->params; ctx->ws_key_Nc = Nc;
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

// a layer whose per-point work runs in the fused small-model kernels (forward AND backward: they share the row-major t_d)
bool small_fused(const dgp_ctx* ctx, const Layer& y) {
  return ctx->store_t && y.d.kernel_kind != DGP_KERNEL_MF && small_layer_ok(y.d.kernel_kind, y.Mp, y.d.D_out, y.d.D_in);
}

// ------------------------------------------------------------------------------- recorded chains (small models)
enum ChainKind : int { CHAIN_PREP = 0, CHAIN_PREP_TRAIN = 1, CHAIN_FINISH = 2 };

bool chain_eligible(const dgp_ctx* ctx, const Layer& y) {
  return ctx->chain_on && !ctx->capturing && y.Mp <= 64 && y.d.D_out <= 64 && y.d.kernel_kind != DGP_KERNEL_MF && y.off_white < 0;
}

void drop_chains(dgp_ctx* ctx) {
  for (auto& kv : ctx->chains) if (kv.second.dev) (void)hipFree(kv.second.dev);
  ctx->chains.clear();
}

// Runs `enqueue` (a function that issues one layer's chain on ctx->st) as ONE launch: the first time the launches are
// recorded instead of issued and the list is uploaded; afterwards chain_exec replays it.  Every pointer a chain touches
// (parameters, the layer's matrices, the scratch set, the partial-sum buffer, the gradient) is fixed until the model or
// the bound buffers change (free_model / dgp_acc_bind drop the programs).
template <class F>
int run_chain(dgp_ctx* ctx, int kind, size_t li, F&& enqueue) {
  if (!chain_eligible(ctx, ctx->L[li])) return enqueue();
  int w = 0;
  for (int i = 0; i < dgp_ctx::kSide; ++i) if (ctx->sm == ctx->smset[i]) w = i;
  const long key = ((long)kind * 1024 + (long)li) * 8 + w;
  auto it = ctx->chains.find(key);
  if (it == ctx->chains.end()) {
    ChainRec rec;
    chain_set_recorder(&rec);
    const int r = enqueue();
    chain_set_recorder(nullptr);
    dgp_ctx::ChainProg prog;
    if (r != DGP_OK || rec.failed || rec.ops.empty()) {
      prog.n = -1;                               // not recordable (nothing was launched): launch by launch from now on
    } else {
      prog.n = (int)rec.ops.size();
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&prog.dev), rec.ops.size() * sizeof(ChainOp)));
      HIPCHK(hipMemcpy(prog.dev, rec.ops.data(), rec.ops.size() * sizeof(ChainOp), hipMemcpyHostToDevice));
    }
    it = ctx->chains.emplace(key, prog).first;
  }
  if (it->second.n < 0) return enqueue();
  ProfScope ps(ctx, 2, 0, 0);
  HIPCHK(chain_exec(ctx->st, it->second.dev, it->second.n));
  return DGP_OK;
}

// ------------------------------------------------------------------------------- prep: small matrices + KL
// One layer's chain (Kuu, Cholesky + inverse, W, u, KL, the backward's transposes) on ctx->st with the scratch set ctx->sm.
int prep_layer_launches(dgp_ctx* ctx, size_t li, bool train) {
  Layer& y = ctx->L[li];
  const int M = y.d.M, Mp = y.Mp, D = y.d.D_out, Din = y.d.D_in;
  const long MM = (long)Mp * Mp;
  HIPCHK(pack_q(ctx->st, P(ctx, y.off_qsqrt), P(ctx, y.off_qmu), M, Mp, D, y.Lq, y.qmu_p));
  if (y.d.kernel_kind == DGP_KERNEL_MF)
    HIPCHK(mf_kuu(ctx->st, P(ctx, y.off_Z), P(ctx, y.off_var), M, Mp, Din, y.Kuu));
  else
    HIPCHK(rbf_kuu(ctx->st, y.d.kernel_kind, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), M, Mp, Din, y.Kuu,
                   train ? y.Euu : nullptr));
  if (y.off_white >= 0) HIPCHK(add_diag_dev(ctx->st, y.Kuu, M, Mp, P(ctx, y.off_white)));
  HIPCHK(copy_mat(ctx->st, y.Kuu, y.Lu, MM));
  HIPCHK(potrf_inv(ctx->st, y.Lu, y.Linv, ctx->sm[9], Mp, 1, ctx->info, M));
  HIPCHK(wcat_transpose(ctx->st, y.Linv, Mp, 1, y.LinvT));
  if (y.d.white) {
    HIPCHK(lq_to_wcat(ctx->st, y.Lq, Mp, D, y.Wcat));
    HIPCHK(copy_mat(ctx->st, y.qmu_p, y.u, (long)Mp * D));
  } else {
    RET(G(ctx, 2, GEMM_NN, Mp, Mp, Mp, y.Linv, Mp, y.Lq, Mp, y.Wcat, (long)D * Mp, 1.0, 0, D, 0, MM, Mp));
    RET(G(ctx, 2, GEMM_NN, Mp, D, Mp, y.Linv, Mp, y.qmu_p, D, y.u, D, 1.0, 0));
  }
  HIPCHK(layer_kl(ctx->st, y.Wcat, y.u, y.Lq, y.Lu, M, Mp, D, y.d.white, ctx->scal));
  if (y.alpha) RET(G(ctx, 2, GEMM_NN, Mp, D, Mp, y.LinvT, Mp, y.u, D, y.alpha, D, 1.0, 0));
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
  return DGP_OK;
}

int prep_layer(dgp_ctx* ctx, size_t li, bool train) {
  {
    Layer& y = ctx->L[li];
    if (small_fused(ctx, y) && y.off_white < 0 && !ctx->chain_on) {      // at most 64 inducing points: one launch (small_layer.hip)
      ProfScope ps(ctx, 2, 0, 0);
      HIPCHK(small_prep(ctx->st, P(ctx, y.off_qsqrt), P(ctx, y.off_qmu), P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), y.d.M,
                        y.d.D_out, y.d.D_in, y.d.kernel_kind, y.d.white, train ? 1 : 0, y.Lq, y.qmu_p, y.Kuu, train ? y.Euu : nullptr,
                        y.Lu, y.Linv, y.LinvT, y.Wcat, y.u, train ? y.Scat : nullptr, train ? y.Z1 : nullptr, ctx->scal, ctx->info));
      return DGP_OK;
    }
  }
  return run_chain(ctx, train ? CHAIN_PREP_TRAIN : CHAIN_PREP, li, [&]() { return prep_layer_launches(ctx, li, train); });
}

// The chains of the layers above the first, each on a side stream, WITHOUT a join: forward_chunk waits for layer l's event
// right before it needs that layer's matrices.  Called by forward_chunk once the first layer's forward kernels have been
// ENQUEUED: a chain is ~35 launches, i.e. ~0.2 ms of host time, and with all of them issued up front the first forward
// kernel reached the device 0.75-1 ms into the iteration (seen in the kernel trace of a small shard: the device idle while
// the host was still issuing the upper layers' chains).
int prep_deferred_layers(dgp_ctx* ctx) {
  if (!ctx->prep_deferred) return DGP_OK;
  ctx->prep_deferred = false;
  const bool train = ctx->prep_deferred_train;
  hipStream_t main_st = ctx->st;
  for (size_t li = 1; li < ctx->L.size(); ++li) {
    const int w = 1 + (int)((li - 1) % 2);
    ctx->st = ctx->side[w - 1];
    ctx->sm = ctx->smset[w];
    const int r = prep_layer(ctx, li, train);
    if (r == DGP_OK) { (void)hipEventRecord(ctx->ev_prep[li], ctx->st); ctx->prep_wait[li] = true; }
    ctx->st = main_st;
    ctx->sm = ctx->smset[0];
    RET(r);
  }
  ctx->prep_level = train ? 2 : 1;        // every layer's chain has been enqueued
  return DGP_OK;
}

// `overlap`: the first layer's chain on the context's stream now, the others deferred to prep_deferred_layers (the first
// layer's forward pass covers them); without it all chains run at once on three workers and are joined here.
// `zero_acc`: the partial-sum buffer is zeroed by the same launch as the scalars (a training step's two fills in one)
int prep(dgp_ctx* ctx, bool train = false, bool overlap = false, bool zero_acc = false) {
  double* za = zero_acc ? ctx->acc : nullptr;
  const long zn = zero_acc ? ctx->n_acc : 0;
  if (ctx->prep_level >= (train ? 2 : 1)) {        // parameters unchanged: keep the factorisation and the KL in scal[0]
    HIPCHK(zero_two(ctx->st, ctx->scal + 1, 3, za, zn));
    return DGP_OK;
  }
  ctx->prep_level = 0;        // set only once every layer's chain has been enqueued: an error return below must not
                              // leave a half-built factorisation marked as current
  ctx->prep_deferred = false;
  HIPCHK(zero_two(ctx->st, ctx->scal, 4, za, zn));
  const bool ov = overlap && ctx->use_side && ctx->events_ok && ctx->L.size() > 1 && ctx->L.size() <= (size_t)dgp_ctx::kMaxEv;
  if (ov) {
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->st));      // (the side streams start behind the previous parameter update)
    for (int i = 0; i < 2; ++i) { HIPCHK(hipStreamWaitEvent(ctx->side[i], ctx->ev_fork, 0)); ctx->side_touched[i] = true; }
    {
      ProfScope ps(ctx, 2, 0, 0);
      RET(prep_layer(ctx, 0, train));
    }
    ctx->prep_deferred = true;
    ctx->prep_deferred_train = train;
    return DGP_OK;
  }
  {
    LayerFork fork(ctx, (int)ctx->L.size());
    for (size_t li = 0; li < ctx->L.size(); ++li) {
      fork.use((int)li);
      RET(prep_layer(ctx, li, train));
    }
  }   // the fork has joined the side streams
  ctx->prep_level = train ? 2 : 1;
  return DGP_OK;
}

// the deferred chains enqueued and every prep chain joined into the context's stream (callers that may not have run a
// forward pass, which is what normally consumes both)
int prep_flush(dgp_ctx* ctx) {
  RET(prep_deferred_layers(ctx));
  for (int l = 0; l < dgp_ctx::kMaxEv && l < (int)ctx->L.size(); ++l)
    if (ctx->prep_wait[l]) { HIPCHK(hipStreamWaitEvent(ctx->st, ctx->ev_prep[l], 0)); ctx->prep_wait[l] = false; }
  return DGP_OK;
}

// joins every side stream back into the context's stream on EVERY exit of dgp_grad_step / dgp_grad_partial (also error returns: chains or
// all-reduces already enqueued on the side streams must not race with the next call's memset of the partial-sum buffer)
struct SideJoin {
  dgp_ctx* ctx;
  hipStream_t main;
  bool on;
  ~SideJoin() {
    if (!on) return;
    ctx->st = main;
    ctx->sm = ctx->smset[0];
    for (int i = 0; i < dgp_ctx::kSide; ++i) {
      if (!ctx->side_touched[i]) continue;
      (void)hipEventRecord(ctx->ev_join[i], ctx->side[i]);
      (void)hipStreamWaitEvent(main, ctx->ev_join[i], 0);
      ctx->side_touched[i] = false;
    }
    for (int l = 0; l < dgp_ctx::kMaxEv; ++l) ctx->prep_wait[l] = false;   // (all prep chains are joined now)
  }
};

ZSource zsrc_of(dgp_ctx* ctx, int l, bool use_zs, uint64_t seed, long n_goff, long Ntot) {
  ZSource z;
  z.zs = use_zs ? ctx->zs_dev[l] : nullptr;
  z.seed = seed;
  z.seed_dev = ctx->capturing ? reinterpret_cast<const uint64_t*>(ctx->it_dev) : nullptr;
  z.layer = l;
  z.n_global0 = n_goff;
  z.Ntot = Ntot;
  return z;
}

// Q' (the Cholesky factor's cotangent through c = Lu^-1 k) is not reduced over the points for this layer: it follows from the
// layer's other sums (optim.hip: qprime_from_sums).  Not for the composite kernel (its Q is dK^T C, reduced directly) and not on
// the fused small-layer path (whose one sums kernel produces Q' beside the rest).
bool q_from_sums(dgp_ctx* ctx, const Layer& y) {
  // (nor under DGP_CHAIN=1: the assembly kernel is not one of the recordable chain bodies)
  return ctx->q_from_g && !ctx->chain_on && y.d.kernel_kind != DGP_KERNEL_MF && !(small_fused(ctx, y) && y.Tt);
}

// ------------------------------------------------------------------------------- the layer's point contractions
// Their GemmArgs are built in ONE place: forward_chunk / backward_chunk and the unit hook dgp_dev_layer_products (which the
// tests compare element by element with NumPy at sizes where the wide-tile and tall-tile kernels are selected) issue
// literally the same launches.
// c = Lu^-1 k: Ct = Kt * Linv^T as an NN product with the upper-triangular LinvT, row sums |c|^2 into planes
// with `alpha` (= LinvT u, [Mp x D]) and `mean0`: the launch may also produce mean0 = Ct u as Kt alpha (gemm_wide_mean_ok)
GemmArgs args_Ct(long Pm, int Mp, const double* Kt, const double* LinvT, double* Ct, double* cnp, const double* alpha = nullptr,
                 double* mean0 = nullptr, int D = 0) {
  GemmArgs a = mk(Pm, Mp, Mp, Kt, Mp, LinvT, Mp, Ct, Mp);
  a.tri = TRI_B_UPPER; a.triblk = Mp; a.epi = 2; a.rowsq = cnp; a.rowsq_ld = Pm;
  if (alpha && mean0 && D >= 1 && D <= 8) { a.mean_alpha = alpha; a.mean_out = mean0; a.mean_d = D; }
  return a;
}
// T is stored in the engine's blocked layout, unless the dC product of this shape runs on the row-panel kernel
// (gemm_dcpanel.h), which reads a plain row-major T
bool t_blocked(const dgp_ctx* ctx, long Pm, int Mp, int D) { return ctx->blocked_t && !dc_panel_shape_ok(Pm, Mp, D); }
// t_d = W_d^T c: T = Ct * Wcat (W_d lower), row sums |t_d|^2 into planes; T itself (blocked or row-major) only when Tt != nullptr
GemmArgs args_T(long Pm, int Mp, int D, const double* Ct, const double* Wcat, double* Tt, double* tnp, bool blocked) {
  GemmArgs a = mk(Pm, (long)D * Mp, Mp, Ct, Mp, Wcat, (long)D * Mp, Tt, (long)D * Mp);
  a.tri = TRI_B_LOWER; a.triblk = Mp; a.epi = Tt ? 2 : 1; a.rowsq = tnp; a.rowsq_ld = Pm;
  a.c_blocked = (Tt && blocked) ? 1 : 0;       // t_d is read back only by the dC product: blocked layout
  return a;
}
// dC = sum_d 2 vbar_d (W_d t_d - c) + mbar u^T: [2 vbar .* T] * WTcat, W_d lower => k <= n per block; "- c" and the rank-D
// term in the epilogue
GemmArgs args_Cbar(long Pm, int Mp, int D, const double* Tt, const double* Scat, double* Cbar, const double* vbar,
                   const double* Ct, const double* mbar, const double* u, bool blocked) {
  const long DM = (long)D * Mp;
  GemmArgs a = mk(Pm, Mp, DM, Tt, DM, Scat, Mp, Cbar, Mp, 2.0, 0);
  a.ascale = vbar; a.as_ld = D; a.a_kblk = Mp; a.ascale_mode = 1; a.a_wrap = 0;
  a.a_blocked = blocked ? 1 : 0;
  a.tri = TRI_B_UPPER; a.triblk = Mp;
  a.eadd = Ct; a.eadd_nsc = D;
  a.rowf = mbar; a.colf = u; a.rank = D;
  return a;
}
// dK = dC * Linv (Linv lower) and, with `emul`, only its second output g = dK .* emul
GemmArgs args_g(long Pm, int Mp, const double* Cbar, const double* Linv, double* Kbar, const double* emul, double* Gt) {
  GemmArgs a = mk(Pm, Mp, Mp, Cbar, Mp, Linv, Mp, Kbar, Mp);
  a.tri = TRI_B_LOWER; a.triblk = Mp;
  if (emul) { a.emul = emul; a.C2 = Gt; a.c2_only = 1; }
  return a;
}

// mean (before the mean function) = Ct u, and its adjoint reduction over the points du += Ct^T mbar
GemmArgs args_mean0(long Pm, int Mp, int D, const double* Ct, const double* u, double* mean0) {
  return mk(Pm, D, Mp, Ct, Mp, u, D, mean0, D);
}
// G_d = sum_p vbar_pd c_p c_p^T (lower triangle, added to G): Mp = 256 on the Gram kernel (gemm_gram.h)
GemmArgs args_G(dgp_ctx* ctx, long Pl, int Mp, int D, const double* Ct, const double* vbar, double* G) {
  GemmArgs a = mk(Mp, Mp, Pl, Ct, Mp, Ct, Mp, G, Mp, 1.0, 1);
  a.batch = D; a.sC = (long)Mp * Mp; a.tri = TRI_OUT_LOWER; a.triblk = Mp; a.splits = pick_splits_tiles(ctx, lower_tiles(Mp) * D, Pl, (long)Mp * 8);
  a.ascale = vbar; a.as_ld = D; a.ascale_mode = 2;
  a.gram_ws = ctx->gram_ws; a.gram_ws_bytes = ctx->gram_ws ? gemm_gram_ws_bytes(ctx->cu_count) : 0;
  a.cu_count = ctx->cu_count; a.reserve_cus = ctx->reserved_cus;
  return a;
}
GemmArgs args_du(dgp_ctx* ctx, long Pl, int Mp, int D, const double* Ct, const double* mbar, double* du) {
  GemmArgs a = mk(Mp, D, Pl, Ct, Mp, mbar, D, du, D, 1.0, 1);
  a.splits = pick_splits(ctx, Mp, D, Pl);
  return a;
}

// Ct then T.  Each goes to the wide-tile / tall-tile kernels when IT is eligible (row-norm partials in Mp/128 planes), else to
// the 128 x 64 engine (Mp/32 planes; the triangular solve then as an NT product with the lower Linv): finalize_layer takes
// the two plane counts separately.  (Until round 3 both had to be eligible: a 12 496-point first layer - one rank's share of
// 8 - has 784 tiles of T but 98 of Ct, and its T ran on the engine: 212 us instead of ~120.)
// *mean_done: the Ct launch also wrote mean0 (aC carried alpha and ran as the wide-tile kernel's mode 4)
void plan_Ct_T(dgp_ctx* ctx, GemmArgs& aC, GemmArgs& aT, const double* Linv, int* nplane_c, int* nplane_t) {
  const long Mp = aC.N;
  aC.cu_count = ctx->cu_count; aT.cu_count = ctx->cu_count;
  const bool wide_c = gemm_wide_ok(aC), wide_t = gemm_wide_ok(aT);
  if (!wide_c) { aC.B = Linv; aC.no_wide = 1; }
  if (!wide_t) aT.no_wide = 1;
  *nplane_c = wide_c ? (int)(Mp / 128) : (int)(Mp / 32);
  *nplane_t = wide_t ? (int)(Mp / 128) : (int)(Mp / 32);
}
int launch_Ct_T(dgp_ctx* ctx, GemmArgs aC, GemmArgs aT, const double* Linv, long Pl, int* nplane_c, int* nplane_t, bool* mean_done) {
  const long Mp = aC.N, D = aT.N / Mp;
  const double tri1 = (double)Pl * Mp * (Mp + 1.0);
  plan_Ct_T(ctx, aC, aT, Linv, nplane_c, nplane_t);
  *mean_done = !aC.no_wide && gemm_wide_mean_ok(aC);
  RET(GX(ctx, 0, aC.no_wide ? GEMM_NT : GEMM_NN, aC, tri1 + (*mean_done ? 2.0 * Pl * Mp * D : 0.0), (double)Pl * Mp * 16));
  RET(GX(ctx, 0, GEMM_NN, aT, tri1 * D, (double)Pl * Mp * 8 * (aT.C ? 1 + D : 1)));
  return DGP_OK;
}

// The two skinny contractions of the RBF backward pass through Kuf: R1 [Pm x w1] = g [Z | 1] and, with GXacc, GXacc [Mp x w1] +=
// g^T [X | 1].  One pass over g on the matrix cores where points.hip's kernel applies (Mp = 256, D_in <= 8, >= 2048 points), else
// two products on the 128 x 64 engine.  *fused (may be null): which of the two ran.
int launch_rbf_contract(dgp_ctx* ctx, const double* Gt, long Pl, long Pm, int Mp, const double* Z1, const double* X1, int w1,
                        double* R1, double* GXacc, int* fused) {
  const bool f = rbf_bwd_contract_ok(Mp, w1, Pl);
  if (fused) *fused = f ? 1 : 0;
  if (f) {
    if (GXacc && !ctx->rg_ws) RET(dev_alloc(ctx, &ctx->rg_ws, (size_t)rbf_bwd_contract_ws_doubles()));
    ProfScope ps(ctx, 0, (GXacc ? 4.0 : 2.0) * Pl * Mp * w1, (double)Pl * Mp * 8);
    HIPCHK(rbf_bwd_contract(ctx->st, Gt, Pl, Z1, X1, w1, R1, GXacc, ctx->rg_ws));
    return DGP_OK;
  }
  RET(GX(ctx, 0, GEMM_NN, mk(Pm, w1, Mp, Gt, Mp, Z1, w1, R1, w1), 2.0 * Pl * Mp * w1, (double)Pl * Mp * 8));
  if (GXacc) {
    GemmArgs a = mk(Mp, w1, Pl, Gt, Mp, X1, w1, GXacc, w1, 1.0, 1);
    a.splits = pick_splits(ctx, Mp, w1, Pl);
    RET(GX(ctx, 0, GEMM_TN, a, 2.0 * Pl * Mp * w1, (double)Pl * Mp * 8));
  }
  return DGP_OK;
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
    if (l < dgp_ctx::kMaxEv && ctx->prep_wait[l]) {     // this layer's Kuu chain was left running on a side stream
      HIPCHK(hipStreamWaitEvent(ctx->st, ctx->ev_prep[l], 0));
      ctx->prep_wait[l] = false;
    }
    // c = Lu^-1 k (|c|^2 partials) and t_d = W_d^T c (|t_d|^2 partials always leave the kernel; t_d itself only for
    // the backward pass, store_t).  Both run on the wide-tile kernel (gemm_wide.h: 2 partial planes per 256 columns)
    // when it applies, else on the 128 x 64 engine (Mp/32 planes, the triangular solve as an NT product).
    GemmArgs aC = args_Ct(Pm, Mp, y.Kt, y.LinvT, y.Ct, y.cnp, y.alpha, y.mean0, D);
    GemmArgs aT = args_T(Pm, Mp, D, y.Ct, y.Wcat, y.Tt, y.tnp, t_blocked(ctx, Pm, Mp, D));
    int nplane = 0, nplane_t = 0;
    if (small_fused(ctx, y)) {
      // at most 64 inducing points: Kuf, c, t_d, mean0 and the row norms in ONE launch (small_layer.hip)
      ProfScope ps(ctx, 0, (double)Pl * Mp * (Mp + 1.0) * (1 + D), (double)Pl * Mp * 8 * (2 + D));
      HIPCHK(small_layer_fwd(ctx->st, y.d.kernel_kind, Xin, Pl, row0, Din, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), M, D,
                             y.LinvT, y.Wcat, y.u, y.Kt, y.Et, y.Ct, y.Tt, y.cnp, y.tnp, Pm, y.mean0));
      nplane = nplane_t = 1;
    } else {
      {
        ProfScope ps(ctx, 1, 0, (double)Pl * (Mp + Din) * 8);
        if (y.d.kernel_kind == DGP_KERNEL_MF)
          HIPCHK(mf_kuf(ctx->st, Xin, Pl, row0, P(ctx, y.off_Z), P(ctx, y.off_var), M, Mp, Din, y.Kt));
        else
          HIPCHK(rbf_kuf(ctx->st, y.d.kernel_kind, Xin, Pl, row0, P(ctx, y.off_Z), P(ctx, y.off_var), P(ctx, y.off_ls), M, Mp, Din,
                         y.Kt, y.Et));
      }
      bool mean_done = false;
      RET(launch_Ct_T(ctx, aC, aT, y.Linv, Pl, &nplane, &nplane_t, &mean_done));
      if (!mean_done) RET(GX(ctx, 0, GEMM_NN, args_mean0(Pm, Mp, D, y.Ct, y.u, y.mean0), 2.0 * Pl * Mp * D, (double)Pl * Mp * 8));
    }
    {
      ProfScope ps(ctx, 1, 0, (double)Pl * 8 * (nplane + nplane_t * D));
      HIPCHK(finalize_layer(ctx->st, y.cnp, y.tnp, nplane, nplane_t, Pm, y.mean0, Xin, row0, Pl, Nc, S, dedup ? 1 : 0, Din, D,
                            P(ctx, y.off_var), y.d.mean_kind, y.meanW, y.meanb,
                            zsrc_of(ctx, l, use_zs, seed, n_goff, Ntot), n0, y.mean, y.var, y.F, y.d.kernel_kind,
                            y.off_white >= 0 ? P(ctx, y.off_white) : nullptr));
    }
    if (l == 0) RET(prep_deferred_layers(ctx));     // (after the first layer's launches: see there)
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
  // called twice per layer (last chunk): phase 0 right after layer l's launches (its sums are final behind them: the hook
  // marks the spot), phase 1 once the launches of the layer BELOW have been issued - only then does the host spend its
  // time on layer l's ~20 chain launches for a side stream; issued before them they left the main stream without work
  // (seen in the trace of a small model: 0.1 ms gaps in front of every layer's backward kernels)
  int (*after_layer)(dgp_ctx*, int, int) = nullptr;
};

int backward_chunk(dgp_ctx* ctx, long n0, long Nc, int S, uint64_t seed, bool use_zs, const BwdOpts& o) {
  const int nl = (int)ctx->L.size();
  double* acc = ctx->acc;
  for (int l = nl - 1; l >= 0; --l) {
    Layer& y = ctx->L[l];
    const int Mp = y.Mp, D = y.d.D_out, Din = y.d.D_in;
    const bool dedup = (l == 0);
    const long Pl = dedup ? Nc : (long)S * Nc, Pm = pad_rows(Pl);
    const double* Xin = dedup ? o.X : ctx->L[l - 1].F;
    const long row0 = dedup ? n0 : 0;
    const long DM = (long)D * Mp;
    const double tri1 = (double)Pl * Mp * (Mp + 1.0);
    if (small_fused(ctx, y) && y.Tt) {
      // at most 64 inducing points: dC, g, g^T [Z | 1] and [X | 1] in one launch, the four sums over the points in a second
      // (small_layer.hip); x-gradient and fold as for the large layers
      const int w1 = Din + 1;
      {
        ProfScope ps(ctx, 0, tri1 * (1 + D), (double)Pl * Mp * 8 * (4 + D));
        HIPCHK(small_layer_bwd(ctx->st, Xin, Pl, row0, Din, D, y.Linv, y.Scat, y.u, y.Z1, y.Ct, y.Tt, y.Et ? y.Et : y.Kt, y.vbar, y.mbar,
                               ctx->Cbar, ctx->Gt, ctx->R1, ctx->X1));
      }
      if (o.params) {
        ProfScope ps(ctx, 0, tri1 * (1 + D), (double)Pl * Mp * 8 * 3);
        if (!ctx->sl_ws) RET(dev_alloc(ctx, &ctx->sl_ws, (size_t)small_layer_sums_ws_doubles()));
        HIPCHK(small_layer_sums(ctx->st, Pl, Din, D, y.Ct, ctx->Cbar, ctx->Gt, y.vbar, y.mbar, ctx->X1, acc + y.acc_G, acc + y.acc_Q,
                                acc + y.acc_du, acc + y.acc_GX, ctx->sl_ws));
      }
      {
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
      if (o.after_layer) { RET(o.after_layer(ctx, l, 0)); if (l + 1 < nl) RET(o.after_layer(ctx, l + 1, 1)); }
      continue;
    }
    if (y.Tt) {  // dC = sum_d 2 vbar_d (W_d t_d - c): [2 vbar .* T] * WTcat, W_d lower => k <= n per block; "- c" in the epilogue
      GemmArgs a = args_Cbar(Pm, Mp, D, y.Tt, y.Scat, ctx->Cbar, y.vbar, y.Ct, y.mbar, y.u, t_blocked(ctx, Pm, Mp, D));
      RET(GX(ctx, 0, GEMM_NN, a, tri1 * D, (double)Pl * Mp * 8 * (2 + D)));
    } else {  // dC = sum_d 2 vbar_d .* (C S'_d)      (A operand scaled on the fly; K = D*Mp re-reads C per block)
      GemmArgs a = mk(Pm, Mp, DM, y.Ct, Mp, y.Scat, Mp, ctx->Cbar, Mp, 2.0, 0);
      a.ascale = y.vbar; a.as_ld = D; a.a_kblk = Mp; a.ascale_mode = 1; a.a_wrap = 2;
      a.rowf = y.mbar; a.colf = y.u; a.rank = D;             // + mbar u^T in the epilogue
      // algorithmic count: the D triangular products W_d t_d of SURVEY App. C (the dense S' form executes 2x that)
      RET(GX(ctx, 0, GEMM_NN, a, tri1 * D, (double)Pl * Mp * 16));
    }
    // ---- from here the layer's work splits into two independent parts that only read dC:
    //   B (HBM-bound, no matrix-core time to speak of): g = dK .* k, the two skinny contractions of g, the x-gradient, the
    //     fold into the layer below, du = Ct^T mbar - 6 GB + 3 x 2 GB of streaming per 10^6 points;
    //   A (matrix-core bound): the reductions over the points G_d (and Q' where it is still reduced: q_from_sums).
    // With `overlap_b` they run side by side: B on the context's low-priority stream, A on the main stream with R CUs left
    // free by its persistent one-workgroup-per-CU kernels (GemmArgs::reserve_cus; a workgroup of those fills its CU, so
    // without the reservation B would only start when A ends).  The main stream rejoins B before it leaves the layer
    // (the layer below needs mbar / vbar from B's fold; the layer's sums must be final for the after_layer hook).
    hipStream_t main_st = ctx->st;
    const bool ovl = ctx->overlap_b && ctx->bst != nullptr && o.params && y.d.kernel_kind != DGP_KERNEL_MF &&
                     (double)Pl * Mp >= ctx->overlap_min_work;
    if (ovl) {
      HIPCHK(hipEventRecord(ctx->ev_b[0], main_st));
      HIPCHK(hipStreamWaitEvent(ctx->bst, ctx->ev_b[0], 0));
      ctx->st = ctx->bst;
    }
    struct BackToMain { dgp_ctx* c; hipStream_t m; ~BackToMain() { c->st = m; } } back_to_main{ctx, main_st};
    // Stationary kernels at Mp = 256: g, g [Z | 1] and g^T [X | 1] in one launch that never stores g (gemm_gpanel.h)
    const bool gpanel = y.d.kernel_kind != DGP_KERNEL_MF && !ovl && g_panel_ok(Mp, Din + 1, Pl);
    if (!gpanel) {
      // g = dK .* e (e = -2 dk/dr2; = k for the squared exponential).  dK itself has one more reader, Q = dK^T C =
      // Linv^T (Cbar^T C): that reduction takes Cbar instead (finish_layer applies Linv^T to the summed 256 x 256 result), so
      // the stationary kernels never write dK: 2 GB per 10^6 points.  (The composite kernel differentiates dK directly.)
      GemmArgs a = args_g(Pm, Mp, ctx->Cbar, y.Linv, ctx->Kbar, y.d.kernel_kind != DGP_KERNEL_MF ? (y.Et ? y.Et : y.Kt) : nullptr, ctx->Gt);
      RET(GX(ctx, 0, GEMM_NN, a, tri1, (double)Pl * Mp * 32));
    }
    // du = Ct^T mbar rides on the launch that streams Ct for G_d when that one goes to the Gram kernel (gemm_gram.h, form DU)
    const int reserve = ovl ? (D >= 4 ? ctx->overlap_reserve_wide : ctx->overlap_reserve_narrow) : 0;
    GemmArgs aG = args_G(ctx, Pl, Mp, D, y.Ct, y.vbar, acc + y.acc_G);
    if (reserve > aG.reserve_cus) aG.reserve_cus = reserve;
    const bool du_in_gram = o.params && ctx->du_in_gram && gemm_engine_of(GEMM_TN, aG) == 4;
    if (o.params && !du_in_gram) {
      GemmArgs a = args_du(ctx, Pl, Mp, D, y.Ct, y.mbar, acc + y.acc_du);
      RET(GX(ctx, 0, GEMM_TN, a, 2.0 * Pl * Mp * D, (double)Pl * Mp * 8));
    }
    if (y.d.kernel_kind == DGP_KERNEL_MF) {   // non-stationary kernel: direct pair reductions of dK (mfkern.hip)
      ProfScope ps(ctx, 1, 0, (double)Pl * Mp * 16);
      if (l > 0 || o.xgrad0)
        HIPCHK(mf_kuf_bwd_x(ctx->st, ctx->Kbar, Xin, row0, Pl, P(ctx, y.off_Z), P(ctx, y.off_var), y.d.M, Mp, Din, y.vbar, D,
                            ctx->xbar));
      if (o.params) {
        HIPCHK(mf_kuf_bwd_p(ctx->st, ctx->Kbar, Xin, row0, Pl, P(ctx, y.off_Z), P(ctx, y.off_var), y.d.M, Mp, Din,
                            acc + y.acc_GX, acc + y.acc_x2));
        HIPCHK(mf_kdiag_bwd(ctx->st, y.vbar, Xin, row0, Pl, Din, D, P(ctx, y.off_var), acc + y.acc_x2));
      }
    } else {  // RBF backward through Kuf: two skinny contractions of g with [Z | 1] and [X | 1]
      const int w1 = Din + 1;
      {
        ProfScope ps(ctx, 1, 0, (double)Pl * w1 * 16);
        HIPCHK(make_x1(ctx->st, Xin, row0, Pl, Din, ctx->X1));
      }
      if (gpanel) {
        if (o.params && !ctx->rg_ws) RET(dev_alloc(ctx, &ctx->rg_ws, (size_t)rbf_bwd_contract_ws_doubles()));
        ProfScope ps(ctx, 0, tri1 + (o.params ? 4.0 : 2.0) * Pl * Mp * w1, (double)Pl * Mp * 16);
        HIPCHK(g_panel(ctx->st, ctx->Cbar, y.Linv, y.Et ? y.Et : y.Kt, Pl, y.Z1, ctx->X1, w1, ctx->R1, o.params ? acc + y.acc_GX : nullptr,
                       ctx->rg_ws, ctx->cu_count, ctx->reserved_cus));
      } else {
        RET(launch_rbf_contract(ctx, ctx->Gt, Pl, Pm, Mp, y.Z1, ctx->X1, w1, ctx->R1, o.params ? acc + y.acc_GX : nullptr, nullptr));
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
    if (ovl) {
      HIPCHK(hipEventRecord(ctx->ev_b[1], ctx->bst));
      ctx->st = main_st;
    }
    // reductions over the chunk's points (accumulate into the all-reduce buffer)
    if (o.params) {
    {  // G_d = sum_p vbar_pd c_p c_p^T   (lower triangle; dW_d = 2 G_d W_d after the all-reduce)
      GemmArgs a = aG;
      if (du_in_gram) { a.gram_mb = y.mbar; a.gram_du = acc + y.acc_du; }
      RET(GX(ctx, 0, GEMM_TN, a, tri1 * D + (du_in_gram ? 2.0 * Pl * Mp * D : 0.0), (double)Pl * Mp * 8));
    }
    if (!q_from_sums(ctx, y)) {
      // (only where Q' is not assembled from the layer's other sums: q_from_sums)
      // Q' = Cbar^T C (stationary kernels; Q = Linv^T Q' in finish_layer) or Q = dK^T C (composite kernel: dK is stored)
      GemmArgs a = mk(Mp, Mp, Pl, y.d.kernel_kind != DGP_KERNEL_MF ? ctx->Cbar : ctx->Kbar, Mp, y.Ct, Mp, acc + y.acc_Q, Mp, 1.0, 1);
      a.tri = TRI_OUT_LOWER; a.triblk = Mp; a.splits = pick_splits_tiles(ctx, lower_tiles(Mp), Pl, (long)Mp * 16);
      a.gram_ws = ctx->gram_ws; a.gram_ws_bytes = ctx->gram_ws ? gemm_gram_ws_bytes(ctx->cu_count) : 0;     // (Mp = 256: gemm_gram.h, two sources)
      a.reserve_cus = reserve;
      RET(GX(ctx, 0, GEMM_TN, a, tri1, (double)Pl * Mp * 16));
    }
    }
    if (ovl) HIPCHK(hipStreamWaitEvent(main_st, ctx->ev_b[1], 0));
    if (o.after_layer) { RET(o.after_layer(ctx, l, 0)); if (l + 1 < nl) RET(o.after_layer(ctx, l + 1, 1)); }
  }
  if (o.after_layer) RET(o.after_layer(ctx, 0, 1));
  return DGP_OK;
}

int check_flags(dgp_ctx* ctx) {
  int h = 0;
  HIPCHK(hipMemcpyAsync(&h, ctx->info, sizeof(int), hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  if (h) {
    ctx->prep_level = 0;
    HIPCHK(hipMemsetAsync(ctx->info, 0, sizeof(int), ctx->st));
    return fail(ctx, DGP_ERR_NOT_PD, "Cholesky: matrix is not positive definite");
  }
  return DGP_OK;
}

}  // namespace

