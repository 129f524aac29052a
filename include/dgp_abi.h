/* libdgp_hip.so — C-ABI of the MI355X-native doubly-stochastic DGP engine.
 *
 * The reference (Hebbalali/dgp-toolbox) has no native/FFI boundary: its hot path is reached through
 * the Python object API of dgp_dace.models.dgp.DGP and executed by TensorFlow/GPflow.  This header
 * is the boundary the build introduces underneath that API; each entry point names the reference
 * interface it replaces (file:line under the reference root).  Plain C types only: host pointers
 * are owned by the caller for the duration of the call; the context owns all device memory and one
 * HIP stream (or borrows the caller's); nothing throws across the ABI.  Every function returns 0 on
 * success or a negative dgp_status; the message is available from dgp_last_error().  A context is
 * bound to one device and is not thread-safe (one context per device per process).
 *
 * All arrays are fp64, C-contiguous (row-major), exactly as the reference's NumPy inputs
 * (layer_initializations.py:34) and tf.float64 outputs (layers.py:79).
 */
#ifndef DGP_ABI_H
#define DGP_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dgp_ctx dgp_ctx;

enum dgp_status {
  DGP_OK = 0,
  DGP_ERR_INVALID = -1,   /* bad argument / call order                                      */
  DGP_ERR_HIP = -2,       /* HIP runtime error (message has hipGetErrorString)              */
  DGP_ERR_NOT_PD = -3,    /* Cholesky met a non-positive pivot (TF: InvalidArgumentError)   */
  DGP_ERR_NO_DEVICE = -4, /* no usable gfx950 device: the library has no CPU fallback       */
  DGP_ERR_NONFINITE = -5  /* NaN/Inf in the ELBO (TF would propagate NaN silently)          */
};

enum dgp_kernel_kind {      /* gpflow stationary kernels with ARD lengthscales (SO_BO.py:192-197,239-244) */
  DGP_KERNEL_RBF = 0,       /* SquaredExponential / RBF */
  DGP_KERNEL_MATERN32 = 1,
  DGP_KERNEL_MATERN52 = 2,
  /* the multi-fidelity layer kernel of MF_DGP_EM.py:346-352 on inputs [x (D_in-1 columns), f (last column)]:
   *   k_corr(x,x') * (k_prev(f,f') + Linear(f,f')) + k_in(x,x'),  k_* = SquaredExponential with ONE lengthscale each.
   * Its 7 parameters replace `variance, lengthscales` in flat_params, in this order:
   *   corr.variance, corr.lengthscales, prev.variance, prev.lengthscales, linear.variance (0 = `add_linear=False`),
   *   in.variance, in.lengthscales.   Zero mean function only.                                                   */
  DGP_KERNEL_MF = 3
};
enum dgp_mean_kind { DGP_MEAN_ZERO = 0, DGP_MEAN_IDENTITY = 1, DGP_MEAN_LINEAR = 2 };

/* One SVGP_Layer (layers.py:181-224): kernel on D_in inputs, M inducing points, D_out outputs. */
typedef struct dgp_layer_desc {
  int32_t D_in, D_out, M;
  int32_t white;        /* layers.py:181 `white=`                        */
  int32_t kernel_kind;  /* dgp_kernel_kind                                */
  int32_t mean_kind;    /* dgp_mean_kind (layer_initializations.py:41-55) */
  int32_t kernel_white; /* 1: the kernel is `... + gpflow.kernels.White(variance)` (MF_DGP_EM.py:364-367): one more
                           parameter, white.variance, follows the kernel parameters; it enters Kuu and K_diag only */
} dgp_layer_desc;

/* ---- lifetime --------------------------------------------------------------------------------- */
/* `hip_stream`: a hipStream_t to launch on (e.g. torch's current stream), or NULL to create one. */
int dgp_create(int device, void* hip_stream, dgp_ctx** out);
void dgp_destroy(dgp_ctx* ctx);
const char* dgp_last_error(const dgp_ctx* ctx);
int dgp_sync(dgp_ctx* ctx);                 /* hipStreamSynchronize + deferred error flags */
int dgp_device_info(dgp_ctx* ctx, char* name_out, int name_len, int* cu_count, int64_t* hbm_bytes);
/* How many distinct copies of libamdhip64 are mapped in this process (dl_iterate_phdr), their paths newline-separated in
 * `paths_out`.  More than one = two ROCm stacks (torch's wheel ships its own under the system's soname): dgp_comm_init then
 * refuses with DGP_ERR_INVALID naming both, and the ctypes binding refuses to create a context.  No device is touched.  */
int dgp_hip_runtimes(char* paths_out, int cap);

/* ---- model: replaces DGP.__init__ / init_layers_linear / SVGP_Layer.__init__ state
 *      (dgp.py:245-254, layer_initializations.py:24-68, layers.py:181-224).
 * flat_params packs the *constrained* values, per layer in this order:
 *   Z[M*D_in], kern.variance[1], kern.lengthscales[D_in] (DGP_KERNEL_MF: its 7 parameters), white.variance[1] when
 *   kernel_white, q_mu[M*D_out], q_sqrt[D_out*M*M] (dense, lower-triangular), and after the last layer
 *   likelihood.variance[1].
 * mean_params packs, for every layer with DGP_MEAN_LINEAR, A[D_in*D_out] then b[D_out].          */
int dgp_model_set(dgp_ctx* ctx, int n_layers, const dgp_layer_desc* layers, const double* flat_params,
                  int64_t n_params, const double* mean_params, int64_t n_mean_params);
int64_t dgp_param_count(const dgp_ctx* ctx);
int dgp_params_get(dgp_ctx* ctx, double* flat_out);        /* Parameter.numpy()  (dgp.py:316-322 users) */
int dgp_params_set(dgp_ctx* ctx, const double* flat_in);   /* Parameter.assign() (dgp.py:268-269)       */

/* ---- data: replaces `self.data = (X, Y)` (dgp.py:254; reassigned by SO_BO.py:288).
 * n_global_offset = global index of local point 0 (non-zero only for a rank that owns a shard);
 * it only keys the counter-based normals so that draws do not depend on the sharding.            */
int dgp_data_set(dgp_ctx* ctx, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy,
                 int64_t n_global_offset);
int dgp_set_workspace_limit(dgp_ctx* ctx, int64_t bytes);  /* bound on per-chunk intermediates in HBM */
/* Minibatch estimate of the bound: dgp_elbo / dgp_grad_partial visit only the resident points [start, start+count)
 * and multiply the data term (and its gradients) by `scale` (= N / batch size).  The reference computes
 * `scale = num_data / batch size` but always feeds the full set (dgp.py:95-99; `minibatch_size` is stored and never
 * used, MF_DGP.py:88), so this is the `num_data=` extension SURVEY App. A asks for.  count == 0: all points.
 * dgp_data_set resets it to (0, 0, 1.0).  The Monte-Carlo normals stay keyed by the global point index.           */
int dgp_batch_set(dgp_ctx* ctx, int64_t start, int64_t count, double scale);

/* ---- forward: replaces DGP_Base.ELBO / ELBO_closure (dgp.py:89-109).
 * zs: NULL (draw N(0,1) from Philox4x32-10 keyed by seed) or n_layers host pointers [S,N,D_out_l]
 * (the reference's `propagate(..., zs=)`, dgp.py:34,54-57).  Outputs: data term and sum of KLs;
 * ELBO = *data_term - *kl (scale == 1, dgp.py:95-99).                                            */
int dgp_elbo(dgp_ctx* ctx, int32_t S, uint64_t seed, const double* const* zs, double* data_term, double* kl);

/* replaces DGP_Base.propagate / predict_f / predict_y (dgp.py:34-77,113-124): every output pointer
 * array has n_layers entries [S,Nn,D_out_l] (entries or whole arrays may be NULL = not wanted).
 * add_lik_var != 0 adds likelihood.variance to the last layer's Fvar (Gaussian.predict_mean_and_var). */
int dgp_propagate(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed, const double* const* zs,
                  double* const* Fs, double* const* Fmeans, double* const* Fvars, int32_t add_lik_var);

/* the full_cov=True branches of the same calls (layers.py:77-80,265-268; utils.py:43-51): per sample s and output d
 * the N x N covariance over the Nn points, samples drawn through its Cholesky factor (jitter 1e-6 as the reference).
 * Fvars entries are [S,Nn,Nn,D_out_l]; Fs / Fmeans as in dgp_propagate.  Small-N path: Nn <= 4096.             */
int dgp_propagate_full_cov(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed,
                           const double* const* zs, double* const* Fs, double* const* Fmeans, double* const* Fvars);

/* vector-Jacobian product of dgp_propagate with respect to Xnew: replaces `tape.gradient(objective, x)` on the
 * acquisition side (Infill_criteria.py:79-85, where the objective is a function of predict_f / propagate outputs).
 * f_bar, mean_bar, var_bar: cotangents of the LAST layer's sample F, Fmean and Fvar, host [S,Nn,D_out_L] each
 * (any may be NULL = zero, at least one given; the likelihood variance that predict_y adds is constant in x).
 * The same (seed | zs) as the forward call must be passed: the draws z are held fixed (reparameterisation,
 * utils.py:40-41).  xbar_out: host [Nn, D_in] = sum over outputs of cotangent * d output / d Xnew.            */
int dgp_propagate_vjp(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed, const double* const* zs,
                      const double* f_bar, const double* mean_bar, const double* var_bar, double* xbar_out);

/* dgp_propagate_vjp WITH the parameter sums: the partial sums of d/d(parameters) of  sum(cotangent * output)  are
 * ADDED to the gradient accumulator (cleared first when `reset` != 0) and the same x-gradient is returned.  Several
 * calls may precede one dgp_grad_finish, which then yields  sum over the calls of the data-side gradients  minus the
 * gradient of the layer KLs, once (the ELBO it reports is then -sum KL): this is how a model that evaluates one layer
 * at several input sets inside one bound is differentiated - the multi-fidelity graph of MF_DGP_EM.py:262-301, where
 * `tape.gradient` (MF_DGP_EM.py:460-464) reaches layers[0] through X_lf, through the mapped high-fidelity inputs and
 * through the Z_right inputs.  xbar_out may be NULL.                                                               */
int dgp_vjp_accumulate(dgp_ctx* ctx, const double* Xnew, int64_t Nn, int32_t S, uint64_t seed, const double* const* zs,
                       const double* f_bar, const double* mean_bar, const double* var_bar, double* xbar_out,
                       int32_t reset);

/* ---- exact GP regression: replaces gpflow.models.GPR as SO_BO builds it for num_layers == 0 (SO_BO.py:187-200,
 *      training :252-256, prediction through Infill_criteria.py:28-35).  Stateless (needs only dgp_create), N <= 4096.
 *   dgp_gpr_lml     : log marginal likelihood  sum_d log N(Y_d | 0, K(X,X) + noise I)  and, if grad_out != NULL, its
 *                     derivatives w.r.t. the constrained hyper-parameters: grad_out[0] variance, [1..D] lengthscales,
 *                     [1+D] noise variance  (what tape.gradient(training_loss, trainable_variables) yields, negated)
 *   dgp_gpr_predict : GPR.predict_f (add_noise == 0) / predict_y (add_noise != 0): mean_out, var_out host [Nn, Dy]  */
int dgp_gpr_lml(dgp_ctx* ctx, int32_t kernel_kind, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy,
                double variance, const double* lengthscales, double noise_variance, double* lml_out, double* grad_out);
int dgp_gpr_predict(dgp_ctx* ctx, int32_t kernel_kind, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy,
                    double variance, const double* lengthscales, double noise_variance, const double* Xnew, int64_t Nn,
                    int32_t add_noise, double* mean_out, double* var_out);
/* d( sum(mean_bar * mean) + sum(var_bar * var) ) / dXnew for the same prediction: what tape.gradient yields in the Adam
 * branch of the acquisition optimisers on a 'gpr' model (Infill_criteria.py:69-85); mean_bar, var_bar host [Nn, Dy],
 * xbar_out host [Nn, D]                                                                                           */
int dgp_gpr_predict_vjp(dgp_ctx* ctx, int32_t kernel_kind, const double* X, const double* Y, int64_t N, int32_t D, int32_t Dy,
                        double variance, const double* lengthscales, double noise_variance, const double* Xnew, int64_t Nn,
                        const double* mean_bar, const double* var_bar, double* xbar_out);

/* ---- backward + optimisers: replaces the tf.GradientTape / Adam / NaturalGradient loop bodies
 *      (dgp.py:270-276, 326-345).  Split so that a multi-GPU host can all-reduce between the stages:
 *   dgp_grad_partial : this rank's sums over its data points (ELBO data term + every point-sum the
 *                      gradient needs) into one contiguous device buffer
 *   dgp_acc_info     : that buffer (device pointer, length in doubles) for an in-place all-reduce, in its TRANSPORT form:
 *                      the lower triangles of the per-output Gram sums G_d in rectangular packed form (Mp (Mp + 1) / 2
 *                      doubles each), Q' only for layers that still reduce it over the points, every other sum verbatim
 *                      (config 2: 4.7 MB instead of 10.6 MB of squares).  Calling dgp_acc_info or dgp_acc_bind switches the
 *                      transport form on: from then on dgp_grad_partial ends by packing into the buffer and dgp_grad_finish
 *                      starts by unpacking it - so ask for (or bind) the buffer BEFORE dgp_grad_partial.  A context that is
 *                      never asked (single process) neither packs nor allocates it.
 *   dgp_grad_finish  : small-matrix chain (Cholesky backward, KL gradient) -> d ELBO / d params     */
int dgp_grad_partial(dgp_ctx* ctx, int32_t S, uint64_t seed, const double* const* zs);
int dgp_acc_info(dgp_ctx* ctx, void** device_ptr, int64_t* n_doubles);
int dgp_acc_bind(dgp_ctx* ctx, void* external_device_ptr);   /* use caller-owned device memory for it */
int dgp_grad_finish(dgp_ctx* ctx, double* elbo_out /* NULL = do not synchronise */);
int dgp_grad_get(dgp_ctx* ctx, double* flat_grad_out);       /* d ELBO / d constrained params, same packing */
int dgp_last_elbo(dgp_ctx* ctx, double* elbo_out);           /* ELBO of the last dgp_grad_finish (synchronises) */
/* dgp_grad_partial + (all-reduce) + dgp_grad_finish as ONE call that overlaps them: the loop body of dgp.py:271-275
 * (one `tape.gradient(-ELBO)` evaluation).  As soon as the backward pass leaves layer l, that layer's slice of the
 * partial-sum buffer is summed over the ranks (when dgp_comm_init was called; on the library's own streams) and the
 * layer's small-matrix chain runs, while the layers below are still in their backward pass; the Kuu factorisations of
 * the layers above the first also run under the first layer's forward pass.  Results are those of the three-call
 * sequence.  elbo_out NULL = do not synchronise.                                                                    */
int dgp_grad_step(dgp_ctx* ctx, int32_t S, uint64_t seed, const double* const* zs, double* elbo_out);

/* ---- multi-GPU: one process per GPU, the data points sharded (dgp.py:87,96 are sums over points).  The reference has
 * no multi-device code.  The library owns an RCCL communicator (librccl is loaded with dlopen,
 * the process's existing copy when torch has loaded one) so that dgp_grad_step can enqueue its all-reduces itself,
 * without a host round trip between the backward pass and the small-matrix chains.
 *   dgp_comm_unique_id : rank 0 fills 128 bytes (ncclUniqueId); the host broadcasts them to the other ranks
 *   dgp_comm_init      : every rank, same id; world == 1 with id128 == NULL creates no communicator (nothing to reduce),
 *                        world == 1 with an id creates a one-rank communicator (exercises the RCCL path on one GPU)
 *   dgp_comm_allreduce : in-place sum of n doubles at a DEVICE pointer, on the context's stream (test / utility)
 *   dgp_comm_available : 0 when librccl.so can be loaded and has every entry point the library binds.  EVERY rank calls it
 *                        and the ranks agree on the answers BEFORE any of them calls dgp_comm_init (ncclCommInitRank is
 *                        collective: a rank that cannot join would leave the others waiting)                              */
int dgp_comm_available(void);
int dgp_comm_unique_id(void* id128_out);
int dgp_comm_init(dgp_ctx* ctx, int32_t rank, int32_t world, const void* id128);
int dgp_comm_destroy(dgp_ctx* ctx);
int dgp_comm_allreduce(dgp_ctx* ctx, void* device_ptr, int64_t n_doubles);

/* tf.optimizers.Adam on the unconstrained variables (dgp.py:267,276; 311,333,342).
 * trainable: 5*n_layers+1 flags in the order [Z, variance, lengthscales, q_mu, q_sqrt]*, lik variance
 * (gpflow.set_trainable, dgp.py:316-322).  Uses the gradient of the last dgp_grad_finish.          */
int dgp_adam_reset(dgp_ctx* ctx);
int dgp_adam_step(dgp_ctx* ctx, double lr, double beta_1, double beta_2, double epsilon, const uint8_t* trainable);
/* n_iter loop bodies of DGP.optimize_adam (dgp.py:271-276: ELBO + gradient with fresh normals, Adam step) or, with
 * gamma > 0, of part 2 of DGP.optimize_nat_adam (dgp.py:326-345: that, then a second evaluation and the natural-gradient
 * step on the layers of layer_mask) in ONE call.  Evaluation k of the call draws its normals with seed0 + k (one
 * evaluation per iteration, two with gamma > 0).  elbo_out (host, [n_iter], may be NULL): the ELBO each iteration would
 * print.  use_graph = 1: one captured hipGraph is replayed per iteration (seed and Adam step count then live in device
 * memory); 0: call by call; -1: the library decides (the graph for models of at most two SVGP layers that run on the fused
 * small-layer kernels, where the replay measured faster; call by call otherwise -- see the implementation).
 * Results are those of the call-by-call sequence dgp_grad_step / dgp_adam_step (/ dgp_grad_step / dgp_natgrad_step).   */
int dgp_adam_iterations(dgp_ctx* ctx, int32_t n_iter, int32_t S, uint64_t seed0, double lr, double beta_1, double beta_2,
                        double epsilon, const uint8_t* trainable, double gamma, const uint8_t* layer_mask,
                        int32_t use_graph, double* elbo_out);
/* gpflow.optimizers.NaturalGradient(gamma).minimize on the (q_mu, q_sqrt) pairs of the layers with
 * layer_mask[l] != 0 (dgp.py:312-322,343).  Uses the gradient of the last dgp_grad_finish.         */
int dgp_natgrad_step(dgp_ctx* ctx, double gamma, const uint8_t* layer_mask);

/* ---- measurement: HIP-event timing of kernel launches on the context's stream, by category.
 * categories: 0 = fp64-MFMA contractions over points, 1 = per-point streaming kernels,
 *             2 = small-matrix chain (Kuu/Cholesky/KL/backward chain/natgrad), 3 = Adam.             */
/* on: 0 = off; 1 = every category; 0x100 | mask = only the categories whose bit (1 << category) is set in mask record events -
 * the others still count launches and algorithmic flops / bytes, their milliseconds read 0.  (A timed region that needs one
 * category's duration need not put every other launch between events; at config 2 it measured the same either way.)       */
int dgp_prof_enable(dgp_ctx* ctx, int32_t on);
int dgp_prof_read(dgp_ctx* ctx, int32_t n_cat, double* ms_out, int64_t* launches_out, double* alg_flops_out,
                  double* alg_bytes_out);   /* synchronises, returns totals since enable, then resets */

/* Step boundaries: dgp_prof_mark records a HIP event on the context's stream (no synchronisation); dgp_prof_marks_read waits
 * for the last one, returns the milliseconds between consecutive marks (n_marks - 1 intervals, at most n_max) and clears
 * them.  bench.py brackets every iteration with marks and reports the median interval.           */
int dgp_prof_mark(dgp_ctx* ctx);
int dgp_prof_marks_read(dgp_ctx* ctx, int32_t n_max, double* ms_between_out, int32_t* n_out);

/* ---- unit-level entry points (used by tests/ to check single kernels against NumPy) -------------
 * dgp_dev_gemm: C = alpha op(A) op(B) (+ C if beta) on the engine the shapes select (128 x 64 engine, wide-tile kernel,
 * weighted Gram kernel).  The same host pointer for A and B (equal shapes) is uploaded ONCE and used as both operands:
 * the one-source Gram form  C += A^T A  (tri = 3, M = N = 256, K a multiple of 16 >= 8192). */
int dgp_dev_gemm(dgp_ctx* ctx, int32_t op /*0 NN,1 NT,2 TN*/, int64_t M, int64_t N, int64_t K, const double* A,
                 int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double alpha, int32_t beta,
                 int32_t splits, int32_t tri, int64_t triblk, int32_t repeats, double* ms_per_call);
/* G[d] += sum_p s[p][d] c_p c_p^T (lower triangles; s == NULL: weights 1, D = 1) through the library's dispatcher, i.e.
 * the weighted Gram kernel when eligible (Mp = 256, P a multiple of 16 >= 8192, D <= 8), else the 128 x 64 engine:
 * the reduction  G_d = sum_p vbar_pd c_p c_p^T  of SURVEY App. C step 2 on its own.  With mb / du (both or neither; needs
 * s): also du[m][d] += sum_p mb[p][d] C[p][m] (the adjoint of mean = c^T u_d, App. C step 1), inside the same launch when the
 * Gram kernel runs it, as in the backward pass. */
int dgp_dev_gram(dgp_ctx* ctx, const double* C /* [P,256] */, const double* s /* [P,D] or NULL */, int64_t P, int32_t D,
                 double* G /* [D,256,256], in/out */, const double* mb /* [P,D] or NULL */, double* du /* [256,D] in/out, or NULL */);
/* The point contractions of ONE SVGP layer on caller-supplied operands, issued through the same argument builders as the
 * forward / backward pass (csrc/dgp_ctx.h: args_Ct, args_T, args_mean0, args_Cbar, args_g, args_du, args_G), so that a size selects the
 * kernel it selects in training (128 x 64 engine, wide-tile, tall-tile): layers.py:243-263 in whitened form and its
 * adjoint.  Mp a multiple of 64, Linv lower triangular, Wcat = [W_0 | ... | W_{D-1}], W_d lower.
 *   Ct   = Kt Linv^T                                  cn[p]    = |c_p|^2
 *   T_d  = Ct W_d   (returned row-major [P, D*Mp])    tn[p][d] = |t_pd|^2        mean0 = Ct u
 *   Cbar = sum_d 2 vbar_pd (T_d W_d^T - Ct) + mbar u^T
 *   g    = (Cbar Linv) .* Kt                          du = Ct^T mbar
 *   Gd[d] = sum_p vbar_pd c_p c_p^T  (lower triangle of [D, Mp, Mp]; with Mp = 256 and enough points du rides on this launch)
 * engines[7] (may be NULL): the kernel family (gemm_engine_of) that ran Ct, T, Cbar, g, du, Gd; [6] = 1 when mean0 was
 * produced inside the Ct launch (as Kt (LinvT u): wide-tile kernel, Mp = 256, D <= 8), 0 when by its own product Ct u.   */
int dgp_dev_layer_products(dgp_ctx* ctx, int64_t P, int32_t Mp, int32_t D, const double* Kt, const double* Linv,
                           const double* Wcat, const double* u, const double* vbar, const double* mbar, double* Ct,
                           double* cn, double* T, double* tn, double* mean0, double* Cbar, double* g, double* du,
                           double* Gd, int32_t* engines);
/* The two skinny contractions of the backward pass through an RBF / Matern Kuf (what tf.GradientTape derives for Z, the
 * lengthscales and the layer's inputs through gpflow's K(Z, X): layers.py:243 under dgp.py:272-275), through
 * the launcher backward_chunk uses:  R1 [P, w1] = G Z1  and  GX [Mp, w1] += G^T X1  (G [P, Mp], Z1 = [Z | 1], X1 = [X | 1], w1 =
 * D_in + 1).  *fused (may be NULL): 1 when both ran as one pass over G on the matrix cores (Mp = 256, w1 <= 9, P >= 2048), 0 when
 * as two products on the 128 x 64 engine. */
int dgp_dev_rbf_contract(dgp_ctx* ctx, const double* G, const double* Z1, const double* X1, int64_t P, int32_t Mp, int32_t w1,
                         double* R1, double* GX /* in/out */, int32_t* fused);
/* The same backward step in the form training uses from round 4 on for stationary kernels with 256 (padded) inducing points:
 * g = (Cbar Linv) .* E is never stored; one launch (csrc/gemm_gpanel.h) returns R1 [P, w1] = g Z1 and GX [256, w1] += g^T X1
 * (Cbar, E [P, 256]; Linv [256, 256] lower triangular; 2 <= w1 <= 9).  *used: 1 when that kernel ran, 0 when the size / switches
 * send the backward pass down the stored-g path instead (then nothing is computed here). */
int dgp_dev_g_panel(dgp_ctx* ctx, const double* Cbar, const double* Linv, const double* E, const double* Z1, const double* X1,
                    int64_t P, int32_t w1, double* R1, double* GX /* in/out */, int32_t* used);
int dgp_dev_chol(dgp_ctx* ctx, double* A, int32_t M, int32_t batch);            /* in place, lower */
int dgp_dev_trinv(dgp_ctx* ctx, const double* L, double* X, int32_t M, int32_t batch);
int dgp_dev_normals(dgp_ctx* ctx, uint64_t seed, int32_t layer, int32_t S, int64_t n0, int64_t N, int32_t D,
                    double* z_out /* [S,N,D] */);
int dgp_dev_mfma_peak(dgp_ctx* ctx, int32_t iters, double* tflops_out);          /* v_mfma_f64_16x16x4_f64 issue rate */

#ifdef __cplusplus
}
#endif
#endif /* DGP_ABI_H */
