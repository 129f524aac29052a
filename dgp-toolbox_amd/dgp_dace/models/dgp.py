"""Doubly-stochastic deep GP with the API of the reference's ``dgp_dace/models/dgp.py``.

Same constructor, attributes and methods as the reference's ``DGP_Base`` / ``DGP`` (dgp.py:21-366)
so that the notebooks and ``SO_BO`` (SO_BO.py:248-258,288; Infill_criteria.py:39,49,124) call it
unchanged; what differs is where the arithmetic happens.  Every ELBO evaluation, gradient,
Adam / natural-gradient update and prediction is executed by hand-written HIP kernels for gfx950
(libdgp_hip.so, include/dgp_abi.h) — there is no TensorFlow graph, no autodiff tape and no CPU path.

Monte-Carlo normals: the reference draws ``tf.random.normal`` inside the graph (layers.py:112-113);
here evaluation number e (0, 1, 2, … per model) uses a counter-based Philox4x32-10 stream keyed by
``seed + e`` and the global point index, so results do not depend on chunking or on the number of
GPUs.  ``propagate(..., zs=...)`` injects normals exactly like the reference (dgp.py:34,54-57).

Multi-GPU: if ``torch.distributed`` is initialised with world_size > 1, each rank keeps the data
points [rank*N/W, (rank+1)*N/W) and the per-point sums are all-reduced once per ELBO evaluation
(RCCL over xGMI) before the replicated small-matrix chain and parameter update.
"""
from __future__ import annotations

import numpy as np

from .. import _native
from .. import parallel
from ..gpflow_compat import (Parameter, Zero, as_tensor, likelihood_from_any, set_trainable)
from ..utils.layer_initializations import init_layers_linear
from ..utils.utils import BroadcastingLikelihood


class DGP_Base:
    """The base class for deep GP models: Monte-Carlo variational bound and convenience functions."""

    def __init__(self, likelihood, layers, num_samples=1, seed=0, device=None, minibatch_size=None, **kwargs):
        self.name = "dgp"
        self.num_samples = num_samples
        # None: full batch, as every caller of the reference runs it (dgp.py:95-99 has scale == 1).  An integer B makes the
        # training loops estimate the bound on B points per iteration (all ranks together) times N / B: the points are
        # shuffled once at upload and visited in consecutive windows (C-ABI dgp_batch_set).
        self.minibatch_size = minibatch_size
        self._batch_pos = 0
        self._n_local = self._n_total = 0
        self.likelihood = BroadcastingLikelihood(likelihood_from_any(likelihood))
        self.layers = layers
        self.seed = int(seed)
        self._eval_count = 0
        self._device = device
        self._ctx = None
        self._model_on_device = False
        self._host_dirty = True        # host Parameters changed since the last upload
        self._device_newer = False     # device parameters changed since the last download
        self._data_key = None
        self._dist = None
        self._acc_tensor = None
        for p in self._packed_parameters():
            p._owner = self

    # ------------------------------------------------------------------ parameter bookkeeping
    def _packed_parameters(self):
        ps = []
        for l in self.layers:
            ps += l.parameters()
        ps.append(self.likelihood.likelihood.variance)
        return ps

    @property
    def parameters(self):
        ps = list(self._packed_parameters())
        for l in self.layers:
            if l.mean_function.kind == "linear":
                ps += [l.mean_function.A, l.mean_function.b]
        return ps

    @property
    def trainable_parameters(self):
        return [p for p in self.parameters if p.trainable]

    def _before_read(self):
        if self._device_newer and self._ctx is not None:
            flat = self._ctx.params_get()
            off = 0
            for p in self._packed_parameters():
                n = p._value.size
                p._value = flat[off:off + n].reshape(p._value.shape).copy()
                off += n
            self._device_newer = False

    def _after_write(self):
        self._host_dirty = True

    def _flat(self):
        return np.concatenate([p._value.ravel() for p in self._packed_parameters()])

    def _trainable_flags(self):
        return [p.trainable for p in self._packed_parameters()]

    # ------------------------------------------------------------------ device plumbing
    def _engine(self):
        """Create the device context on first use; raises if the HIP library / GPU is missing."""
        if self._ctx is None:
            self._dist = parallel.current()
            dev = self._device if self._device is not None else (self._dist.local_device() if self._dist else 0)
            stream = self._dist.stream_handle(dev) if self._dist else None
            self._ctx = _native.Context(dev, stream)
            self._native_comm = False
            if self._dist and self._dist.native_comm():
                # the library owns an RCCL communicator: rank 0's unique id travels through the process group once.
                # Every rank must end up on the same path, so a failure anywhere (librccl not loadable, init error) is
                # agreed on through the group and all ranks keep the process group's collective instead.
                # Every step is agreed on through the group BEFORE the collective ncclCommInitRank: (1) librccl loads and
                # has every symbol on every rank, (2) rank 0 obtained a unique id.
                ok = 1.0 if _native.Context.comm_available() else 0.0
                if ok < 0.5:
                    self._say_err("librccl.so not loadable here; using the process group's all-reduce")
                if self._dist.all_reduce_min(ok, dev) > 0.5:
                    try:
                        uid = _native.Context.comm_unique_id() if self._dist.rank == 0 else bytes(128)
                    except Exception as e:           # noqa: BLE001
                        uid, ok = bytes(128), 0.0
                        self._say_err(f"dgp_comm_unique_id failed ({e}); using the process group's all-reduce")
                    uid = self._dist.broadcast_bytes(uid, dev)
                    if self._dist.all_reduce_min(ok, dev) > 0.5:
                        try:
                            self._ctx.comm_init(self._dist.rank, self._dist.world, uid)
                        except Exception as e:       # noqa: BLE001
                            ok = 0.0
                            self._say_err(f"dgp_comm_init failed ({e}); using the process group's all-reduce")
                        if self._dist.all_reduce_min(ok, dev) > 0.5:
                            self._native_comm = True
                        else:
                            self._ctx.comm_destroy()
        return self._ctx

    def _sync_model(self):
        ctx = self._engine()
        if not self._model_on_device:
            self._before_read()
            mean_params = np.concatenate([l.mean_params() for l in self.layers] + [np.zeros(0)])
            ctx.model_set([l.desc() for l in self.layers], self._flat(), mean_params)
            self._model_on_device, self._host_dirty = True, False
            if self._dist:
                ptr, n = ctx.acc_info()
                self._acc_tensor = self._dist.device_buffer(n, ctx.device)
                ctx.acc_bind(self._acc_tensor.data_ptr())
        elif self._host_dirty:
            ctx.params_set(self._flat())
            self._host_dirty = False
        return ctx

    def _sync_data(self, data):
        X, Y = data
        shuffled = self.minibatch_size is not None
        key = (id(X), id(Y), np.shape(X), np.shape(Y), shuffled)
        if key != self._data_key:
            X = np.ascontiguousarray(X.numpy() if hasattr(X, "numpy") else X, dtype=np.float64)
            Y = np.ascontiguousarray(Y.numpy() if hasattr(Y, "numpy") else Y, dtype=np.float64)
            if X.ndim != 2 or Y.ndim != 2 or X.shape[0] != Y.shape[0]:
                raise Exception("data must be (X [N, D], Y [N, D_y])")
            if shuffled:          # consecutive windows of a fixed random order = minibatches without replacement
                perm = np.random.default_rng(self.seed).permutation(X.shape[0])      # the same on every rank
                X, Y = X[perm], Y[perm]
            lo, hi = (self._dist.shard(X.shape[0]) if self._dist else (0, X.shape[0]))
            self._ctx.data_set(X[lo:hi], Y[lo:hi], n_global_offset=lo)
            self._n_local, self._n_total, self._batch_pos = hi - lo, X.shape[0], 0
            self._data_key = key
            self._keepalive = data

    def _select_batch(self, ctx, minibatch):
        """Full batch, or the next window of the shuffled resident points with the data term scaled by N / B."""
        if not minibatch or self.minibatch_size is None:
            ctx.batch_set(0, 0, 1.0)
            return
        world = self._dist.world if self._dist else 1
        b = max(1, min(self._n_local, int(self.minibatch_size) // world))
        if self._batch_pos + b > self._n_local:
            self._batch_pos = 0
        ctx.batch_set(self._batch_pos, b, self._n_total / float(b * world))
        self._batch_pos += b

    def _next_seed(self):
        s = self.seed + self._eval_count
        self._eval_count += 1
        self.last_seed = s          # the seed of the most recent evaluation (propagate_vjp replays it)
        return s

    # ------------------------------------------------------------------ forward API (dgp.py:34-124)
    def propagate(self, X, full_cov=False, S=1, zs=None):
        """Propagate the inputs through all the layers: returns (Fs, Fmeans, Fvars), one [S,N,D_l] each per layer."""
        ctx = self._sync_model()
        X = np.asarray(X.numpy() if hasattr(X, "numpy") else X, dtype=np.float64)
        if full_cov:        # N x N covariance per sample and output, samples through its Cholesky (small-N path)
            Fs, Fm, Fv = ctx.propagate_full_cov(X, int(S), self._next_seed(), zs)
        else:
            Fs, Fm, Fv = ctx.propagate(X, int(S), self._next_seed(), zs)
        return [as_tensor(a) for a in Fs], [as_tensor(a) for a in Fm], [as_tensor(a) for a in Fv]

    def propagate_vjp(self, X, S=1, f_bar=None, mean_bar=None, var_bar=None, zs=None, seed=None):
        """Gradient with respect to X of  sum(f_bar*F_L) + sum(mean_bar*Fmean_L) + sum(var_bar*Fvar_L)  for the LAST
        layer's outputs of `propagate(X, S=S)` ([S,N,D_L] cotangents, any may be None): the vector-Jacobian product
        that the reference obtains from `tf.GradientTape` on x (Infill_criteria.py:79-85). The Monte-Carlo draws are
        those of the forward evaluation being differentiated: pass the same `zs`, or `seed` (default: the seed of
        the most recent forward call, `self.last_seed`). Returns [N, D]."""
        ctx = self._sync_model()
        X = np.asarray(X.numpy() if hasattr(X, "numpy") else X, dtype=np.float64)
        if seed is None:
            seed = getattr(self, "last_seed", self.seed)
        return as_tensor(ctx.propagate_vjp(X, int(S), seed, zs, f_bar=f_bar, mean_bar=mean_bar, var_bar=var_bar))

    def predict_f(self, X, full_cov=False, S=1):
        if full_cov:
            _, Fm, Fv = self.propagate(X, full_cov=True, S=S)
            return Fm[-1], Fv[-1]
        ctx = self._sync_model()
        X = np.asarray(X.numpy() if hasattr(X, "numpy") else X, dtype=np.float64)
        nl = len(self.layers)
        _, Fm, Fv = ctx.propagate(X, int(S), self._next_seed(), None, want=(False, True, True))
        return as_tensor(Fm[nl - 1]), as_tensor(Fv[nl - 1])

    def predict_y(self, Xnew, num_samples):
        ctx = self._sync_model()
        Xnew = np.asarray(Xnew.numpy() if hasattr(Xnew, "numpy") else Xnew, dtype=np.float64)
        _, Fm, Fv = ctx.propagate(Xnew, int(num_samples), self._next_seed(), None, want=(False, True, True),
                                  add_lik_var=True)
        return as_tensor(Fm[-1]), as_tensor(Fv[-1])

    def predict_density(self, Xnew, Ynew, num_samples):
        """log (1/S) sum_s N(Ynew | Fmean_s, Fvar_s + sigma^2)   (the intent of dgp.py:126-130)."""
        m, v = self.predict_y(Xnew, num_samples)
        l = -0.5 * np.log(2 * np.pi * v) - 0.5 * (np.asarray(Ynew)[None] - m) ** 2 / v
        mx = l.max(0)
        return as_tensor(mx + np.log(np.mean(np.exp(l - mx[None]), 0)))

    def E_log_p_Y(self, X, Y):
        """[N, D_y] expected log-likelihood per point, averaged over the MC samples (dgp.py:79-87)."""
        Fmean, Fvar = self.predict_f(X, full_cov=False, S=self.num_samples)
        s2 = float(self.likelihood.likelihood.variance.numpy())
        ve = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(s2) - 0.5 * ((np.asarray(Y)[None] - Fmean) ** 2 + Fvar) / s2
        return as_tensor(np.mean(ve, 0))

    def ELBO(self, data):
        """Evidence lower bound: sum_n E_q[log p(y_n | f_n)] - sum_l KL_l   (dgp.py:89-100, scale == 1)."""
        ctx = self._sync_model()
        self._sync_data(data)
        self._select_batch(ctx, False)
        L, KL = ctx.elbo(self.num_samples, self._next_seed(), None)
        if self._dist:
            L = self._dist.all_reduce_scalar(L, ctx.device)
        return L - KL

    def ELBO_closure(self, data):
        return self.ELBO(data)

    # ------------------------------------------------------------------ training (dgp.py:132-220)
    def _grad_step(self, data):
        """One ELBO evaluation with fresh normals + its gradient, left on the device."""
        ctx = self._sync_model()
        self._sync_data(data)
        self._select_batch(ctx, True)
        if self._dist and not getattr(self, "_native_comm", False):
            # collective owned by the process group (gloo rehearsals, or when the library's own communicator is switched off):
            # three stages with the reduce between them
            ctx.grad_partial(self.num_samples, self._next_seed(), None)
            self._dist.all_reduce_(self._acc_tensor)
            ctx.grad_finish()
        else:
            # one call: per-layer all-reduce (when sharded) and small-matrix chains overlap the backward pass
            ctx.grad_step(self.num_samples, self._next_seed(), None)
        return ctx

    def _say_err(self, msg):
        import sys
        print(f"[dgp rank {self._dist.rank if self._dist else 0}] {msg}", file=sys.stderr, flush=True)

    def _say(self, msg):
        if not self._dist or self._dist.rank == 0:
            print(msg)

    def _adam_loop(self, data, iterations, lr, beta_1, beta_2, epsilon, messages, natgrad=None):
        ctx = self._sync_model()
        if self.minibatch_size is None and (self._dist is None or getattr(self, "_native_comm", False)):
            # the loop bodies run inside the library, many per call (small models: one captured hipGraph per iteration);
            # the printed lines are the same, they appear when a batch of iterations returns
            self._sync_data(data)
            self._select_batch(ctx, True)
            gamma, mask = natgrad if natgrad is not None else (0.0, None)
            per = 2 if natgrad is not None else 1
            flags = self._trainable_flags()
            step = 0
            while step < iterations:
                n = min(iterations - step, max(64, int(messages)) if messages else 4096)
                seed0 = self.seed + self._eval_count
                elbos = ctx.adam_iterations(n, self.num_samples, seed0, lr, beta_1, beta_2, epsilon, flags, gamma, mask,
                                            want_elbo=bool(messages))
                self._eval_count += per * n
                self.last_seed = seed0 + per * n - 1
                self._device_newer = True
                if messages:
                    for i in range(n):
                        if (step + i) % messages == 0:
                            self._say(f"ELBO: {elbos[i]}")
                step += n
            return
        for step in range(iterations):
            ctx = self._grad_step(data)
            ctx.adam_step(lr, beta_1, beta_2, epsilon, self._trainable_flags())
            self._device_newer = True
            if messages and step % messages == 0:
                self._say(f"ELBO: {ctx.last_elbo()}")
            if natgrad is not None:
                gamma, mask = natgrad
                ctx = self._grad_step(data)
                ctx.natgrad_step(gamma, mask)
                self._device_newer = True

    def _natgrad_setup(self, ng_all):
        nl = len(self.layers)
        if ng_all:
            for layer in self.layers:
                set_trainable(layer.q_mu, False)
                set_trainable(layer.q_sqrt, False)
            return [True] * nl
        set_trainable(self.layers[-1].q_mu, False)
        set_trainable(self.layers[-1].q_sqrt, False)
        return [False] * (nl - 1) + [True]

    def optimize_adam(self, data, iterations=5000, lr=0.01, beta_1=0.9, beta_2=0.999, epsilon=1e-07, messages=100):
        """Adam on all trainable parameters (base-class variant, dgp.py:132-154: no q_sqrt rescaling)."""
        self._sync_model().adam_reset()
        self._adam_loop(data, iterations, lr, beta_1, beta_2, epsilon, messages)

    def optimize_nat_adam(self, data, iterations1=100, iterations2=5000, lr_adam=0.01, lr_gamma=0.01, beta_1=0.9,
                          beta_2=0.999, epsilon=1e-07, ng_all=True, messages=100):
        """Adam on hyper-parameters, then alternating Adam / natural-gradient steps (dgp.py:155-220)."""
        mask = self._natgrad_setup(ng_all)
        self._sync_model().adam_reset()
        self._adam_loop(data, iterations1, lr_adam, beta_1, beta_2, epsilon, messages)
        self._adam_loop(data, iterations2, lr_adam, beta_1, beta_2, epsilon, messages, natgrad=(lr_gamma, mask))

    def sync(self):
        """Wait for queued device work; raises if a Cholesky failed meanwhile."""
        if self._ctx is not None:
            self._ctx.sync()


class DGP(DGP_Base):
    """Doubly-stochastic deep GP with linear/identity mean functions at each layer (dgp.py:221-366).

    :param X: input observations [N, D]
    :param Y: observed values [N, D_y]
    :param Z: inducing inputs [M, D]
    :param kernels: one (GPflow-style) squared-exponential kernel per layer
    :param num_units: widths of the hidden layers
    :param likelihood: a (GPflow-style) Gaussian likelihood
    :param mean_function: the final layer mean function
    """

    def __init__(self, X, Y, Z, kernels, num_units, likelihood, num_outputs=None, mean_function=None, white=False,
                 **kwargs):
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64)
        layers = init_layers_linear(X, Y, Z, kernels, num_units, num_outputs=num_outputs,
                                    mean_function=mean_function if mean_function is not None else Zero(), white=white)
        DGP_Base.__init__(self, likelihood, layers, **kwargs)
        self.data = (X, Y)

    def optimize_adam(self, iterations=5000, lr=0.01, beta_1=0.9, beta_2=0.999, epsilon=1e-07, messages=100):
        """Adam on all trainable parameters (dgp.py:255-278)."""
        for layer in self.layers[:-1]:
            layer.q_sqrt.assign(layer.q_sqrt * 1e-3)     # dgp.py:268-269 (compounds on repeated calls, as there)
        self._sync_model().adam_reset()
        self._adam_loop(self.data, iterations, lr, beta_1, beta_2, epsilon, messages)

    def optimize_nat_adam(self, iterations1=100, iterations2=5000, lr_adam=0.01, lr_gamma=0.01, beta_1=0.9,
                          beta_2=0.999, epsilon=1e-07, ng_all=True, messages=100):
        """Part 1: Adam on kernel parameters, inducing inputs and likelihood variance with q(u) fixed;
        part 2: per iteration one Adam step and one natural-gradient step on q(u) (dgp.py:280-345)."""
        mask = self._natgrad_setup(ng_all)
        for layer in self.layers[:-1]:
            layer.q_sqrt.assign(layer.q_sqrt * 1e-3)     # dgp.py:323-324
        self._sync_model().adam_reset()
        self._adam_loop(self.data, iterations1, lr_adam, beta_1, beta_2, epsilon, messages)
        self._adam_loop(self.data, iterations2, lr_adam, beta_1, beta_2, epsilon, messages, natgrad=(lr_gamma, mask))

    def ELBO(self, data=None):
        return DGP_Base.ELBO(self, self.data if data is None else data)

    def number_parameters(self, trainable=True):
        """Total number of (constrained) parameter entries (dgp.py:347-360)."""
        ps = self.trainable_parameters if trainable else self.parameters
        return int(sum(np.size(p._value) for p in ps))

    def predict(self, Xnew, num_samples):
        """Moment-matched predictive mean and variance over the MC samples (dgp.py:362-366)."""
        y_m, y_v = self.predict_y(Xnew, num_samples=num_samples)
        y_m, y_v = np.asarray(y_m), np.asarray(y_v)
        mean = np.mean(y_m, axis=0)
        return mean, np.mean(y_v + y_m ** 2, 0) - mean ** 2
