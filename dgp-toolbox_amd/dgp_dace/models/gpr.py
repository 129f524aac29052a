"""Exact GP regression on the MI355X engine: the model the reference builds with ``gpflow.models.GPR`` when
``num_layers == 0`` (R/dgp_dace/BO/SO_BO.py:187-200), trains with Adam on ``training_loss`` (:252-256) and queries
through ``predict_y`` (Infill_criteria.py:28-35).  Same surface: ``name == 'gpr'``, ``kernel``, ``likelihood.variance``,
``log_marginal_likelihood`` / ``training_loss`` / ``training_loss_closure``, ``predict_f`` / ``predict_y``.
The kernel matrix, its Cholesky factor, the solves and the hyper-parameter gradients run on the device
(``dgp_gpr_lml`` / ``dgp_gpr_predict``); this file holds state and the Adam arithmetic on the D + 2 scalars.
Where the reference hands ``trainable_variables`` to ``tf.optimizers.Adam().minimize``, call ``optimize_adam``.
"""
from __future__ import annotations

import numpy as np

from .. import _native
from ..gpflow_compat import KERNEL_KINDS, Gaussian, as_tensor, kernel_from_any, likelihood_from_any

LIK_LOWER = 1e-6          # gpflow.likelihoods.Gaussian: variance = softplus(u) + 1e-6


def _softplus(u):
    return np.logaddexp(0.0, u)


def _softplus_inv(x):
    x = np.asarray(x, dtype=np.float64)
    return x + np.log(-np.expm1(-x))


class GPR:
    def __init__(self, data, kernel, mean_function=None, noise_variance=1.0, device=None):
        if mean_function is not None and type(mean_function).__name__ != "Zero":
            raise NotImplementedError("GPR: only the zero mean function (what SO_BO constructs) is implemented")
        X, Y = data
        self.data = (np.ascontiguousarray(X, dtype=np.float64), np.ascontiguousarray(Y, dtype=np.float64))
        if self.data[0].ndim != 2 or self.data[1].ndim != 2 or self.data[0].shape[0] != self.data[1].shape[0]:
            raise Exception("data must be (X [N, D], Y [N, D_y])")
        self.kernel = kernel_from_any(kernel, self.data[0].shape[1])
        self.likelihood = likelihood_from_any(Gaussian(variance=noise_variance))
        self.name = "gpr"
        self._device = 0 if device is None else int(device)
        self._ctx = None

    # ------------------------------------------------------------------ engine
    def _engine(self):
        if self._ctx is None:
            self._ctx = _native.Context(self._device)
        return self._ctx

    def _hyper(self):
        return (KERNEL_KINDS[self.kernel.kind], float(self.kernel.variance.numpy()), self.kernel.lengthscales.numpy(),
                float(self.likelihood.variance.numpy()))

    @property
    def trainable_parameters(self):
        return [p for p in (self.kernel.variance, self.kernel.lengthscales, self.likelihood.variance) if p.trainable]

    trainable_variables = trainable_parameters

    # ------------------------------------------------------------------ gpflow.models.GPR surface
    def log_marginal_likelihood(self):
        kind, v, ls, s2 = self._hyper()
        return self._engine().gpr_lml(kind, *self.data, v, ls, s2, want_grad=False)[0]

    def training_loss(self):
        return -self.log_marginal_likelihood()

    def training_loss_closure(self, compile=True):
        return self.training_loss

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        if full_cov or full_output_cov:
            raise NotImplementedError("GPR.predict_f: only the marginal variances are implemented")
        kind, v, ls, s2 = self._hyper()
        Xnew = np.asarray(Xnew.numpy() if hasattr(Xnew, "numpy") else Xnew, dtype=np.float64)
        mean, var = self._engine().gpr_predict(kind, *self.data, v, ls, s2, Xnew, add_noise=False)
        return as_tensor(mean), as_tensor(var)

    def predict_y(self, Xnew, full_cov=False, full_output_cov=False):
        kind, v, ls, s2 = self._hyper()
        Xnew = np.asarray(Xnew.numpy() if hasattr(Xnew, "numpy") else Xnew, dtype=np.float64)
        mean, var = self._engine().gpr_predict(kind, *self.data, v, ls, s2, Xnew, add_noise=True)
        return as_tensor(mean), as_tensor(var)

    def predict_vjp(self, Xnew, mean_bar, var_bar):
        """d( sum(mean_bar*mean) + sum(var_bar*var) ) / dXnew of predict_y / predict_f (the noise is constant in x):
        what `tape.gradient` gives the Adam branch of the acquisition optimisers (Infill_criteria.py:69-85)."""
        kind, v, ls, s2 = self._hyper()
        Xnew = np.asarray(Xnew.numpy() if hasattr(Xnew, "numpy") else Xnew, dtype=np.float64)
        return as_tensor(self._engine().gpr_predict_vjp(kind, *self.data, v, ls, s2, Xnew, mean_bar, var_bar))

    # ------------------------------------------------------------------ training (SO_BO.py:252-256)
    def loss_and_grad(self):
        """training_loss and its gradient w.r.t. the constrained (variance, lengthscales[D], noise variance)."""
        kind, v, ls, s2 = self._hyper()
        lml, g = self._engine().gpr_lml(kind, *self.data, v, ls, s2, want_grad=True)
        return -lml, -g

    def optimize_adam(self, iterations=1000, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-07, messages=0):
        """`opt = tf.optimizers.Adam(); opt.minimize(training_loss, trainable_variables)` x iterations, on the
        unconstrained variables (softplus; the noise variance with its 1e-6 lower bound)."""
        D = self.kernel.lengthscales.shape[0]
        kind, v, ls, s2 = self._hyper()
        u = np.concatenate([[_softplus_inv(v)], _softplus_inv(ls), [_softplus_inv(s2 - LIK_LOWER)]])
        shift = np.concatenate([np.zeros(1 + D), [LIK_LOWER]])
        mask = np.concatenate([[self.kernel.variance.trainable], [self.kernel.lengthscales.trainable] * D,
                               [self.likelihood.variance.trainable]]).astype(np.float64)
        m, vv = np.zeros_like(u), np.zeros_like(u)
        loss = None
        for t in range(1, iterations + 1):
            x = _softplus(u) + shift
            self.kernel.variance.assign(x[0]); self.kernel.lengthscales.assign(x[1:1 + D]); self.likelihood.variance.assign(x[1 + D])
            loss, g = self.loss_and_grad()
            g_u = g * (1.0 / (1.0 + np.exp(-u))) * mask                      # d softplus(u)/du = sigmoid(u)
            m = beta_1 * m + (1 - beta_1) * g_u
            vv = beta_2 * vv + (1 - beta_2) * g_u * g_u
            lr_t = lr * np.sqrt(1 - beta_2 ** t) / (1 - beta_1 ** t)
            u = u - mask * lr_t * m / (np.sqrt(vv) + epsilon)
            if messages and t % messages == 0:
                print("training_loss:", loss)
        x = _softplus(u) + shift
        self.kernel.variance.assign(x[0]); self.kernel.lengthscales.assign(x[1:1 + D]); self.likelihood.variance.assign(x[1 + D])
        return loss
