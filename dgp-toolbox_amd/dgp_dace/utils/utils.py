"""Likelihood wrapper and reparameterisation helper of the DGP (reference: dgp_dace/utils/utils.py:22-117).

On the model paths of this package (ELBO, training, prediction) these operations never run as separate steps: the
Gaussian variational expectations and the predictive moments are evaluated inside the HIP kernels
(csrc/points.hip: gauss_lik_kernel, lik_predict_var) and ``reparameterize`` is fused into finalize_layer_kernel /
fc_sample_kernel.  The functions below keep the reference's *callable surface* for code that uses them directly on
arrays it already holds (element-wise formulas on [S, N, D] NumPy arrays); the models do not call them.
Only the Gaussian likelihood is supported (for it the reference's wrapper does no tiling, utils.py:64-73).
"""
from __future__ import annotations

import numpy as np

from ..gpflow_compat import as_tensor

JITTER = 1e-6      # gpflow.default_jitter()


def _np(a):
    return np.asarray(a.numpy() if hasattr(a, "numpy") else a, dtype=np.float64)


def reparameterize(mean, var, z, full_cov=False):
    """Sample from N(mean, var) given z ~ N(0, 1) (utils.py:22-51): mean, z [S,N,D]; var [S,N,D], or [S,N,N,D] with
    full_cov=True (then the Cholesky factor of var + jitter*I is used per sample and output)."""
    if var is None:
        return mean
    mean, var, z = _np(mean), _np(var), _np(z)
    if full_cov is False:
        return as_tensor(mean + z * (var + JITTER) ** 0.5)
    N = mean.shape[1]
    chol = np.linalg.cholesky(np.transpose(var, (0, 3, 1, 2)) + JITTER * np.eye(N)[None, None])      # S,D,N,N
    f = np.transpose(mean, (0, 2, 1)) + (chol @ np.transpose(z, (0, 2, 1))[..., None])[..., 0]
    return as_tensor(np.transpose(f, (0, 2, 1)))


class BroadcastingLikelihood:
    """utils.py:54-117 for the Gaussian likelihood: inputs of shape [S,N,D], Y of shape [N,D]."""

    def __init__(self, likelihood):
        self.likelihood = likelihood
        self.needs_broadcasting = False

    def _s2(self):
        return float(_np(self.likelihood.variance))

    def variational_expectations(self, Fmu, Fvar, Y):
        Fmu, Fvar, Y = _np(Fmu), _np(Fvar), _np(Y)
        s2 = self._s2()
        return as_tensor(-0.5 * np.log(2 * np.pi) - 0.5 * np.log(s2) - 0.5 * ((Y[None] - Fmu) ** 2 + Fvar) / s2)

    def logp(self, F, Y):
        F, Y = _np(F), _np(Y)
        s2 = self._s2()
        return as_tensor(-0.5 * np.log(2 * np.pi) - 0.5 * np.log(s2) - 0.5 * (Y[None] - F) ** 2 / s2)

    def conditional_mean(self, F):
        return as_tensor(_np(F))

    def conditional_variance(self, F):
        return as_tensor(np.full(_np(F).shape, self._s2()))

    def predict_mean_and_var(self, Fmu, Fvar):
        return as_tensor(_np(Fmu)), as_tensor(_np(Fvar) + self._s2())

    def predict_density(self, Fmu, Fvar, Y):
        Fmu, Fvar, Y = _np(Fmu), _np(Fvar), _np(Y)
        v = Fvar + self._s2()
        return as_tensor(-0.5 * np.log(2 * np.pi * v) - 0.5 * (Y[None] - Fmu) ** 2 / v)
