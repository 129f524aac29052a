"""Host-side description of one sparse variational GP layer.

Mirrors the state of the reference's ``SVGP_Layer`` (dgp_dace/utils/layers.py:181-224): kernel,
inducing inputs ``feature.Z``, ``q_mu [M, D_out]``, ``q_sqrt [D_out, M, M]``, mean function, ``white``.
The layer's arithmetic (Kuu/Cholesky, Kuf, whitened solves, mean/variance, sampling, KL —
layers.py:227-308) is not executed here: it runs in libdgp_hip.so (csrc/), which the owning DGP
drives.  ``conditional_ND`` / ``sample_from_conditional`` / ``KL`` are offered for API parity and
evaluate through a private one-layer device context.
"""
from __future__ import annotations

import numpy as np

from ..gpflow_compat import KERNEL_KINDS, Parameter, as_tensor, kernel_matrix_host, split_white

JITTER = 1e-6      # gpflow.default_jitter() (layers.py:222)
MEAN_KINDS = {"zero": 0, "identity": 1, "linear": 2}


class InducingPoints:
    """gpflow.inducing_variables.InducingPoints stand-in: holds Z."""

    def __init__(self, Z):
        self.Z = Parameter(np.array(Z, dtype=np.float64), "Z")


class Layer:
    def __init__(self, input_prop_dim=None, **kwargs):
        if input_prop_dim:
            raise NotImplementedError("input_prop_dim is not used by DGP (layers.py:116-128) and is not implemented")
        self.input_prop_dim = None


class SVGP_Layer(Layer):
    def __init__(self, kern, Z, num_outputs, mean_function, augmented=False, layers=None, white=False,
                 input_prop_dim=None, **kwargs):
        Layer.__init__(self, input_prop_dim)
        if augmented:
            raise NotImplementedError("augmented inducing points belong to the multi-fidelity models (out of scope)")
        Z = np.array(Z, dtype=np.float64)
        self.num_inducing = Z.shape[0]
        self.num_outputs = int(num_outputs)
        self.kern = kern
        self.mean_function = mean_function
        self.white = bool(white)
        self.feature = InducingPoints(Z)
        M = self.num_inducing
        self.q_mu = Parameter(np.zeros((M, self.num_outputs)), "q_mu")
        if self.white:
            q_sqrt = np.tile(np.eye(M)[None], [self.num_outputs, 1, 1])
        else:   # q(u) = prior p(u): q_sqrt = chol(K(Z) + jitter I), once, on the host as the reference does
            Lu = np.linalg.cholesky(kernel_matrix_host(kern, Z) + np.eye(M) * JITTER)
            q_sqrt = np.tile(Lu[None], [self.num_outputs, 1, 1])
        self.q_sqrt = Parameter(q_sqrt, "q_sqrt", "tril")
        self._private = None

    # ---- packing for the C-ABI (include/dgp_abi.h: dgp_model_set) ----
    @property
    def input_dim(self):
        return self.feature.Z.shape[1]

    def desc(self):
        base, white = split_white(self.kern)
        return (self.input_dim, self.num_outputs, self.num_inducing, 1 if self.white else 0,
                KERNEL_KINDS[base.kind], MEAN_KINDS[self.mean_function.kind], 0 if white is None else 1)

    def parameters(self):
        """[Z, variance, lengthscales, (white.variance,) q_mu, q_sqrt] — the packing order of dgp_model_set."""
        base, white = split_white(self.kern)
        return [self.feature.Z, base.variance, base.lengthscales] + ([] if white is None else [white.variance]) + \
               [self.q_mu, self.q_sqrt]

    def mean_params(self):
        if self.mean_function.kind != "linear":
            return np.zeros(0)
        A = self.mean_function.A._value
        b = np.broadcast_to(self.mean_function.b._value, (self.num_outputs,))
        return np.concatenate([A.ravel(), b.ravel()])

    # ---- API parity with the reference layer (evaluated on the device) ----
    def _ctx(self):
        from .. import _native
        if self._private is None:
            self._private = _native.Context(0)
        flat = np.concatenate([p._value.ravel() for p in self.parameters()] + [np.ones(1)])
        self._private.model_set([self.desc()], flat, self.mean_params())
        return self._private

    def conditional_ND(self, X, full_cov=False):
        """layers.py:236-278.  full_cov=True: mean [N, D], var [N, N, D] (the reference transposes [D, N, N])."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        if full_cov:
            _, Fm, Fv = self._ctx().propagate_full_cov(X, 1, 0, None)
            return as_tensor(Fm[0][0]), as_tensor(Fv[0][0])
        _, Fm, Fv = self._ctx().propagate(X, 1, 0, None, want=(False, True, True))
        return as_tensor(Fm[0][0]), as_tensor(Fv[0][0])

    def conditional_SND(self, X, full_cov=False):
        """layers.py:63-85: full_cov=True maps conditional_ND over the samples (var [S, N, N, D])."""
        S, N, D = X.shape
        if full_cov:
            ms, vs = zip(*[self.conditional_ND(np.asarray(X[s]), full_cov=True) for s in range(S)])
            return [as_tensor(np.stack([np.asarray(m) for m in ms])), as_tensor(np.stack([np.asarray(v) for v in vs]))]
        mean, var = self.conditional_ND(np.reshape(X, (S * N, D)))
        return [as_tensor(np.reshape(m, (S, N, self.num_outputs))) for m in (mean, var)]

    def sample_from_conditional(self, X, z=None, full_cov=False):
        """layers.py:87-130.  full_cov=True draws through the Cholesky factor of `var + jitter I` per (sample, output)
        (utils.py:43-51) on the device: one `dgp_propagate_full_cov` call per sample, each with its own inputs."""
        X = np.asarray(X, dtype=np.float64)
        S, N, _ = X.shape
        if z is None:
            z = np.random.standard_normal((S, N, self.num_outputs))
        z = np.ascontiguousarray(z, dtype=np.float64)
        if z.shape != (S, N, self.num_outputs):
            raise ValueError(f"z must be [S, N, D_out] = {(S, N, self.num_outputs)}, got {z.shape}")
        if full_cov:
            ctx = self._ctx()
            Fs, Fm, Fv = zip(*[[a[0][0] for a in ctx.propagate_full_cov(np.ascontiguousarray(X[s]), 1, 0, [z[s:s + 1]])]
                               for s in range(S)])
            return as_tensor(np.stack(Fs)), as_tensor(np.stack(Fm)), as_tensor(np.stack(Fv))
        mean, var = self.conditional_SND(X, full_cov=False)
        return as_tensor(mean + z * (var + JITTER) ** 0.5), mean, var

    def KL(self):
        ctx = self._ctx()
        ctx.data_set(np.zeros((1, self.input_dim)), np.zeros((1, self.num_outputs)))
        return ctx.elbo(1, 0, None)[1]
