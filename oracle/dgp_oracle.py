"""CPU oracle for the doubly-stochastic DGP / SVGP-layer ELBO path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  The product path (``dgp-toolbox_amd/``) never imports this module.

What it is: a NumPy fp64 restatement, op for op (dense ``SK @ A`` form, triangular solves,
tiled ``[D_out, M, P]`` intermediates), of the reference files

* ``dgp_dace/utils/layers.py:47-130,180-308``   (Layer / SVGP_Layer)
* ``dgp_dace/utils/utils.py:22-117``            (reparameterize, BroadcastingLikelihood)
* ``dgp_dace/utils/layer_initializations.py:24-68`` (init_layers_linear)
* ``dgp_dace/models/dgp.py:21-366``             (DGP_Base / DGP)

plus the behaviour of the third-party GPflow 2.x pieces those files call (kernels,
``covariances.Kuu/Kuf``, ``likelihoods.Gaussian``, mean functions, ``NaturalGradient``,
Keras ``Adam``) restated from their published definitions.  GPflow,
TensorFlow and TFP are not installed in the build container and cannot be fetched; their
version is unpinned by the reference (README says "GPflow 2.0").

Parity pin: the reference has no tests.  The oracle is pinned by the three known answers its
notebooks store - ``ELBO == -85.98812279560475`` for the freshly built model and
``number_parameters(trainable=False) == 2032`` (``Notebooks_dgp/nb_DGP_regression.ipynb`` cells
22/26 and 30), ``ELBO: -73.6722504558447`` (``nb_dgp_BO.ipynb`` cells 30/61: the non-white KL at
q != prior through ``SO_BO``) - and, since round 4, by closed forms written from the literature in
``tests/helpers.py`` (not from the reference's code): Titsias' collapsed bound, its optimal q(u)
and predictive equations (one natural-gradient step of size one must land there), the SVGP bound
at an arbitrary q(u) (Hensman et al. 2013) and the two-layer doubly-stochastic bound for given
normals (Salimbeni & Deisenroth 2017), each with its central differences for the gradients
(tests/test_oracle.py).  Multi-iteration trajectories rest on self-consistency (NumPy restatement vs
torch-autograd twin); the product's single Adam / natural-gradient steps are checked against their
published formulas in tests/test_gpu_parity.py.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

JITTER = 1e-6          # gpflow.default_jitter()  [ext]
LIK_VAR_LOWER = 1e-6   # gpflow.likelihoods.Gaussian DEFAULT_VARIANCE_LOWER_BOUND  [ext]


# --------------------------------------------------------------------------------------
# GPflow stand-ins  [ext]
# --------------------------------------------------------------------------------------
class RBF:
    """gpflow.kernels.SquaredExponential (ARD).  K = s2 * exp(-0.5 r2), r2 by the expanded form."""

    kind = "rbf"

    def __init__(self, variance=1.0, lengthscales=1.0):
        self.variance = float(variance)
        self.lengthscales = np.atleast_1d(np.asarray(lengthscales, dtype=np.float64)).copy()

    def _scaled(self, X):
        return X / self.lengthscales

    def _r2(self, X, X2=None):
        """gpflow.utilities.ops.square_distance of the scaled inputs (expanded form, no clamp)  [ext]"""
        Xs = self._scaled(X)
        if X2 is None:
            sq = np.sum(Xs * Xs, -1)
            return -2.0 * Xs @ Xs.T + sq[:, None] + sq[None, :]
        X2s = self._scaled(X2)
        return -2.0 * Xs @ X2s.T + np.sum(Xs * Xs, -1)[:, None] + np.sum(X2s * X2s, -1)[None, :]

    def K(self, X, X2=None):
        return self.variance * np.exp(-0.5 * self._r2(X, X2))

    def K_diag(self, X):
        return np.full(X.shape[0], self.variance)

    def copy(self):
        return type(self)(self.variance, self.lengthscales.copy())


class Matern32(RBF):
    """gpflow.kernels.Matern32 (SO_BO.py:194-195,241-242): s2 (1 + sqrt3 r) exp(-sqrt3 r), r = sqrt(max(r2, 1e-36))
    (gpflow IsotropicStationary.scaled_squared_euclid_dist / K_r; the clamp keeps the sqrt differentiable)  [ext]"""

    kind = "matern32"

    def K(self, X, X2=None):
        r = np.sqrt(np.maximum(self._r2(X, X2), 1e-36))
        a = np.sqrt(3.0) * r
        return self.variance * (1.0 + a) * np.exp(-a)


class Matern52(RBF):
    """gpflow.kernels.Matern52 (SO_BO.py:196-197,243-244): s2 (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)  [ext]"""

    kind = "matern52"

    def K(self, X, X2=None):
        r = np.sqrt(np.maximum(self._r2(X, X2), 1e-36))
        a = np.sqrt(5.0) * r
        return self.variance * (1.0 + a + 5.0 / 3.0 * r * r) * np.exp(-a)


class MeanFunction:
    """kind in {'zero','identity','linear'}; Linear(A, b) with b = zeros(1) when not given  [ext]."""

    def __init__(self, kind="zero", A=None, b=None):
        self.kind = kind
        self.A = None if A is None else np.asarray(A, dtype=np.float64)
        self.b = np.zeros(1) if (kind == "linear" and b is None) else b

    def __call__(self, X):
        if self.kind == "zero":
            return 0.0
        if self.kind == "identity":
            return X
        return X @ self.A + self.b


class OracleLayer:
    """State of one reference ``SVGP_Layer`` (layers.py:181-224)."""

    def __init__(self, kern, Z, num_outputs, mean_function, white=False):
        Z = np.asarray(Z, dtype=np.float64)
        self.num_inducing = Z.shape[0]
        self.num_outputs = int(num_outputs)
        self.q_mu = np.zeros((self.num_inducing, num_outputs))                      # :203-204
        self.q_sqrt = np.tile(np.eye(self.num_inducing)[None], [num_outputs, 1, 1])  # :205
        self.Z = Z.copy()                                                             # :209
        self.kern = kern
        self.mean_function = mean_function
        self.white = bool(white)
        if not self.white:                                                            # :219-223
            Ku = kern.K(Z)
            Lu = np.linalg.cholesky(Ku + np.eye(Z.shape[0]) * JITTER)
            self.q_sqrt = np.tile(Lu[None], [num_outputs, 1, 1])

    # layers.py:227-234
    def build_cholesky(self):
        self.Ku = self.kern.K(self.Z) + JITTER * np.eye(self.num_inducing)
        self.Lu = np.linalg.cholesky(self.Ku)

    # layers.py:237-278
    def conditional_ND(self, X, full_cov=False):
        self.build_cholesky()
        Kuf = self.kern.K(self.Z, X)                                     # :243  [M,P]
        A = sla.solve_triangular(self.Lu, Kuf, lower=True)               # :245
        if not self.white:
            A = sla.solve_triangular(self.Lu.T, A, lower=False)          # :247
        mean = A.T @ self.q_mu                                           # :249
        A_tiled = np.tile(A[None], [self.num_outputs, 1, 1])             # :251
        I = np.eye(self.num_inducing)[None]
        SK = -I if self.white else -np.tile(self.Ku[None], [self.num_outputs, 1, 1])
        SK = SK + self.q_sqrt @ np.transpose(self.q_sqrt, (0, 2, 1))     # :260
        B = SK @ A_tiled                                                 # :263
        if full_cov:
            delta_cov = np.transpose(A_tiled, (0, 2, 1)) @ B             # :267  [D,P,P]
            Kff = self.kern.K(X)                                         # :268
            var = np.transpose(Kff[None] + delta_cov)                    # :275-276  [P,P,D]
            return mean + self.mean_function(X), var
        delta_cov = np.sum(A_tiled * B, 1)                               # :271
        Kff = self.kern.K_diag(X)                                        # :272
        var = (Kff[None] + delta_cov).T                                  # :275-276
        return mean + self.mean_function(X), var                         # :278

    # layers.py:63-85
    def conditional_SND(self, X, full_cov=False):
        S, N, D = X.shape
        if full_cov:                                                     # :77-80: one conditional per sample
            out = [self.conditional_ND(X[s], full_cov=True) for s in range(S)]
            return np.stack([m for m, _ in out]), np.stack([v for _, v in out])       # [S,N,D], [S,N,N,D]
        mean, var = self.conditional_ND(X.reshape(S * N, D))
        return mean.reshape(S, N, self.num_outputs), var.reshape(S, N, self.num_outputs)

    # layers.py:87-130 (input_prop_dim unused by DGP)
    def sample_from_conditional(self, X, z, full_cov=False):
        mean, var = self.conditional_SND(X, full_cov=full_cov)
        if full_cov:                                                     # utils.py:43-51
            S, N, D = mean.shape
            v = np.transpose(var, (0, 3, 1, 2)) + JITTER * np.eye(N)[None, None]      # SDNN
            chol = np.linalg.cholesky(v)
            f = np.transpose(mean, (0, 2, 1)) + (chol @ np.transpose(z, (0, 2, 1))[..., None])[..., 0]
            return np.transpose(f, (0, 2, 1)), mean, var
        samples = mean + z * (var + JITTER) ** 0.5                       # utils.py:41
        return samples, mean, var

    # layers.py:280-308
    def KL(self):
        self.build_cholesky()
        KL = -0.5 * self.num_outputs * self.num_inducing
        diag = np.diagonal(self.q_sqrt, axis1=1, axis2=2)
        KL -= 0.5 * np.sum(np.log(diag ** 2))
        if not self.white:
            KL += np.sum(np.log(np.diag(self.Lu))) * self.num_outputs
            LiS = np.stack([sla.solve_triangular(self.Lu, self.q_sqrt[d], lower=True)
                            for d in range(self.num_outputs)])
            KL += 0.5 * np.sum(LiS ** 2)
            Kinv_m = sla.cho_solve((self.Lu, True), self.q_mu)
            KL += 0.5 * np.sum(self.q_mu * Kinv_m)
        else:
            KL += 0.5 * np.sum(self.q_sqrt ** 2)
            KL += 0.5 * np.sum(self.q_mu ** 2)
        return KL


# layer_initializations.py:24-68
def init_layers_linear(X, Y, Z, kernels, num_units, num_outputs=None, mean_function=None, white=False):
    num_outputs = num_outputs or Y.shape[1]
    mean_function = mean_function or MeanFunction("zero")
    layers = []
    num_units = [X.shape[1]] + list(num_units)
    X_running, Z_running = X.copy(), Z.copy()
    for dim_in, dim_out, kern_in in zip(num_units[:-1], num_units[1:], kernels[:-1]):
        if dim_in == dim_out:
            mf = MeanFunction("identity")
            W = None
        else:
            if dim_in > dim_out:
                _, _, V = np.linalg.svd(X_running, full_matrices=False)
                W = V[:dim_out, :].T
            else:
                W = np.concatenate([np.eye(dim_in), np.zeros((dim_in, dim_out - dim_in))], 1)
            mf = MeanFunction("linear", A=W)
        layers.append(OracleLayer(kern_in, Z_running, dim_out, mf, white=white))
        if dim_in != dim_out:
            Z_running = Z_running.dot(W)
            X_running = X_running.dot(W)
    layers.append(OracleLayer(kernels[-1], Z_running, num_outputs, mean_function, white=white))
    return layers


class OracleDGP:
    """Reference ``DGP`` (dgp.py:221-254) with a Gaussian likelihood of variance ``lik_variance``."""

    def __init__(self, X, Y, Z, kernels, num_units, lik_variance=1.0, white=False, num_samples=1,
                 num_outputs=None, mean_function=None):
        self.layers = init_layers_linear(X, Y, Z, kernels, num_units, num_outputs=num_outputs,
                                         mean_function=mean_function, white=white)
        self.lik_variance = float(lik_variance)
        self.num_samples = num_samples
        self.data = (np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64))

    # dgp.py:34-63
    def propagate(self, X, S, zs, full_cov=False):
        F = np.tile(X[None], [S, 1, 1])
        Fs, Fmeans, Fvars = [], [], []
        for layer, z in zip(self.layers, zs):
            F, Fmean, Fvar = layer.sample_from_conditional(F, z, full_cov=full_cov)
            Fs.append(F), Fmeans.append(Fmean), Fvars.append(Fvar)
        return Fs, Fmeans, Fvars

    def predict_f(self, X, S, zs):
        Fs, Fmeans, Fvars = self.propagate(X, S, zs)
        return Fmeans[-1], Fvars[-1]

    # gpflow Gaussian.variational_expectations  [ext]
    def variational_expectations(self, Fmu, Fvar, Y):
        s2 = self.lik_variance
        return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(s2) - 0.5 * ((Y - Fmu) ** 2 + Fvar) / s2

    # dgp.py:79-87
    def E_log_p_Y(self, X, Y, zs):
        Fmean, Fvar = self.predict_f(X, self.num_samples, zs)
        return np.mean(self.variational_expectations(Fmean, Fvar, Y[None]), 0)

    # dgp.py:89-100 (scale == 1)
    def ELBO(self, zs, data=None):
        X, Y = data if data is not None else self.data
        L = np.sum(self.E_log_p_Y(X, Y, zs))
        KL = np.sum([layer.KL() for layer in self.layers])
        return L - KL

    def elbo_terms(self, zs, data=None):
        X, Y = data if data is not None else self.data
        return np.sum(self.E_log_p_Y(X, Y, zs)), [layer.KL() for layer in self.layers]

    # dgp.py:113-124; Gaussian.predict_mean_and_var = (mu, var + s2)  [ext]
    def predict_y(self, Xnew, num_samples, zs):
        Fmean, Fvar = self.predict_f(Xnew, num_samples, zs)
        return Fmean, Fvar + self.lik_variance

    # dgp.py:362-366
    def predict(self, Xnew, num_samples, zs):
        y_m, y_v = self.predict_y(Xnew, num_samples, zs)
        mean = np.mean(y_m, axis=0)
        return mean, np.mean(y_v + y_m ** 2, 0) - mean ** 2

    # dgp.py:347-360 with gpflow Parameter sizes (constrained arrays; Linear mean fn has A and b)
    def number_parameters(self):
        n = 1  # likelihood variance
        for l in self.layers:
            n += l.q_mu.size + l.q_sqrt.size + l.Z.size + 1 + l.kern.lengthscales.size
            if l.mean_function.kind == "linear":
                n += l.mean_function.A.size + np.size(l.mean_function.b)
        return n


# --------------------------------------------------------------------------------------
# Counter-based normals (shared definition with the HIP kernels; NOT in the reference, whose
# tf.random.normal stream cannot be reproduced — SURVEY.md §8d).  Philox4x32-10 keyed by the
# 64-bit evaluation seed, counter = (n_lo, n_hi, s, layer<<16 | d); Box–Muller cosine branch.
# --------------------------------------------------------------------------------------
_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & _MASK32 for c in (c0, c1, c2, c3)]
    k0 = np.uint64(k0) & _MASK32
    k1 = np.uint64(k1) & _MASK32
    for _ in range(10):
        p0 = _PH_M0 * c0
        p1 = _PH_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0), lo1, (hi0 ^ c3 ^ k1), lo0
        k0 = (k0 + _PH_W0) & _MASK32
        k1 = (k1 + _PH_W1) & _MASK32
    return c0, c1, c2, c3


def philox_normal(seed, layer, S, n_index, D):
    """z[s, i, d] for global point indices ``n_index`` (int64 array), layer ``layer``."""
    n_index = np.asarray(n_index, dtype=np.uint64)
    s = np.arange(S, dtype=np.uint64)[:, None, None]
    n = n_index[None, :, None]
    d = np.arange(D, dtype=np.uint64)[None, None, :]
    shape = (S, n_index.size, D)
    c0 = np.broadcast_to(n & _MASK32, shape)
    c1 = np.broadcast_to(n >> np.uint64(32), shape)
    c2 = np.broadcast_to(s, shape)
    c3 = np.broadcast_to((np.uint64(layer) << np.uint64(16)) | d, shape)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    r0, r1, r2, r3 = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, seed >> 32)
    two53 = 9007199254740992.0
    u1 = (((r0 >> np.uint64(5)) * np.uint64(67108864) + (r1 >> np.uint64(6))).astype(np.float64) + 0.5) / two53
    u2 = (((r2 >> np.uint64(5)) * np.uint64(67108864) + (r3 >> np.uint64(6))).astype(np.float64) + 0.5) / two53
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def draw_zs(model, seed, S, N, n_offset=0):
    idx = np.arange(n_offset, n_offset + N, dtype=np.int64)
    return [philox_normal(seed, li, S, idx, l.num_outputs) for li, l in enumerate(model.layers)]


# --------------------------------------------------------------------------------------
# Parameter transforms / optimisers  [ext]
# --------------------------------------------------------------------------------------
def softplus(u):
    return np.logaddexp(0.0, u)


def softplus_inv(x):
    x = np.asarray(x, dtype=np.float64)
    return x + np.log(-np.expm1(-x))


def adam_update(u, m, v, g, t, lr, beta_1, beta_2, epsilon):
    """One Keras-Adam step (TF 2.x ``tf.optimizers.Adam``) on unconstrained variables; t = 1,2,…"""
    m[...] = beta_1 * m + (1 - beta_1) * g
    v[...] = beta_2 * v + (1 - beta_2) * g * g
    lr_t = lr * np.sqrt(1 - beta_2 ** t) / (1 - beta_1 ** t)
    u[...] = u - lr_t * m / (np.sqrt(v) + epsilon)


def chol_backward_to_sigma(L, Lbar):
    """G = sym(dloss/dSigma) from the lower-triangular gradient w.r.t. L = chol(Sigma) (App. B)."""
    Phi = np.tril(L.T @ np.tril(Lbar))
    Phi[np.diag_indices_from(Phi)] *= 0.5
    Li = sla.solve_triangular(L, np.eye(L.shape[0]), lower=True)
    G = Li.T @ Phi @ Li
    return 0.5 * (G + G.T)


def natgrad_step(q_mu, q_sqrt, g_mu, g_sqrt, gamma):
    """Closed-form XiNat natural-gradient step for one layer.

    q_mu [M,D], q_sqrt [D,M,M]; g_* = d loss / d (q_mu, q_sqrt) with loss = -ELBO.
    Returns the new (q_mu, q_sqrt)."""
    M, D = q_mu.shape
    mu_new = np.empty_like(q_mu)
    sq_new = np.empty_like(q_sqrt)
    for d in range(D):
        L = np.tril(q_sqrt[d])
        G = chol_backward_to_sigma(L, g_sqrt[d])
        Li = sla.solve_triangular(L, np.eye(M), lower=True)
        Sinv = Li.T @ Li
        Pn = Sinv + 2.0 * gamma * G
        R = np.linalg.cholesky(Pn)                       # var_sqrt_inv = chol(-2 nat2)   [ext]
        Ri = sla.solve_triangular(R, np.eye(M), lower=True)
        Sn = Ri.T @ Ri
        mu_new[:, d] = q_mu[:, d] - gamma * Sn @ g_mu[:, d]
        sq_new[d] = np.linalg.cholesky(Sn)
    return mu_new, sq_new
