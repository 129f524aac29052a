"""Shared helpers: rebuild models from the golden fixtures (oracle side and product side)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["case_A_nonwhite", "case_A_white", "case_B_nonwhite", "case_B_white"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def n_layers(g):
    return len(g["num_units"]) + 1


def notebook_data():
    """nb_DGP_regression.ipynb cells 2, 6, 10 regenerated from NumPy's legacy seed-0 stream."""
    np.random.seed(0)
    f_step = lambda x: 0. if x < 0.5 else 1.
    X = np.random.uniform(0, 1, 50)[:, None]
    Z = np.random.uniform(0, 1, 25)[:, None]
    Y = np.reshape([f_step(x) for x in X], X.shape) + np.random.randn(*X.shape) * 1e-2
    return X, Y, Z


def oracle_from_golden(g, prefix=""):
    import dgp_oracle as O
    dims = [g["X"].shape[1]] + list(g["num_units"])
    kernels = [O.RBF(1.0, np.ones(d)) for d in dims]
    m = O.OracleDGP(g["X"], g["Y"], g["Z_init"], kernels, list(g["num_units"]), lik_variance=1.0,
                    white=bool(g["white"]), num_samples=int(g["S"]))
    set_oracle_state(m, g, prefix)
    return m


def set_oracle_state(m, g, prefix=""):
    m.lik_variance = float(g[prefix + "lik_variance"])
    for i, l in enumerate(m.layers):
        l.Z = g[f"{prefix}L{i}_Z"].copy()
        l.kern.variance = float(g[f"{prefix}L{i}_variance"])
        l.kern.lengthscales = g[f"{prefix}L{i}_lengthscales"].copy()
        l.q_mu = g[f"{prefix}L{i}_q_mu"].copy()
        l.q_sqrt = g[f"{prefix}L{i}_q_sqrt"].copy()


def product_from_golden(g, prefix="", **kw):
    """Build the product DGP (HIP path) in the state stored in a golden fixture."""
    from dgp_dace.gpflow_compat import RBF, Gaussian
    from dgp_dace.models.dgp import DGP
    dims = [g["X"].shape[1]] + list(g["num_units"])
    kernels = [RBF(1.0, np.ones(d)) for d in dims]
    m = DGP(g["X"], g["Y"], g["Z_init"], kernels, list(g["num_units"]), Gaussian(), white=bool(g["white"]),
            num_samples=int(g["S"]), **kw)
    m.likelihood.likelihood.variance.assign(g[prefix + "lik_variance"])
    for i, l in enumerate(m.layers):
        l.feature.Z.assign(g[f"{prefix}L{i}_Z"])
        l.kern.variance.assign(g[f"{prefix}L{i}_variance"])
        l.kern.lengthscales.assign(g[f"{prefix}L{i}_lengthscales"])
        l.q_mu.assign(g[f"{prefix}L{i}_q_mu"])
        l.q_sqrt.assign(g[f"{prefix}L{i}_q_sqrt"])
    return m


def split_flat(m, flat):
    """flat vector in dgp_model_set packing -> {(layer, name): array}"""
    out, off = {}, 0
    names = ["Z", "variance", "lengthscales", "q_mu", "q_sqrt"]
    for i, l in enumerate(m.layers):
        for nm, p in zip(names, l.parameters()):
            n = p._value.size
            out[(i, nm)] = flat[off:off + n].reshape(p._value.shape)
            off += n
    out[("lik", "variance")] = flat[off]
    return out


def stationary_kernel(A, B, variance, lengthscales, kind="rbf"):
    """k(a, b) for the ARD squared-exponential, Matern-3/2 and Matern-5/2 kernels, from their published formulas
    (Rasmussen & Williams eq. 4.9, 4.17); r is clamped away from 0 as gpflow does (r^2 >= 1e-36)."""
    ls = np.asarray(lengthscales, dtype=float)
    A = A / ls
    B = B / ls
    d2 = np.maximum((A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * A @ B.T, 0.0)
    if kind == "rbf":
        return variance * np.exp(-0.5 * d2)
    r = np.sqrt(np.maximum(d2, 1e-36))
    if kind == "matern32":
        return variance * (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)
    if kind == "matern52":
        return variance * (1.0 + np.sqrt(5.0) * r + 5.0 / 3.0 * d2) * np.exp(-np.sqrt(5.0) * r)
    raise ValueError(kind)


def collapsed_bound(X, Y, Z, variance, lengthscales, noise, jitter, kind="rbf"):
    """Titsias' collapsed bound of sparse GP regression with a stationary ARD kernel (`kind`) and Gaussian noise, and the q(u) that
    attains it - textbook closed forms, independent of the oracle and of the reference's code:
        bound = log N(y | 0, Qff + noise I) - tr(Kff - Qff) / (2 noise),   Qff = Kfu (Kuu + jitter I)^-1 Kuf,
        S* = Kuu (Kuu + Kuf Kfu / noise)^-1 Kuu,   m* = S* Kuu^-1 Kuf y / noise.
    A one-layer DGP (num_units = []) IS that model (dgp.py:89-100 with the Gaussian variational expectations on the only layer,
    non-white KL of layers.py:293-300), and ONE natural-gradient step of size 1 from any q(u) lands on (m*, S*): the bound is
    quadratic in q's natural parameters.  Returns (bound, m* [M, 1], S* [M, M])."""
    import scipy.linalg as sla
    def k(A, B):
        return stationary_kernel(A, B, variance, lengthscales, kind)
    N, M = X.shape[0], Z.shape[0]
    L = np.linalg.cholesky(k(Z, Z) + jitter * np.eye(M))
    A = sla.solve_triangular(L, k(Z, X), lower=True) / np.sqrt(noise)          # [M, N]
    LB = np.linalg.cholesky(np.eye(M) + A @ A.T)
    c = sla.solve_triangular(LB, A @ Y, lower=True) / np.sqrt(noise)           # [M, Dy]
    Dy = Y.shape[1]
    bound = Dy * (-0.5 * N * np.log(2.0 * np.pi * noise) - np.log(np.diag(LB)).sum()) - 0.5 * (Y * Y).sum() / noise + 0.5 * (c * c).sum()
    bound += -0.5 * Dy * (N * variance - noise * (A * A).sum()) / noise
    R = L @ sla.solve_triangular(LB.T, np.eye(M), lower=False)                 # S* = R R^T
    return float(bound), L @ sla.solve_triangular(LB.T, c, lower=False), R @ R.T


def sparse_gp_predict(X, Y, Z, Xnew, variance, lengthscales, noise, jitter, kind="rbf"):
    """Predictive mean and variance of f at Xnew under the optimal q(u) of sparse GP regression (Titsias 2009, eq. 6):
        mean = K*u Sigma Kuf y / noise,  var = k** - K*u Kuu^-1 Ku* + K*u Sigma Ku*,  Sigma = (Kuu + Kuf Kfu / noise)^-1.
    Textbook closed form (same kernel and jitter convention as collapsed_bound)."""
    import scipy.linalg as sla
    def k(A, B):
        return stationary_kernel(A, B, variance, lengthscales, kind)
    M = Z.shape[0]
    L = np.linalg.cholesky(k(Z, Z) + jitter * np.eye(M))
    A = sla.solve_triangular(L, k(Z, X), lower=True) / np.sqrt(noise)
    LB = np.linalg.cholesky(np.eye(M) + A @ A.T)
    c = sla.solve_triangular(LB, A @ Y, lower=True) / np.sqrt(noise)
    As = sla.solve_triangular(L, k(Z, Xnew), lower=True)                  # L^-1 Ku*
    Bs = sla.solve_triangular(LB, As, lower=True)                         # LB^-1 L^-1 Ku*
    return Bs.T @ c, (variance - (As * As).sum(0) + (Bs * Bs).sum(0))[:, None] * np.ones((1, Y.shape[1]))


def svgp_elbo(X, Y, Z, variance, lengthscales, noise, q_mu, q_sqrt, jitter, kind="rbf", white=False):
    """The SVGP regression bound at an ARBITRARY q(u) = N(q_mu, q_sqrt q_sqrt^T) (Hensman et al. 2013, eq. 4), from the textbook:
        sum_n [ log N(y_n | mu_n, noise) - v_n / (2 noise) ] - sum_d KL[ N(m_d, S_d) || N(0, Kuu) ],
        mu = Kfu Kuu^-1 m,   v_n = k_nn - (Kfu Kuu^-1 Kuf)_nn + (Kfu Kuu^-1 S Kuu^-1 Kuf)_nn;
    white=True: q is over v with u = chol(Kuu) v.  q_mu [M, Dy], q_sqrt [Dy, M, M] (lower triangles used)."""
    import scipy.linalg as sla
    N, M, Dy = X.shape[0], Z.shape[0], Y.shape[1]
    Kuu = stationary_kernel(Z, Z, variance, lengthscales, kind) + jitter * np.eye(M)
    Kuf = stationary_kernel(Z, X, variance, lengthscales, kind)
    L = np.linalg.cholesky(Kuu)
    A = sla.solve_triangular(L, Kuf, lower=True)                       # L^-1 Kuf
    total = 0.0
    for d in range(Dy):
        Lq = np.tril(q_sqrt[d])
        if white:
            mv, Lv = q_mu[:, d], Lq                                   # v-space
        else:
            mv = sla.solve_triangular(L, q_mu[:, d], lower=True)      # v = L^-1 u
            Lv = sla.solve_triangular(L, Lq, lower=True)
        mu = A.T @ mv
        B = Lv.T @ A                                                   # [M, N]: (L^-1 Lq)^T L^-1 Kuf
        v = variance - (A * A).sum(0) + (B * B).sum(0)
        total += (-0.5 * np.log(2 * np.pi * noise) - 0.5 * ((Y[:, d] - mu) ** 2 + v) / noise).sum()
        # KL[N(mv, Lv Lv^T) || N(0, I)] in v-space (equal to the u-space KL against N(0, Kuu))
        total -= 0.5 * ((Lv * Lv).sum() + mv @ mv - M) - np.log(np.abs(np.diag(Lv))).sum()
    return float(total)


def svgp_moments(Xin, Z, variance, lengthscales, q_mu, q_sqrt, jitter, kind="rbf"):
    """Marginal mean / variance of the SVGP posterior at Xin (Hensman et al. 2013, eq. 3), one column per output, and the KL:
        mu = Kfu Kuu^-1 m,  v = k_nn - diag(Kfu Kuu^-1 Kuf) + diag(Kfu Kuu^-1 S Kuu^-1 Kuf),  KL = sum_d KL[N(m_d, S_d) || N(0, Kuu)]."""
    import scipy.linalg as sla
    M, Dy = Z.shape[0], q_mu.shape[1]
    L = np.linalg.cholesky(stationary_kernel(Z, Z, variance, lengthscales, kind) + jitter * np.eye(M))
    A = sla.solve_triangular(L, stationary_kernel(Z, Xin, variance, lengthscales, kind), lower=True)
    mu, v, kl = np.empty((Xin.shape[0], Dy)), np.empty((Xin.shape[0], Dy)), 0.0
    for d in range(Dy):
        mv = sla.solve_triangular(L, q_mu[:, d], lower=True)
        Lv = sla.solve_triangular(L, np.tril(q_sqrt[d]), lower=True)
        B = Lv.T @ A
        mu[:, d] = A.T @ mv
        v[:, d] = variance - (A * A).sum(0) + (B * B).sum(0)
        kl += 0.5 * ((Lv * Lv).sum() + mv @ mv - M) - np.log(np.abs(np.diag(Lv))).sum()
    return mu, v, float(kl)


def dsdgp2_elbo(X, Y, z1, layer1, layer2, noise, jitter):
    """Doubly-stochastic bound of a two-layer DGP (Salimbeni & Deisenroth 2017, eq. 13-16) for GIVEN hidden-layer normals z1 [S, N, D1],
    identity mean function on the hidden layer, zero on the output layer; layer = dict(Z, variance, lengthscales, q_mu, q_sqrt):
        F1[s] = X + mu1(X) + sqrt(v1(X) + jitter) z1[s];   ELBO = (1/S) sum_s sum_n E_q2[log N(y_n | f2(F1[s]_n), noise)] - KL1 - KL2."""
    S = z1.shape[0]
    mu1, v1, kl1 = svgp_moments(X, layer1["Z"], layer1["variance"], layer1["lengthscales"], layer1["q_mu"], layer1["q_sqrt"], jitter)
    total, kl2 = 0.0, 0.0
    for s in range(S):
        F1 = X + mu1 + np.sqrt(v1 + jitter) * z1[s]
        mu2, v2, kl2 = svgp_moments(F1, layer2["Z"], layer2["variance"], layer2["lengthscales"], layer2["q_mu"], layer2["q_sqrt"], jitter)
        total += (-0.5 * np.log(2 * np.pi * noise) - 0.5 * ((Y - mu2) ** 2 + v2) / noise).sum()
    return float(total / S - kl1 - kl2)


def dsdgp_elbo(X, Y, zs, layers, noise, jitter):
    """The doubly-stochastic bound for any depth (Salimbeni & Deisenroth 2017, eq. 13-16), GIVEN the normals of every hidden layer
    (zs[l]: [S, N, D_l]); identity mean function on the hidden layers (equal widths), zero on the last; layers[l] = dict(Z, variance,
    lengthscales, q_mu, q_sqrt).  The first layer's marginals do not depend on the sample (dgp.py:49 tiles X first)."""
    S = zs[0].shape[0]
    kls = []
    mu0, v0, kl0 = svgp_moments(X, layers[0]["Z"], layers[0]["variance"], layers[0]["lengthscales"], layers[0]["q_mu"], layers[0]["q_sqrt"], jitter)
    kls.append(kl0)
    if len(layers) == 1:
        return float((-0.5 * np.log(2 * np.pi * noise) - 0.5 * ((Y - mu0) ** 2 + v0) / noise).sum() - kl0)
    total = 0.0
    for s in range(S):
        F = X + mu0 + np.sqrt(v0 + jitter) * zs[0][s]
        for l in range(1, len(layers)):
            L = layers[l]
            mu, v, kl = svgp_moments(F, L["Z"], L["variance"], L["lengthscales"], L["q_mu"], L["q_sqrt"], jitter)
            if s == 0:
                kls.append(kl)
            if l < len(layers) - 1:
                F = F + mu + np.sqrt(v + jitter) * zs[l][s]
        total += (-0.5 * np.log(2 * np.pi * noise) - 0.5 * ((Y - mu) ** 2 + v) / noise).sum()
    return float(total / S - sum(kls))


def sparse_gp_predict_full_cov(X, Y, Z, Xnew, variance, lengthscales, noise, jitter, kind="rbf"):
    """Full predictive covariance of f at Xnew under the optimal q(u) (Titsias 2009, eq. 6): K** - K*u Kuu^-1 Ku* + K*u Sigma Ku*."""
    import scipy.linalg as sla

    def k(A, B):
        return stationary_kernel(A, B, variance, lengthscales, kind)
    M = Z.shape[0]
    L = np.linalg.cholesky(k(Z, Z) + jitter * np.eye(M))
    A = sla.solve_triangular(L, k(Z, X), lower=True) / np.sqrt(noise)
    LB = np.linalg.cholesky(np.eye(M) + A @ A.T)
    As = sla.solve_triangular(L, k(Z, Xnew), lower=True)
    Bs = sla.solve_triangular(LB, As, lower=True)
    return k(Xnew, Xnew) - As.T @ As + Bs.T @ Bs
