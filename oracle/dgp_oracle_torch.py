"""torch-CPU fp64 autograd twin of ``oracle/dgp_oracle.py``.

TEST INFRASTRUCTURE ONLY (see the header of dgp_oracle.py): it exists to (a) provide gradients
of the reference's ELBO the way ``tf.GradientTape`` does (dgp.py:272-275) so the hand-derived HIP
backward can be checked, (b) restate ``gpflow.optimizers.NaturalGradient`` by its autodiff route
(expectation parameters) so the closed form the product uses can be checked, and (c) serve as the
``cpu_baseline`` ("port") leg of bench.py: the same op sequence TF-CPU would execute for
layers.py:243-276 (materialised Kuf, triangular solves, dense batched SK @ A), chunked over points.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from dgp_oracle import JITTER, LIK_VAR_LOWER

DT = torch.float64


def _t(x, grad=True):
    return torch.tensor(np.asarray(x, dtype=np.float64), dtype=DT, requires_grad=grad)


def params_from_model(model):
    """Constrained parameters of an OracleDGP as torch leaves."""
    P = {"lik_variance": _t(model.lik_variance), "layers": []}
    for l in model.layers:
        P["layers"].append({
            "Z": _t(l.Z), "variance": _t(l.kern.variance), "lengthscales": _t(l.kern.lengthscales),
            "q_mu": _t(l.q_mu), "q_sqrt": _t(l.q_sqrt),
        })
    return P


def rbf_K(variance, lengthscales, X, X2=None, kind="rbf"):
    """Stationary kernels of dgp_oracle.py (RBF / Matern32 / Matern52) on torch tensors."""
    Xs = X / lengthscales
    if X2 is None:
        sq = (Xs * Xs).sum(-1)
        r2 = -2.0 * Xs @ Xs.T + sq[:, None] + sq[None, :]
    else:
        X2s = X2 / lengthscales
        r2 = -2.0 * Xs @ X2s.T + (Xs * Xs).sum(-1)[:, None] + (X2s * X2s).sum(-1)[None, :]
    if kind == "rbf":
        return variance * torch.exp(-0.5 * r2)
    r = torch.sqrt(torch.clamp(r2, min=1e-36))
    if kind == "matern32":
        a = math.sqrt(3.0) * r
        return variance * (1.0 + a) * torch.exp(-a)
    if kind == "matern52":
        a = math.sqrt(5.0) * r
        return variance * (1.0 + a + 5.0 / 3.0 * r * r) * torch.exp(-a)
    raise ValueError(kind)


def mean_fn(layer, X):
    mf = layer.mean_function
    if mf.kind == "zero":
        return 0.0
    if mf.kind == "identity":
        return X
    return X @ torch.as_tensor(mf.A, dtype=DT) + torch.as_tensor(np.asarray(mf.b), dtype=DT)


def conditional_ND(layer, p, X):
    """layers.py:237-278, dense form."""
    M, D = p["q_mu"].shape
    Ku = rbf_K(p["variance"], p["lengthscales"], p["Z"], kind=layer.kern.kind) + JITTER * torch.eye(M, dtype=DT)
    Lu = torch.linalg.cholesky(Ku)
    Kuf = rbf_K(p["variance"], p["lengthscales"], p["Z"], X, kind=layer.kern.kind)
    A = torch.linalg.solve_triangular(Lu, Kuf, upper=False)
    if not layer.white:
        A = torch.linalg.solve_triangular(Lu.T, A, upper=True)
    mean = A.T @ p["q_mu"]
    q_sqrt = torch.tril(p["q_sqrt"])       # FillTriangular: the upper triangle is not a variable
    SK = q_sqrt @ q_sqrt.transpose(1, 2)
    SK = SK - (torch.eye(M, dtype=DT)[None] if layer.white else Ku[None])
    B = SK @ A[None].expand(D, -1, -1)
    var = (p["variance"] + (A[None] * B).sum(1)).T
    return mean + mean_fn(layer, X), var


def layer_KL(layer, p):
    """layers.py:280-308."""
    M, D = p["q_mu"].shape
    q_sqrt = torch.tril(p["q_sqrt"])
    KL = -0.5 * D * M - 0.5 * torch.log(torch.diagonal(q_sqrt, dim1=1, dim2=2) ** 2).sum()
    if not layer.white:
        Ku = rbf_K(p["variance"], p["lengthscales"], p["Z"], kind=layer.kern.kind) + JITTER * torch.eye(M, dtype=DT)
        Lu = torch.linalg.cholesky(Ku)
        KL = KL + torch.log(torch.diagonal(Lu)).sum() * D
        LiS = torch.linalg.solve_triangular(Lu[None].expand(D, -1, -1), q_sqrt, upper=False)
        KL = KL + 0.5 * (LiS ** 2).sum()
        KL = KL + 0.5 * (p["q_mu"] * torch.cholesky_solve(p["q_mu"], Lu)).sum()
    else:
        KL = KL + 0.5 * (q_sqrt ** 2).sum() + 0.5 * (p["q_mu"] ** 2).sum()
    return KL


def data_term(model, P, X, Y, zs, S):
    """sum_n mean_s VE   (dgp.py:79-87,96); X,Y torch [N,*]; zs list of torch [S,N,D_l]."""
    N = X.shape[0]
    F = X[None].expand(S, -1, -1).reshape(S * N, -1)
    for layer, p, z in zip(model.layers, P["layers"], zs):
        mean, var = conditional_ND(layer, p, F)
        F = mean + z.reshape(S * N, -1) * (var + JITTER) ** 0.5
    s2 = P["lik_variance"]
    Yt = Y[None].expand(S, -1, -1).reshape(S * N, -1)
    ve = -0.5 * math.log(2 * math.pi) - 0.5 * torch.log(s2) - 0.5 * ((Yt - mean) ** 2 + var) / s2
    return ve.sum() / S


def elbo_and_grads(model, zs, S=None, data=None, chunk=None, P=None, want_grads=True):
    """ELBO (dgp.py:89-100) and d ELBO / d(constrained params) as numpy, by autograd.

    ``chunk`` = number of data points per block (numerically neutral; bounds the
    [D_out, M, S*chunk] intermediates the dense formulation materialises)."""
    X, Y = data if data is not None else model.data
    S = S or model.num_samples
    P = P or params_from_model(model)
    Xt, Yt = torch.as_tensor(X, dtype=DT), torch.as_tensor(Y, dtype=DT)
    zs_t = [torch.as_tensor(z, dtype=DT) for z in zs]
    N = X.shape[0]
    chunk = chunk or N
    total = 0.0
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        L = data_term(model, P, Xt[a:b], Yt[a:b], [z[:, a:b] for z in zs_t], S)
        if want_grads:
            L.backward()
        total += float(L.detach())
    KL = sum(layer_KL(l, p) for l, p in zip(model.layers, P["layers"]))
    if want_grads:
        (-KL).backward()
    elbo = total - float(KL.detach())
    if not want_grads:
        return elbo, None
    G = {"lik_variance": P["lik_variance"].grad.numpy().copy(), "layers": []}
    for p in P["layers"]:
        g = {k: v.grad.numpy().copy() for k, v in p.items()}
        g["q_sqrt"] = np.tril(g["q_sqrt"])
        G["layers"].append(g)
    return elbo, G


def propagate_vjp(model, X, zs, S, f_bar=None, mean_bar=None, var_bar=None):
    """d/dX of  sum(f_bar*F_L) + sum(mean_bar*Fmean_L) + sum(var_bar*Fvar_L)  by autograd through the layer stack
    with the draws zs held fixed -- what `tape.gradient(objective, x)` computes on the acquisition side
    (Infill_criteria.py:79-85; propagate dgp.py:34-63).  Cotangents are [S,N,D_L] numpy arrays or None.
    Returns (dX [N,D], F_L, Fmean_L, Fvar_L) as numpy."""
    P = params_from_model(model)
    Xt = torch.tensor(np.asarray(X, dtype=np.float64), dtype=DT, requires_grad=True)
    N = Xt.shape[0]
    F = Xt[None].expand(S, -1, -1).reshape(S * N, -1)
    for layer, p, z in zip(model.layers, P["layers"], zs):
        mean, var = conditional_ND(layer, p, F)
        F = mean + torch.as_tensor(z, dtype=DT).reshape(S * N, -1) * (var + JITTER) ** 0.5
    obj = 0.0
    for bar, out in ((f_bar, F), (mean_bar, mean), (var_bar, var)):
        if bar is not None:
            obj = obj + (torch.as_tensor(np.asarray(bar), dtype=DT).reshape(S * N, -1) * out).sum()
    obj.backward()
    shp = (S, N, -1)
    return (Xt.grad.numpy().copy(), F.detach().numpy().reshape(shp), mean.detach().numpy().reshape(shp),
            var.detach().numpy().reshape(shp))


# --------------------------------------------------------------------------------------
# gpflow.optimizers.NaturalGradient (XiNat) by its own autodiff route  [ext]
# --------------------------------------------------------------------------------------
def natgrad_step_autodiff(q_mu, q_sqrt, g_mu, g_sqrt, gamma):
    """q_mu [M,D], q_sqrt [D,M,M], g_* = d loss / d(q_mu, q_sqrt).  Returns new (q_mu, q_sqrt).

    Follows gpflow/optimizers/natgrad.py: dL/d eta is obtained by back-propagating
    (dL/dq_mu, dL/dq_sqrt) through expectation_to_meanvarsqrt; theta' = theta - gamma dL/d eta;
    natural_to_meanvarsqrt uses chol(-2 nat2), its triangular inverse, and chol(S)."""
    mu = torch.as_tensor(q_mu, dtype=DT)
    L = torch.tril(torch.as_tensor(q_sqrt, dtype=DT))
    gm = torch.as_tensor(g_mu, dtype=DT)
    gs = torch.tril(torch.as_tensor(g_sqrt, dtype=DT))
    M, D = mu.shape
    mu_new = torch.empty_like(mu)
    L_new = torch.empty_like(L)
    I = torch.eye(M, dtype=DT)
    for d in range(D):
        Sigma = L[d] @ L[d].T
        eta1 = mu[:, d].clone().requires_grad_(True)
        eta2 = (Sigma + torch.outer(mu[:, d], mu[:, d])).clone().requires_grad_(True)
        var = eta2 - torch.outer(eta1, eta1)
        sq = torch.linalg.cholesky(var)
        d1, d2 = torch.autograd.grad([eta1 * 1.0, sq], [eta1, eta2], [gm[:, d], gs[d]])
        d2 = 0.5 * (d2 + d2.T)
        Li = torch.linalg.solve_triangular(L[d], I, upper=False)
        Sinv = Li.T @ Li
        nat1 = Sinv @ mu[:, d] - gamma * d1
        nat2 = -0.5 * Sinv - gamma * d2
        R = torch.linalg.cholesky(-2.0 * nat2)
        Ri = torch.linalg.solve_triangular(R, I, upper=False)
        Sn = Ri.T @ Ri
        mu_new[:, d] = Sn @ nat1
        L_new[d] = torch.linalg.cholesky(Sn)
    return mu_new.numpy(), L_new.numpy()


# --------------------------------------------------------------------------------------
# cpu_baseline leg: one optimize_adam iteration (dgp.py:270-276) of the dense formulation
# --------------------------------------------------------------------------------------
def adam_iteration_seconds(model, S, n_points, chunk=4096, threads=None, repeats=1, seed=0):
    """Time ELBO forward + autograd backward on the first ``n_points`` data points."""
    import time
    if threads:
        torch.set_num_threads(threads)
    X, Y = model.data
    X, Y = X[:n_points], Y[:n_points]
    rng = np.random.default_rng(seed)
    zs = [rng.standard_normal((S, n_points, l.num_outputs)) for l in model.layers]
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        elbo_and_grads(model, zs, S=S, data=(X, Y), chunk=chunk)
        best = min(best, time.perf_counter() - t0)
    return best
