"""torch-CPU fp64 restatement of the multi-fidelity DGP with input-space mapping (MF-DGP-EM).

TEST INFRASTRUCTURE ONLY (see the header of dgp_oracle.py): nothing under dgp-toolbox_amd/ may import this.

Follows, as text, R/dgp_dace/models/MF_DGP_EM.py:
  sample / sample_Z_right            :33-58     (mean over 50 samples at the same inputs = one evaluation with the
                                                  MEAN of the 50 normals, which is what is injected here as `zbar`)
  init_layers_mf / make_mf_dgp       :60-86,324-374   (kernels, layer wiring, num_samples=100)
  DGP_Base.propagate / project       :123-203
  E_log_p_Y / _likelihood_at_fidelity:205-260
  ELBO                               :262-301   (incl. the literal `scale = N_{f+1} / N_f` of the projection term)
and R/dgp_dace/utils/layers_red.py:223-300 for the SVGP layer (same arithmetic as utils/layers.py).

PARITY UNPINNED: the reference needs a locally patched GPflow (`InducingPoints(layers=, layers_red=)` with
`Z_left` / `Z_right`, not in the repository) and draws its normals inside `ELBO`; the repository holds no stored
output of this model that could serve as a known answer (the notebook prints quality metrics of a random run).
The patched feature is restated from its call sites: `Z_left` is the trainable [M, D_x] parameter and
`feature.Z = concat([Z_left, Z_right], 1)` is what Kuu / Kuf see (MF_DGP_EM.py:270-271).

All normals are injected (tf.random.normal is not reproducible outside TF).
"""
from __future__ import annotations

import math

import numpy as np
import torch

DT = torch.float64
JITTER = 1e-6


def _t(x, grad=True):
    return torch.tensor(np.asarray(x, dtype=np.float64), dtype=DT, requires_grad=grad)


# ----------------------------------------------------------------------------------------------- kernels
def rbf(X, X2, variance, lengthscales):
    """gpflow SquaredExponential on already-sliced inputs (scalar or per-dimension lengthscales)."""
    Xs = X / lengthscales
    X2s = Xs if X2 is None else X2 / lengthscales
    r2 = -2.0 * Xs @ X2s.T + (Xs * Xs).sum(-1)[:, None] + (X2s * X2s).sum(-1)[None, :]
    return variance * torch.exp(-0.5 * r2)


def kern_K(k, X, X2=None):
    """K(X, X2) of a layer kernel description `k` (dict).  White contributes only when X2 is None (gpflow White)."""
    if k["type"] == "rbf":                       # layers[0] and the input-mapping layers: ARD RBF (+ White)
        K = rbf(X, X2, k["variance"], k["lengthscales"])
    else:                                        # k_corr * (k_prev + Linear) + k_in on [x, f]   (MF_DGP_EM.py:346-352)
        Dx = k["Dx"]
        x, f = X[:, :Dx], X[:, Dx:]
        x2, f2 = (None, None) if X2 is None else (X2[:, :Dx], X2[:, Dx:])
        kc = rbf(x, x2, k["corr_variance"], k["corr_lengthscales"])
        kp = rbf(f, f2, k["prev_variance"], k["prev_lengthscales"])
        kl = k["lin_variance"] * (f @ (f if f2 is None else f2).T) if k.get("lin_variance") is not None else 0.0
        K = kc * (kp + kl) + rbf(x, x2, k["in_variance"], k["in_lengthscales"])
    if X2 is None and k.get("white_variance") is not None:
        K = K + k["white_variance"] * torch.eye(X.shape[0], dtype=DT)
    return K


def kern_Kdiag(k, X):
    if k["type"] == "rbf":
        d = k["variance"] * torch.ones(X.shape[0], dtype=DT)
    else:
        f = X[:, k["Dx"]:]
        lin = k["lin_variance"] * (f * f).sum(-1) if k.get("lin_variance") is not None else 0.0
        d = k["corr_variance"] * (k["prev_variance"] + lin) + k["in_variance"]
    if k.get("white_variance") is not None:
        d = d + k["white_variance"]
    return d


# ----------------------------------------------------------------------------------------------- SVGP layer
def conditional(layer, Z, X):
    """layers_red.py:237-278 (non-white, Zero mean function): mean [P, D], var [P, D]."""
    k = layer["kern"]
    M, D = layer["q_mu"].shape
    Ku = kern_K(k, Z) + JITTER * torch.eye(M, dtype=DT)
    Lu = torch.linalg.cholesky(Ku)
    Kuf = kern_K(k, Z, X)
    A = torch.linalg.solve_triangular(Lu, Kuf, upper=False)
    A = torch.linalg.solve_triangular(Lu.T, A, upper=True)
    mean = A.T @ layer["q_mu"]
    q_sqrt = torch.tril(layer["q_sqrt"])
    SK = q_sqrt @ q_sqrt.transpose(1, 2) - Ku[None]
    B = SK @ A[None].expand(D, -1, -1)
    var = kern_Kdiag(k, X)[:, None] + (A[None] * B).sum(1).T
    return mean, var


def layer_KL(layer, Z):
    """layers_red.py:280-308 (non-white)."""
    M, D = layer["q_mu"].shape
    q_sqrt = torch.tril(layer["q_sqrt"])
    Ku = kern_K(layer["kern"], Z) + JITTER * torch.eye(M, dtype=DT)
    Lu = torch.linalg.cholesky(Ku)
    KL = -0.5 * D * M - 0.5 * torch.log(torch.diagonal(q_sqrt, dim1=1, dim2=2) ** 2).sum()
    KL = KL + torch.log(torch.diagonal(Lu)).sum() * D
    LiS = torch.linalg.solve_triangular(Lu[None].expand(D, -1, -1), q_sqrt, upper=False)
    KL = KL + 0.5 * (LiS ** 2).sum()
    KL = KL + 0.5 * (layer["q_mu"] * torch.cholesky_solve(layer["q_mu"], Lu)).sum()
    return KL


def sample_layer(layer, Z, X, z):
    """sample_from_conditional + reparameterize (utils.py:40-41): X [P, D_in], z [P, D_out]."""
    mean, var = conditional(layer, Z, X)
    return mean + z * (var + JITTER) ** 0.5, mean, var


# ----------------------------------------------------------------------------------------------- model
def make_params(X, Z, W, add_linear=True, lik_variance=1.0, proj_variance=1.0):
    """make_mf_dgp + init_layers_mf (MF_DGP_EM.py:60-86,324-374): parameter dict of torch leaves.

    q_sqrt starts at the identity here; the reference initialises it to chol(Kuu) with the sampled Z_right
    (layers_red.py:213-216) and overwrites q_mu / rescales q_sqrt at the start of training anyway
    (MF_DGP_EM.py:434-448) - tests assign explicit values.
    """
    n = len(X)
    P = {"layers": [], "layers_red": [], "lik_variance": _t(lik_variance), "proj_variance": _t(proj_variance)}
    for l in range(n):
        Dx = X[l].shape[1]
        M = Z[l].shape[0]
        if l == 0:
            k = {"type": "rbf", "variance": _t(1.0), "lengthscales": _t(np.ones(Dx))}
        else:
            k = {"type": "mf", "Dx": Dx, "corr_variance": _t(1.0), "corr_lengthscales": _t(1.0),
                 "prev_variance": _t(1.0), "prev_lengthscales": _t(1.0),
                 "lin_variance": _t(1.0) if add_linear else None, "in_variance": _t(1.0), "in_lengthscales": _t(1.0)}
        if l < n - 1:
            k["white_variance"] = _t(1e-6)                       # MF_DGP_EM.py:364-367
        P["layers"].append({"kern": k, "Z": _t(Z[l]), "q_mu": _t(np.zeros((M, 1))), "q_sqrt": _t(np.eye(M)[None])})
    for i in range(1, n):                                        # init_layers_mf: layers_red[i-1]
        Din, Dout = X[-i].shape[1], X[-(1 + i)].shape[1]
        Mw = W[i - 1].shape[0]
        k = {"type": "rbf", "variance": _t(1.0), "lengthscales": _t(np.ones(Din))}
        P["layers_red"].append({"kern": k, "Z": _t(W[i - 1]), "q_mu": _t(np.zeros((Mw, Dout))),
                                "q_sqrt": _t(np.tile(np.eye(Mw)[None], (Dout, 1, 1)))})
    return P


def leaves(P):
    """name -> leaf tensor (every parameter the reference could train)."""
    out = {"lik_variance": P["lik_variance"], "proj_variance": P["proj_variance"]}
    for group in ("layers", "layers_red"):
        for i, l in enumerate(P[group]):
            for name in ("Z", "q_mu", "q_sqrt"):
                out[f"{group}.{i}.{name}"] = l[name]
            for name, v in l["kern"].items():
                if isinstance(v, torch.Tensor):
                    out[f"{group}.{i}.kern.{name}"] = v
    return out


def z_right(P, i, zbar):
    """sample_Z_right(layers[0:i], layers_red[L-i:], Z_left_i)  (MF_DGP_EM.py:38-58).

    zbar: {"red": [per layers_red used: [M_i, D_out]], "layers": [per layers[0:i]: [M_i, 1]]} = means of 50 normals.
    Returns Z_right [M_i, 1]; uses the CURRENT full Z of the earlier augmented layers (P["_Zfull"]).
    """
    L = len(P["layers_red"])
    H = P["layers"][i]["Z"]
    Hs = [H]
    for j, lr in enumerate(P["layers_red"][L - i:]):
        H = sample_layer(lr, lr["Z"], H, zbar["red"][j])[0]
        Hs.append(H)
    Zr = None
    for j in range(i):
        lay = P["layers"][j]
        inp = Hs[-1] if j == 0 else torch.cat([Hs[-(j + 1)], Zr], 1)
        Zr = sample_layer(lay, P["_Zfull"][j], inp, zbar["layers"][j])[0]
    return Zr


def propagate(P, X, S, zs, ws, fidelity_dim, project=False):
    """DGP_Base.propagate (MF_DGP_EM.py:123-168) on flattened [S*N, D] arrays."""
    N = X.shape[0]
    L = len(P["layers_red"])
    H = X[None].expand(S, -1, -1).reshape(S * N, -1)
    Hs, Hmeans, Hvars = [H], [], []
    for j, lr in enumerate(P["layers_red"][L - fidelity_dim:]):
        H, m, v = sample_layer(lr, lr["Z"], H, ws[j].reshape(S * N, -1))
        Hs.append(H); Hmeans.append(m); Hvars.append(v)
    if project:
        return Hs, Hmeans, Hvars
    Fs, Fmeans, Fvars = [], [], []
    F = None
    for i in range(fidelity_dim + 1):
        inp = Hs[-1] if i == 0 else torch.cat([Hs[-(i + 1)], F], 1)
        F, m, v = sample_layer(P["layers"][i], P["_Zfull"][i], inp, zs[i].reshape(S * N, -1))
        Fs.append(F); Fmeans.append(m); Fvars.append(v)
    return Fs, Fmeans, Fvars


def conditional_full(layer, Z, X):
    """layers_red.py:237-272 with full_cov=True (non-white, Zero mean function): mean [N, D], var [N, N, D]."""
    k = layer["kern"]
    M, D = layer["q_mu"].shape
    Ku = kern_K(k, Z) + JITTER * torch.eye(M, dtype=DT)
    Lu = torch.linalg.cholesky(Ku)
    Kuf = kern_K(k, Z, X)
    A = torch.linalg.solve_triangular(Lu, Kuf, upper=False)
    A = torch.linalg.solve_triangular(Lu.T, A, upper=True)
    mean = A.T @ layer["q_mu"]
    q_sqrt = torch.tril(layer["q_sqrt"])
    SK = q_sqrt @ q_sqrt.transpose(1, 2) - Ku[None]
    At = A[None].expand(D, -1, -1)
    delta = At.transpose(1, 2) @ (SK @ At)                    # [D, N, N]
    var = kern_K(k, X)[None] + delta                          # kern.K(X): a White term sits on the diagonal
    return mean, var.permute(1, 2, 0)


def sample_layer_full(layer, Z, X, z):
    """sample_from_conditional(full_cov=True) + reparameterize (utils.py:43-51) for ONE sample: X [N, D_in], z [N, D]."""
    mean, var = conditional_full(layer, Z, X)
    N = X.shape[0]
    chol = torch.linalg.cholesky(var.permute(2, 0, 1) + JITTER * torch.eye(N, dtype=DT)[None])     # [D, N, N]
    f = mean + (chol @ z.T[:, :, None])[:, :, 0].T
    return f, mean, var


def propagate_full_cov(P, X, S, zs, ws, fidelity_dim, project=False):
    """DGP_Base.propagate(full_cov=True) (MF_DGP_EM.py:123-168): conditional_SND maps the layer over the samples
    (layers_red.py:63-80); returns per layer [S, N, D] samples / means and [S, N, N, D] covariances."""
    L = len(P["layers_red"])
    out = None
    for s in range(S):
        H = X
        Hs, Hm, Hv = [H], [], []
        for j, lr in enumerate(P["layers_red"][L - fidelity_dim:]):
            H, m, v = sample_layer_full(lr, lr["Z"], H, ws[j][s])
            Hs.append(H); Hm.append(m); Hv.append(v)
        if project:
            res = (Hs, Hm, Hv)
        else:
            Fs, Fm, Fv = [], [], []
            F = None
            for i in range(fidelity_dim + 1):
                inp = Hs[-1] if i == 0 else torch.cat([Hs[-(i + 1)], F], 1)
                F, m, v = sample_layer_full(P["layers"][i], P["_Zfull"][i], inp, zs[i][s])
                Fs.append(F); Fm.append(m); Fv.append(v)
            res = (Fs, Fm, Fv)
        if out is None:
            out = tuple([[t] for t in part] for part in res)
        else:
            for part, acc in zip(res, out):
                for t, a in zip(part, acc):
                    a.append(t)
    return tuple([torch.stack(a) for a in part] for part in out)


def _gauss_ve(mean, var, Y, s2, S):
    Yt = Y[None].expand(S, -1, -1).reshape(mean.shape)
    return -0.5 * math.log(2 * math.pi) - 0.5 * torch.log(s2) - 0.5 * ((Yt - mean) ** 2 + var) / s2


def elbo(P, X, Y, X_red, normals, S):
    """DGP_Base.ELBO (MF_DGP_EM.py:262-301).  X, Y, X_red: lists of torch arrays; normals:
        {"zright": [None, zbar_1, ...], "zs": [per fidelity f: list over layers[:f+1] of [S,N_f,1]],
         "ws": [per fidelity f: list over layers_red[L-f:]], "ws_proj": [per f < n-1: list over layers_red[L-(f+1):]]}
    Returns (ELBO, {"L":, "L_red":, "KL":, "KL_red":})."""
    n = len(P["layers"])
    P["_Zfull"] = [P["layers"][0]["Z"]]
    for i in range(1, n):
        P["_Zfull"].append(torch.cat([P["layers"][i]["Z"], z_right(P, i, normals["zright"][i])], 1))
    Lt = KL = L_red = KL_red = 0.0
    for f in range(n):
        Xl, Yl = X[f], Y[f]
        _, Fmeans, Fvars = propagate(P, Xl, S, normals["zs"][f], normals["ws"][f], f)
        s2 = P["lik_variance"] if f == n - 1 else P["layers"][f]["kern"]["white_variance"]
        Lt = Lt + _gauss_ve(Fmeans[f], Fvars[f], Yl, s2, S).sum() / S
        KL = KL + layer_KL(P["layers"][f], P["_Zfull"][f])
        if f < n - 1:
            Xn, Yn = X[f + 1], X_red[f]
            _, Hmeans, Hvars = propagate(P, Xn, S, None, normals["ws_proj"][f], f + 1, project=True)
            scale = Xn.shape[0] / Xl.shape[0]                    # MF_DGP_EM.py:292-294, literally
            L_red = L_red + _gauss_ve(Hmeans[f], Hvars[f], Yn, P["proj_variance"], S).sum() / S * scale
            lr = P["layers_red"][f]
            KL_red = KL_red + layer_KL(lr, lr["Z"])
    return Lt + L_red - KL - KL_red, {"L": Lt, "L_red": L_red, "KL": KL, "KL_red": KL_red}


def elbo_and_grads(P, X, Y, X_red, normals, S):
    """ELBO and d ELBO / d every leaf (constrained parameters), as numpy."""
    lv = leaves(P)
    for v in lv.values():
        v.grad = None
    Xt = [torch.as_tensor(np.asarray(x), dtype=DT) for x in X]
    Yt = [torch.as_tensor(np.asarray(y), dtype=DT) for y in Y]
    Xr = [torch.as_tensor(np.asarray(x), dtype=DT) for x in X_red]
    nt = _normals_to_torch(normals)
    val, parts = elbo(P, Xt, Yt, Xr, nt, S)
    val.backward()
    grads = {k: (np.zeros(tuple(v.shape)) if v.grad is None else v.grad.detach().numpy().copy()) for k, v in lv.items()}
    if "layers.0.q_sqrt" in grads:
        for k in grads:
            if k.endswith("q_sqrt"):
                grads[k] = np.tril(grads[k])
    return float(val.detach()), {k: float(v.detach()) for k, v in parts.items()}, grads


def _normals_to_torch(normals):
    def conv(o):
        if o is None:
            return None
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [conv(v) for v in o]
        return torch.as_tensor(np.asarray(o), dtype=DT)
    return conv(normals)


def draw_normals(rng, X, P, S, n_zright=50):
    """All the N(0,1) draws one ELBO evaluation consumes, in the layout `elbo` expects."""
    n = len(X)
    L = n - 1
    out = {"zright": [None], "zs": [], "ws": [], "ws_proj": []}
    for i in range(1, n):
        Mi = P["layers"][i]["Z"].shape[0]
        red = [rng.standard_normal((n_zright, Mi, lr["q_mu"].shape[1])).mean(0) for lr in P["layers_red"][L - i:]]
        lay = [rng.standard_normal((n_zright, Mi, 1)).mean(0) for _ in range(i)]
        out["zright"].append({"red": red, "layers": lay})
    for f in range(n):
        N = X[f].shape[0]
        out["ws"].append([rng.standard_normal((S, N, lr["q_mu"].shape[1])) for lr in P["layers_red"][L - f:]])
        out["zs"].append([rng.standard_normal((S, N, 1)) for _ in range(f + 1)])
        if f < n - 1:
            Nn = X[f + 1].shape[0]
            out["ws_proj"].append([rng.standard_normal((S, Nn, lr["q_mu"].shape[1])) for lr in P["layers_red"][L - (f + 1):]])
    return out
