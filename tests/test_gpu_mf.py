"""GPU parity of the multi-fidelity pieces against oracle/mf_dgp_em_oracle.py: the MF layer kernel and the
White variance inside one SVGP layer (forward, x-gradient, parameter gradients through dgp_vjp_accumulate)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


def _layer_case(kind, white, D_out, rng, M=21, P=37, Dx=3):
    import torch
    import mf_dgp_em_oracle as mo
    Din = Dx + 1 if kind == "mf" else Dx
    Z = rng.uniform(-1, 1, (M, Din))
    X = rng.uniform(-1, 1, (P, Din))
    if kind == "mf":
        hyp = rng.uniform(0.5, 1.5, 7)
        k = {"type": "mf", "Dx": Dx, "corr_variance": mo._t(hyp[0]), "corr_lengthscales": mo._t(hyp[1]),
             "prev_variance": mo._t(hyp[2]), "prev_lengthscales": mo._t(hyp[3]), "lin_variance": mo._t(hyp[4]),
             "in_variance": mo._t(hyp[5]), "in_lengthscales": mo._t(hyp[6])}
        kp = list(hyp)
        names = ["corr_variance", "corr_lengthscales", "prev_variance", "prev_lengthscales", "lin_variance", "in_variance",
                 "in_lengthscales"]
    else:
        var, ls = rng.uniform(0.5, 1.5), rng.uniform(0.6, 1.4, Din)
        k = {"type": "rbf", "variance": mo._t(var), "lengthscales": mo._t(ls)}
        kp = [var] + list(ls)
        names = ["variance", "lengthscales"]
    if white:
        wv = 0.07
        k["white_variance"] = mo._t(wv)
        kp.append(wv)
        names.append("white_variance")
    q_mu = rng.standard_normal((M, D_out))
    q_sqrt = np.tril(rng.standard_normal((D_out, M, M)) * 0.1) + np.eye(M)[None] * 0.6
    layer = {"kern": k, "Z": mo._t(Z), "q_mu": mo._t(q_mu), "q_sqrt": mo._t(q_sqrt)}
    flat = np.concatenate([Z.ravel(), np.array(kp), q_mu.ravel(), q_sqrt.ravel(), [1.0]])
    desc = (Din, D_out, M, 0, 3 if kind == "mf" else 0, 0, 1 if white else 0)
    return layer, names, flat, desc, X


@pytest.mark.parametrize("kind,white,D_out", [("mf", False, 1), ("mf", True, 2), ("rbf", True, 2)])
def test_single_layer_with_mf_kernel_and_white_variance(kind, white, D_out):
    import torch
    import mf_dgp_em_oracle as mo
    from dgp_dace import _native
    rng = np.random.default_rng(5)
    layer, names, flat, desc, X = _layer_case(kind, white, D_out, rng)
    P, Din = X.shape
    M = desc[2]
    z = rng.standard_normal((1, P, D_out))
    fb, mb, vb = (rng.standard_normal((1, P, D_out)) for _ in range(3))
    # oracle
    Xt = torch.tensor(X, dtype=mo.DT, requires_grad=True)
    F, mean, var = mo.sample_layer(layer, layer["Z"], Xt, torch.as_tensor(z[0]))
    obj = (torch.as_tensor(fb[0]) * F + torch.as_tensor(mb[0]) * mean + torch.as_tensor(vb[0]) * var).sum()
    kl = mo.layer_KL(layer, layer["Z"])
    (obj - kl).backward()
    # device
    ctx = _native.Context(0)
    ctx.model_set([desc], flat, None)
    Fs, Fm, Fv = ctx.propagate(X, 1, 0, [z])
    np.testing.assert_allclose(Fm[0][0], mean.detach().numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(Fv[0][0], var.detach().numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Fs[0][0], F.detach().numpy(), rtol=1e-8, atol=1e-10)
    xbar = ctx.propagate_vjp(X, 1, 0, [z], f_bar=fb, mean_bar=mb, var_bar=vb, accumulate="reset")
    np.testing.assert_allclose(xbar, Xt.grad.numpy(), rtol=1e-7, atol=1e-8)
    minus_kl = ctx.grad_finish(want_elbo=True)
    assert abs(minus_kl + float(kl.detach())) <= 1e-9 * max(1.0, abs(float(kl.detach())))
    g = ctx.grad_get()
    off = 0
    np.testing.assert_allclose(g[off:off + M * Din].reshape(M, Din), layer["Z"].grad.numpy(), rtol=1e-7, atol=1e-7)
    off += M * Din
    for nm in names:
        ref = np.atleast_1d(layer["kern"][nm].grad.numpy())
        np.testing.assert_allclose(g[off:off + ref.size], ref, rtol=1e-7, atol=1e-7, err_msg=nm)
        off += ref.size
    np.testing.assert_allclose(g[off:off + M * D_out].reshape(M, D_out), layer["q_mu"].grad.numpy(), rtol=1e-7, atol=1e-8)
    off += M * D_out
    np.testing.assert_allclose(g[off:off + D_out * M * M].reshape(D_out, M, M), np.tril(layer["q_sqrt"].grad.numpy()),
                               rtol=1e-7, atol=1e-8)
    # a second accumulate call adds the data side once more; the KL side stays single
    ctx.propagate_vjp(X, 1, 0, [z], f_bar=fb, mean_bar=mb, var_bar=vb, accumulate="reset")
    ctx.propagate_vjp(X, 1, 0, [z], f_bar=fb, mean_bar=mb, var_bar=vb, accumulate="add")
    ctx.grad_finish()
    g2 = ctx.grad_get()
    ctx.propagate_vjp(X, 1, 0, [z], f_bar=0 * fb, mean_bar=0 * mb, var_bar=0 * vb, accumulate="reset")
    ctx.grad_finish()
    g0 = ctx.grad_get()          # - d KL only
    np.testing.assert_allclose(g2 - g0, 2.0 * (g - g0), rtol=1e-9, atol=1e-9)
