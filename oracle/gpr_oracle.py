"""CPU restatement of gpflow.models.GPR as the reference uses it for ``num_layers == 0``
(R/dgp_dace/BO/SO_BO.py:187-200 construction, :252-256 training, Infill_criteria.py:28-35 prediction).

TEST INFRASTRUCTURE ONLY (see the header of dgp_oracle.py).  GPflow is not installed here, so its published
definitions are restated  [ext]:
  log_marginal_likelihood = sum over output columns of log N(y | 0, K(X,X) + s2 I)      (GPR.log_marginal_likelihood)
  predict_f: base_conditional(K(X,X*), K(X,X) + s2 I, K_diag(X*), Y):  mean = A2^T Y, var = kdiag - sum A^2
             with A = L^-1 K(X,X*), A2 = L^-T A;  predict_y adds s2 to the variance         (GPR.predict_f / predict_y)
Pinned against scikit-learn's GaussianProcessRegressor in tests/test_oracle.py.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla


def log_marginal_likelihood(kern, X, Y, noise):
    N, Dy = Y.shape
    L = np.linalg.cholesky(kern.K(X) + noise * np.eye(N))
    A = sla.solve_triangular(L, Y, lower=True)
    return float(-0.5 * np.sum(A * A) - Dy * np.sum(np.log(np.diag(L))) - 0.5 * N * Dy * math.log(2.0 * math.pi))


def predict_y(kern, X, Y, noise, Xnew):
    N = X.shape[0]
    L = np.linalg.cholesky(kern.K(X) + noise * np.eye(N))
    A = sla.solve_triangular(L, kern.K(X, Xnew), lower=True)                 # [N, N*]
    mean = A.T @ sla.solve_triangular(L, Y, lower=True)
    var = kern.K_diag(Xnew) - np.sum(A * A, 0)
    return mean, np.tile(var[:, None], [1, Y.shape[1]]) + noise


def lml_and_grads(kern, X, Y, noise):
    """(lml, d/d variance, d/d lengthscales, d/d noise) by torch autograd on the same expression."""
    import torch
    import dgp_oracle_torch as OT
    v = torch.tensor(float(kern.variance), dtype=OT.DT, requires_grad=True)
    ls = torch.tensor(np.asarray(kern.lengthscales, dtype=np.float64), dtype=OT.DT, requires_grad=True)
    s2 = torch.tensor(float(noise), dtype=OT.DT, requires_grad=True)
    Xt, Yt = torch.as_tensor(X, dtype=OT.DT), torch.as_tensor(Y, dtype=OT.DT)
    N, Dy = Yt.shape
    K = OT.rbf_K(v, ls, Xt, kind=kern.kind) + s2 * torch.eye(N, dtype=OT.DT)
    L = torch.linalg.cholesky(K)
    A = torch.linalg.solve_triangular(L, Yt, upper=False)
    lml = -0.5 * (A * A).sum() - Dy * torch.log(torch.diagonal(L)).sum() - 0.5 * N * Dy * math.log(2.0 * math.pi)
    lml.backward()
    return float(lml.detach()), float(v.grad), ls.grad.numpy().copy(), float(s2.grad)
