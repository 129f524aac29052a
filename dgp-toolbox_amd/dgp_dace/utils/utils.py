"""Likelihood wrapper of the DGP (reference: dgp_dace/utils/utils.py:54-117).

Only the Gaussian likelihood is on the accelerated path; for it the reference's wrapper does no
tiling (utils.py:64-73).  The variational expectations and predictive moments of the ELBO/predict
paths are evaluated inside the HIP kernels (csrc/points.hip gauss_lik_kernel, lik_predict_var);
``reparameterize`` (utils.py:22-51, diagonal branch) is fused into var_mean_sample_kernel.
"""
from __future__ import annotations


class BroadcastingLikelihood:
    def __init__(self, likelihood):
        self.likelihood = likelihood
        self.needs_broadcasting = False
