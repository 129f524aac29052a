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
