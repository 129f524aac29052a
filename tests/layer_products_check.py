"""Element-by-element check of one SVGP layer's point contractions (C-ABI dgp_dev_layer_products: the launches of
forward_chunk / backward_chunk on explicit operands) against NumPy.

Reference arithmetic being checked (whitened form of R/dgp_dace/utils/layers.py:243-263 and its adjoint):
    Ct = Kt Linv^T, |c|^2, T_d = Ct W_d, |t_d|^2, mean0 = Ct u,
    Cbar = sum_d 2 vbar_d (T_d W_d^T - Ct) + mbar u^T,  g = (Cbar Linv) .* Kt,  du = Ct^T mbar,
    G_d = tril(Ct^T diag(vbar_d) Ct)  (du rides on G_d's launch where the Gram kernel runs it: gemm_gram.h, form DU).

Used in-process by tests/test_gpu_units.py and as a child process (the kernel switches DGP_TALL / DGP_TALLU / DGP_WIDE are
read once per process):  python tests/layer_products_check.py P:Mp:D:engCt,engT,engCbar [...]
"""
import os
import sys

import numpy as np

TOL = 1e-12          # of the largest reference magnitude of the array, for EVERY element


def operands(P, Mp, D, seed):
    rng = np.random.default_rng(seed)
    Kt = rng.standard_normal((P, Mp)) / 16.0
    Linv = np.tril(rng.standard_normal((Mp, Mp))) / 4.0
    Wcat = np.concatenate([np.tril(rng.standard_normal((Mp, Mp))) / 8.0 for _ in range(D)], axis=1)
    u = rng.standard_normal((Mp, D))
    vbar = rng.standard_normal((P, D))          # signed weights
    mbar = rng.standard_normal((P, D))
    return Kt, Linv, Wcat, u, vbar, mbar


def reference(Kt, Linv, Wcat, u, vbar, mbar):
    P, Mp = Kt.shape
    D = u.shape[1]
    Ct = Kt @ Linv.T
    T = Ct @ Wcat
    Cbar = mbar @ u.T - 2.0 * vbar.sum(axis=1, keepdims=True) * Ct
    for d in range(D):
        Td = T[:, d * Mp:(d + 1) * Mp]
        Cbar += 2.0 * vbar[:, d:d + 1] * (Td @ Wcat[:, d * Mp:(d + 1) * Mp].T)
    return {"Ct": Ct, "cn": (Ct * Ct).sum(axis=1), "T": T,
            "tn": (T.reshape(P, D, Mp) ** 2).sum(axis=2), "mean0": Ct @ u, "Cbar": Cbar,
            "g": (Cbar @ Linv) * Kt, "du": Ct.T @ mbar,
            "Gd": np.stack([np.tril(Ct.T @ (vbar[:, d:d + 1] * Ct)) for d in range(D)])}


def check(ctx, P, Mp, D, expect=None, seed=None):
    """Runs the layer's products at [P, Mp] with D outputs; compares every element of every output with NumPy; asserts the
    kernel families that ran - (Ct, T, Cbar) or (Ct, T, Cbar, du, Gd) - when `expect` is given.  Returns the engines."""
    ops = operands(P, Mp, D, P + 31 * D + Mp if seed is None else seed)
    got = ctx.dev_layer_products(*ops)
    if expect is not None:
        ran = got["engines"][:3] + (got["engines"][4:6] if len(expect) == 5 else [])
        assert ran == list(expect), (P, Mp, D, got["engines"])
        # mean0 comes out of the Ct launch (as Kt (LinvT u)) exactly where that one is the wide-tile kernel at Mp = 256, D <= 8
        assert got["mean_in_ct"] == (expect[0] == "wide" and Mp == 256 and D <= 8 and os.environ.get("DGP_WIDE_MEAN", "1") != "0"), got["mean_in_ct"]
    ref = reference(*ops)
    for k, r in ref.items():
        scale = np.abs(r).max()
        err = np.abs((np.tril(got[k]) if k == "Gd" else got[k]) - r)
        i = np.unravel_index(np.argmax(err), err.shape)
        assert err.max() <= TOL * scale, (k, P, Mp, D, got["engines"], "worst element", i, float(err.max()), float(scale))
    return got["engines"]


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "dgp-toolbox_amd"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    from dgp_dace import _native
    c = _native.Context(0)
    for spec in sys.argv[1:]:
        P, Mp, D, eng = spec.split(":")
        e = check(c, int(P), int(Mp), int(D), eng.split(",") if eng else None)
        print("ok", spec, e, flush=True)
    c.close()
