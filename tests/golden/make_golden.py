"""Generates the committed golden fixtures (tests/golden/*.npz) from the CPU oracle.

Run in the build container:  python tests/golden/make_golden.py
The reference itself (TensorFlow/GPflow) cannot be imported here (not installed, no network), so
these vectors come from ``oracle/`` — the restatement pinned by the notebook's known answers
(SURVEY.md §8c).  The fixtures are data only: inputs, injected normals ``zs`` and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))

import dgp_oracle as O                      # noqa: E402
import dgp_oracle_torch as T                # noqa: E402
from dgp_oracle_train import OracleTrainer  # noqa: E402


def notebook_case():
    """nb_DGP_regression.ipynb cells 2, 6, 10, 15-18."""
    np.random.seed(0)
    f_step = lambda x: 0. if x < 0.5 else 1.
    N, M = 50, 25
    X = np.random.uniform(0, 1, N)[:, None]
    Z = np.random.uniform(0, 1, M)[:, None]
    Y = np.reshape([f_step(x) for x in X], X.shape) + np.random.randn(*X.shape) * 1e-2
    return X, Y, Z


def model_state(model, prefix=""):
    d = {prefix + "lik_variance": np.float64(model.lik_variance)}
    for i, l in enumerate(model.layers):
        d[f"{prefix}L{i}_Z"] = l.Z
        d[f"{prefix}L{i}_variance"] = np.float64(l.kern.variance)
        d[f"{prefix}L{i}_lengthscales"] = l.kern.lengthscales
        d[f"{prefix}L{i}_q_mu"] = l.q_mu
        d[f"{prefix}L{i}_q_sqrt"] = l.q_sqrt
    return d


def make_case(name, D, num_units, Dy, white, seed, N=64, M=16, S=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1.5, 1.5, (N, D))
    Y = rng.standard_normal((N, Dy))
    Z = rng.uniform(-1.5, 1.5, (M, D))
    dims = [D] + list(num_units)
    kernels = [O.RBF(0.7 + 0.2 * i, 0.8 + 0.15 * rng.uniform(size=d)) for i, d in enumerate(dims)]
    model = O.OracleDGP(X, Y, Z, kernels, num_units, lik_variance=0.4, white=white, num_samples=S)
    for l in model.layers:
        l.Z = l.Z + 0.05 * rng.standard_normal(l.Z.shape)
        mu_w = 0.4 * rng.standard_normal(l.q_mu.shape)
        sq_w = np.tril(0.15 * rng.standard_normal(l.q_sqrt.shape)) + 0.6 * np.eye(M)[None]
        if white:
            l.q_mu, l.q_sqrt = mu_w, sq_w
        else:       # a well-conditioned non-trivial q(u): the whitened draw mapped through chol(Kuu)
            l.build_cholesky()
            l.q_mu, l.q_sqrt = l.Lu @ mu_w, np.tril(l.Lu[None] @ sq_w)
    zs = [rng.standard_normal((S, N, l.num_outputs)) for l in model.layers]

    out = {"X": X, "Y": Y, "Z_init": Z, "num_units": np.array(num_units), "white": np.array(white),
           "S": np.array(S), "seed": np.array(seed)}
    out.update(model_state(model))
    for i, l in enumerate(model.layers):
        out[f"L{i}_mean_kind"] = np.array(l.mean_function.kind)
        if l.mean_function.kind == "linear":
            out[f"L{i}_mean_A"] = l.mean_function.A
        out[f"zs{i}"] = zs[i]

    Fs, Fm, Fv = model.propagate(X, S, zs)
    for i in range(len(model.layers)):
        out[f"Fs{i}"], out[f"Fmeans{i}"], out[f"Fvars{i}"] = Fs[i], Fm[i], Fv[i]
    L, KLs = model.elbo_terms(zs)
    out["data_term"], out["KLs"], out["elbo"] = L, np.array(KLs), L - np.sum(KLs)

    elbo_t, G = T.elbo_and_grads(model, zs)
    assert abs(elbo_t - out["elbo"]) < 1e-9 * max(1, abs(elbo_t))
    out["g_lik_variance"] = G["lik_variance"]
    for i, g in enumerate(G["layers"]):
        for k, v in g.items():
            out[f"g_L{i}_{k}"] = v

    # prediction at new inputs (dgp.py:113-124, 362-366)
    Xn = rng.uniform(-1.5, 1.5, (11, D))
    Sn = 4
    zn = [rng.standard_normal((Sn, 11, l.num_outputs)) for l in model.layers]
    out["Xnew"], out["Snew"] = Xn, np.array(Sn)
    for i, z in enumerate(zn):
        out[f"znew{i}"] = z
    out["predict_y_mean"], out["predict_y_var"] = model.predict_y(Xn, Sn, zn)
    out["predict_mean"], out["predict_var"] = model.predict(Xn, Sn, zn)

    # one natural-gradient step on every layer from this state (gamma = 0.01)
    gamma = 0.01
    out["natgrad_gamma"] = np.array(gamma)
    for i, l in enumerate(model.layers):
        mu_n, sq_n = O.natgrad_step(l.q_mu, l.q_sqrt, -G["layers"][i]["q_mu"], -G["layers"][i]["q_sqrt"], gamma)
        mu_a, sq_a = T.natgrad_step_autodiff(l.q_mu, l.q_sqrt, -G["layers"][i]["q_mu"],
                                             -G["layers"][i]["q_sqrt"], gamma)
        assert np.abs(mu_n - mu_a).max() < 1e-10 and np.abs(sq_n - sq_a).max() < 1e-10
        out[f"ng_L{i}_q_mu"], out[f"ng_L{i}_q_sqrt"] = mu_n, sq_n

    # two Adam iterations with Philox normals (base seed 11): trajectory parity incl. the RNG
    tr = OracleTrainer(model, base_seed=11)
    adam = tr.new_adam(lr=0.01, beta_1=0.9, beta_2=0.999, epsilon=1e-7)
    e0 = tr.adam_iteration(adam)
    e1 = tr.adam_iteration(adam)
    out["adam_elbos"] = np.array([e0, e1])
    out.update(model_state(model, prefix="adam2_"))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "elbo", out["elbo"], "adam", e0, e1)


def make_training_case():
    """optimize_nat_adam trajectory on the notebook data (dgp.py:280-345), Philox base seed 5."""
    X, Y, Z = notebook_case()
    out = {"X": X, "Y": Y, "Z": Z}
    for ng_all in (True, False):
        kernels = [O.RBF(1.0, [1.0]) for _ in range(3)]
        model = O.OracleDGP(X, Y, Z, kernels, [1, 1], num_samples=10)
        tr = OracleTrainer(model, base_seed=5)
        elbos = tr.optimize_nat_adam(iterations1=3, iterations2=4, lr_adam=0.01, lr_gamma=0.01, beta_1=0.8,
                                     beta_2=0.9, ng_all=ng_all)
        tag = "ngall" if ng_all else "nglast"
        out[f"{tag}_elbos"] = np.array(elbos)
        out.update(model_state(model, prefix=f"{tag}_"))
        print("nat_adam", tag, elbos)
    np.savez_compressed(os.path.join(HERE, "notebook_nat_adam.npz"), **out)


def make_mf_case():
    """Two-fidelity MF-DGP-EM (MF_DGP_EM.py): inputs, every injected normal, parameter state, the four terms of the bound
    and its gradients from oracle/mf_dgp_em_oracle.py (parity unpinned, see its header: this fixture guards the
    restatement against accidental change and gives the GPU test a committed target)."""
    import torch
    import mf_dgp_em_oracle as mo
    rng = np.random.default_rng(77)
    X = [rng.uniform(0, 1, (14, 2)), rng.uniform(0, 1, (9, 3))]
    Y = [np.sin(3 * X[0].sum(1, keepdims=True)), np.sin(3 * X[1].sum(1, keepdims=True)) + 0.1]
    X_red = [rng.uniform(0, 1, (9, 2))]
    S = 3
    P = mo.make_params(X, [x.copy() for x in X], [X[1].copy()])
    state = {}
    with torch.no_grad():
        for name, leaf in mo.leaves(P).items():
            if name.endswith(".Z"):
                v = leaf.numpy().copy()
            elif name.endswith("q_mu"):
                v = rng.standard_normal(tuple(leaf.shape)) * 0.5
            elif name.endswith("q_sqrt"):
                D, M, _ = leaf.shape
                v = np.tril(rng.standard_normal((D, M, M)) * 0.05) + 0.4 * np.eye(M)[None]
            elif name.endswith("white_variance"):
                v = np.array(0.05)
            elif name == "lik_variance":
                v = np.array(0.3)
            elif name == "proj_variance":
                v = np.array(0.2)
            else:
                v = rng.uniform(0.7, 1.3, tuple(leaf.shape))
            leaf.copy_(torch.as_tensor(v).reshape(leaf.shape))
            state["p." + name] = np.asarray(v, dtype=np.float64)
    nm = mo.draw_normals(rng, X, P, S)
    elbo, parts, grads = mo.elbo_and_grads(P, X, Y, X_red, nm, S)
    out = {"S": np.array(S), "elbo": np.float64(elbo), "X0": X[0], "X1": X[1], "Y0": Y[0], "Y1": Y[1], "X_red0": X_red[0],
           "zright_red0": nm["zright"][1]["red"][0], "zright_lay0": nm["zright"][1]["layers"][0],
           "zs0_0": nm["zs"][0][0], "zs1_0": nm["zs"][1][0], "zs1_1": nm["zs"][1][1], "ws1_0": nm["ws"][1][0],
           "wsproj0_0": nm["ws_proj"][0][0]}
    out.update({"part." + k: np.float64(v) for k, v in parts.items()})
    out.update(state)
    out.update({"g." + k: v for k, v in grads.items()})
    np.savez_compressed(os.path.join(HERE, "mf_dgp_em_two_fidelities.npz"), **out)
    print("mf-dgp-em elbo", elbo, parts)


def main():
    X, Y, Z = notebook_case()
    kernels = [O.RBF(1.0, [1.0] * u) for u in [1, 1, 1]]
    model = O.OracleDGP(X, Y, Z, kernels, [1, 1], num_samples=10)
    zs = O.draw_zs(model, 0, 10, X.shape[0])
    elbo = model.ELBO(zs)
    np.savez_compressed(os.path.join(HERE, "notebook_known_answer.npz"), X=X, Y=Y, Z=Z,
                        elbo_notebook=np.float64(-85.98812279560475), elbo_oracle=elbo,
                        n_params_notebook=np.array(2032), n_params_oracle=np.array(model.number_parameters()))
    print("notebook elbo", elbo, "params", model.number_parameters())
    make_case("case_A_nonwhite", D=2, num_units=[2, 2], Dy=1, white=False, seed=101)
    make_case("case_A_white", D=2, num_units=[2, 2], Dy=1, white=True, seed=102)
    make_case("case_B_nonwhite", D=3, num_units=[2, 4], Dy=2, white=False, seed=103)
    make_case("case_B_white", D=3, num_units=[2, 4], Dy=2, white=True, seed=104)
    make_training_case()
    make_mf_case()


if __name__ == "__main__":
    main()
