"""CPU tests that pin the oracle (no GPU): known answers of the reference's notebook, fixtures,
NumPy restatement vs torch-autograd twin vs finite differences, natgrad closed form."""
import numpy as np
import pytest

import dgp_oracle as O
import dgp_oracle_torch as T
from helpers import CASES, load, n_layers, notebook_data, oracle_from_golden


def test_notebook_known_answers():
    """nb_DGP_regression.ipynb cells 22/26 (first ELBO line) and cell 30."""
    X, Y, Z = notebook_data()
    g = load("notebook_known_answer")
    np.testing.assert_array_equal(X, g["X"])
    np.testing.assert_array_equal(Y, g["Y"])
    kernels = [O.RBF(1.0, [1.0] * u) for u in [1, 1, 1]]
    m = O.OracleDGP(X, Y, Z, kernels, [1, 1], num_samples=10)
    for seed in (0, 1, 2):                      # z-independent at construction
        elbo = m.ELBO(O.draw_zs(m, seed, 10, 50))
        assert abs(elbo - (-85.98812279560475)) < 1e-9
    assert m.number_parameters() == 2032
    # torch twin agrees, and so does its closed form  sum_n[-0.5 log 2pi - 0.5 (y^2 + 1)]
    et, _ = T.elbo_and_grads(m, O.draw_zs(m, 0, 10, 50), want_grads=False)
    assert abs(et - (-85.98812279560475)) < 1e-9
    closed = np.sum(-0.5 * np.log(2 * np.pi) - 0.5 * (Y ** 2 + 1.0))
    assert abs(closed - (-85.98812279560475)) < 1e-10


def test_q_sqrt_scaling_known_value():
    """with the q_sqrt*1e-3 line (dgp.py:268-269) each inner layer adds KL = 160.19."""
    X, Y, Z = notebook_data()
    m = O.OracleDGP(X, Y, Z, [O.RBF(1.0, [1.0]) for _ in range(3)], [1, 1], num_samples=10)
    for l in m.layers[:-1]:
        l.q_sqrt = l.q_sqrt * 1e-3
        kl = l.KL()
        assert abs(kl - (25 * np.log(1e3) - 12.5 + 12.5e-6)) < 1e-6


BO_NOTEBOOK_ELBO = -73.6722504558447        # nb_dgp_BO.ipynb cells 30 and 61: first `ELBO:` line of the constraint model


def bo_notebook_constraint_data(seed):
    """nb_dgp_BO.ipynb cells 4-6, 18: `Constrained_problem` (constraint = step at 0.25), DoE of 5 points on [0, 1]
    (the notebook's pyDOE draw is unseeded and not stored: any design with both constraint values present will do, the
    stored answer does not depend on it), inputs and constraint column standardised as SO_BO.py:35-39 (population std)."""
    from scipy.stats import qmc
    X = qmc.LatinHypercube(d=1, seed=seed).random(5)
    C = np.where(X > 0.25, 1.0, 0.0)
    assert 0 < C.sum() < 5
    Xn = (X - X.mean(axis=0)) / X.std(axis=0)
    Cn = (C - C.mean(axis=0)) / C.std(axis=0)
    return Xn, Cn


@pytest.mark.parametrize("seed", [0, 1, 3])
def test_bo_notebook_known_answer(seed):
    """The reference's second stored answer.  `SO_BO(problem, DoE_size=5, model_C_dic={'num_layers': 2, 'num_units': 1,
    'kernels': 'rbf', 'num_samples': 10})` builds `DGP(X, C, Z=X, ...)` (SO_BO.py:248); `train_models` ->
    `optimize_nat_adam` scales the inner layers' q_sqrt by 1e-3 (dgp.py:323-324) and prints the ELBO of that state before
    any update reaches it (dgp.py:327-333).  Unlike the -85.988 answer this one is taken at q != prior: it pins the
    non-white KL (layers.py:293-300), the 1e-3 scaling and the standardisation; the last layer is still at the prior, so
    the value is independent of z and of the design:  -5 (log(2 pi)/2 + 1) - 2 * 5 * (-1/2 - ln 1e-3 + 1e-6 / 2)."""
    from dgp_oracle_train import OracleTrainer
    X, C = bo_notebook_constraint_data(seed)
    assert abs(np.sum(C ** 2) - 5.0) < 1e-12
    m = O.OracleDGP(X, C, X.copy(), [O.RBF(1.0, [1.0]) for _ in range(3)], [1, 1], num_samples=10)
    tr = OracleTrainer(m, base_seed=seed)
    tr.scale_inner_q_sqrt()
    for zseed in (0, 7):
        elbo = m.ELBO(O.draw_zs(m, zseed, 10, 5))
        assert abs(elbo - BO_NOTEBOOK_ELBO) < 1e-9, elbo
    et, _ = T.elbo_and_grads(m, O.draw_zs(m, 0, 10, 5), want_grads=False)
    assert abs(et - BO_NOTEBOOK_ELBO) < 1e-9
    closed = -5 * (0.5 * np.log(2 * np.pi) + 1.0) - 2 * 5 * (-0.5 - np.log(1e-3) + 0.5e-6)
    assert abs(closed - BO_NOTEBOOK_ELBO) < 1e-12
    # each inner layer's KL on its own (layers.py:293-300), and what the trainer's first printed line is (dgp.py:327-333)
    for l in m.layers[:-1]:
        assert abs(l.KL() - 5 * (-0.5 - np.log(1e-3) + 0.5e-6)) < 1e-8
    assert abs(m.layers[-1].KL()) < 1e-9
    m2 = O.OracleDGP(X, C, X.copy(), [O.RBF(1.0, [1.0]) for _ in range(3)], [1, 1], num_samples=10)
    first = OracleTrainer(m2, base_seed=seed).optimize_nat_adam(1, 0, beta_1=0.8, beta_2=0.9, lr_gamma=0.01)[0]
    assert abs(first - BO_NOTEBOOK_ELBO) < 1e-9


@pytest.mark.parametrize("case", CASES)
def test_golden_forward_and_elbo(case):
    g = load(case)
    m = oracle_from_golden(g)
    zs = [g[f"zs{i}"] for i in range(n_layers(g))]
    Fs, Fm, Fv = m.propagate(g["X"], int(g["S"]), zs)
    for i in range(n_layers(g)):
        np.testing.assert_allclose(Fs[i], g[f"Fs{i}"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(Fm[i], g[f"Fmeans{i}"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(Fv[i], g[f"Fvars{i}"], rtol=1e-12, atol=1e-12)
    assert abs(m.ELBO(zs) - g["elbo"]) < 1e-9
    assert [l.mean_function.kind for l in m.layers] == [str(g[f"L{i}_mean_kind"]) for i in range(n_layers(g))]


@pytest.mark.parametrize("case", CASES)
def test_torch_twin_matches_and_gradients_by_finite_differences(case):
    g = load(case)
    m = oracle_from_golden(g)
    zs = [g[f"zs{i}"] for i in range(n_layers(g))]
    elbo, G = T.elbo_and_grads(m, zs)
    assert abs(elbo - g["elbo"]) < 1e-9
    elbo_c, Gc = T.elbo_and_grads(m, zs, chunk=17)            # chunking is numerically neutral
    assert abs(elbo_c - elbo) < 1e-9
    for i in range(n_layers(g)):
        for k in ("Z", "variance", "lengthscales", "q_mu", "q_sqrt"):
            np.testing.assert_allclose(G["layers"][i][k], g[f"g_L{i}_{k}"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(Gc["layers"][i][k], G["layers"][i][k], rtol=1e-8, atol=1e-8)
    h = 1e-6

    def fd(setter):
        setter(+h); ep = m.ELBO(zs); setter(-2 * h); em = m.ELBO(zs); setter(+h)
        return (ep - em) / (2 * h)

    for i, l in enumerate(m.layers):
        def sz(d, l=l): l.Z[1, 0] += d
        def sl(d, l=l): l.kern.lengthscales[0] += d
        def sv(d, l=l): l.kern.variance += d
        def sm(d, l=l): l.q_mu[2, 0] += d
        def sq(d, l=l): l.q_sqrt[0, 3, 1] += d
        for setter, val in ((sz, G["layers"][i]["Z"][1, 0]), (sl, G["layers"][i]["lengthscales"][0]),
                            (sv, G["layers"][i]["variance"]), (sm, G["layers"][i]["q_mu"][2, 0]),
                            (sq, G["layers"][i]["q_sqrt"][0, 3, 1])):
            assert abs(fd(setter) - val) < 2e-5 * max(1.0, abs(val))

    def slv(d): m.lik_variance += d
    assert abs(fd(slv) - G["lik_variance"]) < 2e-5 * max(1.0, abs(G["lik_variance"]))


@pytest.mark.parametrize("case", CASES)
def test_natgrad_closed_form_equals_gpflow_autodiff_route(case):
    g = load(case)
    m = oracle_from_golden(g)
    for i, l in enumerate(m.layers):
        a = O.natgrad_step(l.q_mu, l.q_sqrt, -g[f"g_L{i}_q_mu"], -g[f"g_L{i}_q_sqrt"], float(g["natgrad_gamma"]))
        b = T.natgrad_step_autodiff(l.q_mu, l.q_sqrt, -g[f"g_L{i}_q_mu"], -g[f"g_L{i}_q_sqrt"],
                                    float(g["natgrad_gamma"]))
        np.testing.assert_allclose(a[0], b[0], rtol=1e-8, atol=1e-9)      # two routes, cond(Kuu) ~ 1e6
        np.testing.assert_allclose(a[1], b[1], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(a[0], g[f"ng_L{i}_q_mu"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(a[1], g[f"ng_L{i}_q_sqrt"], rtol=1e-10, atol=1e-11)


def test_philox_normals_are_standard_and_index_keyed():
    z = O.philox_normal(3, 1, 4, np.arange(20000), 2)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    # keyed by the global point index: a shard sees exactly the slice of the full draw
    zs = O.philox_normal(3, 1, 4, np.arange(5000, 7000), 2)
    np.testing.assert_array_equal(zs, z[:, 5000:7000])
    # known-answer of Philox4x32-10 (Random123 kat_vectors: counter=key=0)
    r = O.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def test_layer_init_mean_functions():
    """layer_initializations.py:41-61: identity / PCA step-down / zero-pad step-up."""
    rng = np.random.default_rng(0)
    X = rng.standard_normal((40, 3)); Y = rng.standard_normal((40, 1)); Z = X[:7].copy()
    m = O.OracleDGP(X, Y, Z, [O.RBF(1.0, np.ones(d)) for d in (3, 2, 4)], [2, 4])
    kinds = [l.mean_function.kind for l in m.layers]
    assert kinds == ["linear", "linear", "zero"]
    _, _, V = np.linalg.svd(X, full_matrices=False)
    np.testing.assert_allclose(m.layers[0].mean_function.A, V[:2].T)
    np.testing.assert_array_equal(m.layers[1].mean_function.A, np.eye(2, 4))
    np.testing.assert_allclose(m.layers[1].Z, Z @ V[:2].T)
    np.testing.assert_allclose(m.layers[2].Z, Z @ V[:2].T @ np.eye(2, 4))


def test_matern_kernels_against_scikit_learn():
    """Independent pin of the restated gpflow Matern32/52 (the reference holds no fixture for them): scikit-learn's
    Matern(nu=1.5 / 2.5) with anisotropic length scales."""
    from sklearn.gaussian_process.kernels import Matern
    rng = np.random.default_rng(0)
    X, X2, ls = rng.standard_normal((9, 3)), rng.standard_normal((5, 3)), np.array([0.7, 1.3, 2.1])
    for cls, nu in ((O.Matern32, 1.5), (O.Matern52, 2.5)):
        k = cls(1.7, ls)
        np.testing.assert_allclose(k.K(X, X2), 1.7 * Matern(length_scale=ls, nu=nu)(X, X2), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(k.K(X), 1.7 * Matern(length_scale=ls, nu=nu)(X), rtol=1e-12, atol=1e-7)
        np.testing.assert_allclose(k.K_diag(X), np.full(9, 1.7))
        assert type(k.copy()) is cls


def test_matern_torch_twin_matches_numpy_oracle():
    import torch
    import dgp_oracle_torch as OT
    rng = np.random.default_rng(1)
    X, Z, ls = rng.standard_normal((6, 2)), rng.standard_normal((4, 2)), np.array([0.9, 1.4])
    for cls in (O.RBF, O.Matern32, O.Matern52):
        k = cls(1.2, ls)
        got = OT.rbf_K(torch.tensor(1.2, dtype=OT.DT), torch.tensor(ls), torch.tensor(Z), torch.tensor(X), kind=k.kind)
        np.testing.assert_allclose(got.numpy(), k.K(Z, X), rtol=1e-13)


def test_gpr_restatement_against_scikit_learn():
    """Independent pin of the exact-GP restatement (gpflow GPR as SO_BO.py:187-200 builds it)."""
    import gpr_oracle as G
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF as SkRBF, ConstantKernel, Matern
    rng = np.random.default_rng(3)
    X, Y, Xs = rng.uniform(-1, 1, (25, 3)), rng.standard_normal((25, 1)), rng.uniform(-1, 1, (7, 3))
    ls, var, noise = np.array([0.6, 1.1, 0.9]), 1.4, 3e-2
    for ok, sk in ((O.RBF(var, ls), SkRBF(ls)), (O.Matern32(var, ls), Matern(ls, nu=1.5)), (O.Matern52(var, ls), Matern(ls, nu=2.5))):
        gp = GaussianProcessRegressor(ConstantKernel(var, "fixed") * sk, alpha=noise, optimizer=None).fit(X, Y)
        assert abs(G.log_marginal_likelihood(ok, X, Y, noise) - gp.log_marginal_likelihood_value_) < 1e-9
        m, s = gp.predict(Xs, return_std=True)
        mean, v = G.predict_y(ok, X, Y, noise, Xs)
        np.testing.assert_allclose(mean[:, 0], np.ravel(m), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(v[:, 0] - noise, s ** 2, rtol=1e-7, atol=1e-10)
        lml, gv, gl, gn = G.lml_and_grads(ok, X, Y, noise)
        assert abs(lml - G.log_marginal_likelihood(ok, X, Y, noise)) < 1e-10
        h = 1e-6
        fd = (G.log_marginal_likelihood(ok, X, Y, noise + h) - G.log_marginal_likelihood(ok, X, Y, noise - h)) / (2 * h)
        assert abs(fd - gn) < 1e-5 * max(1.0, abs(fd))


def _mf_from_fixture(g):
    """Oracle parameter set, data and normals of tests/golden/mf_dgp_em_two_fidelities.npz."""
    import torch
    import mf_dgp_em_oracle as mo
    X, Y, X_red = [g["X0"], g["X1"]], [g["Y0"], g["Y1"]], [g["X_red0"]]
    P = mo.make_params(X, [x.copy() for x in X], [X[1].copy()])
    with torch.no_grad():
        for name, leaf in mo.leaves(P).items():
            leaf.copy_(torch.as_tensor(g["p." + name]).reshape(leaf.shape))
    nm = {"zright": [None, {"red": [g["zright_red0"]], "layers": [g["zright_lay0"]]}],
          "zs": [[g["zs0_0"]], [g["zs1_0"], g["zs1_1"]]], "ws": [[], [g["ws1_0"]]], "ws_proj": [[g["wsproj0_0"]]]}
    return P, X, Y, X_red, nm, int(g["S"])


def test_mf_dgp_em_restatement_reproduces_its_fixture():
    """The committed multi-fidelity vectors (inputs, injected normals, parameter state -> four terms of the bound and the
    gradients) are reproduced by oracle/mf_dgp_em_oracle.py: a guard against accidental changes of the restatement."""
    import mf_dgp_em_oracle as mo
    g = load("mf_dgp_em_two_fidelities")
    P, X, Y, X_red, nm, S = _mf_from_fixture(g)
    elbo, parts, grads = mo.elbo_and_grads(P, X, Y, X_red, nm, S)
    assert abs(elbo - float(g["elbo"])) <= 1e-10 * abs(float(g["elbo"]))
    for k, v in parts.items():
        assert abs(v - float(g["part." + k])) <= 1e-10 * max(1.0, abs(v))
    for k, v in grads.items():
        ref = g["g." + k]
        np.testing.assert_allclose(v, ref.reshape(v.shape), rtol=1e-9, atol=1e-9 * max(1.0, np.abs(ref).max()))


# ---------------------------------------------------------------------------------------------------------------
# Where the tolerance of the `ng_all=False` trajectory test (tests/test_gpu_parity.py) comes from.
# In that mode Adam (epsilon 1e-7) acts on the inner layers' q_sqrt.  With q_sqrt = 1e-3 chol(Kuu) the strictly-lower
# part of d KL / d q_sqrt = tril(Kuu^-1 L_q - diag(1 / diag L_q)) is zero in exact arithmetic (Kuu^-1 Lu = Lu^-T is upper
# triangular), so what an implementation hands to Adam there is its own rounding noise, and Adam turns a gradient g
# with |g| << epsilon into a step lr * g / epsilon: noise of 1e-10 becomes a 1e-5 step on entries of size 1e-3.
def _nglast_trajectory(noise=0.0, freeze_inner_q_sqrt=False, x_shift=0.0):
    from dgp_oracle_train import OracleTrainer
    from helpers import notebook_data
    X, Y, Z = notebook_data()
    mo = O.OracleDGP(X + x_shift, Y, Z, [O.RBF(1.0, [1.0]) for _ in range(3)], [1, 1], num_samples=10)
    tr = OracleTrainer(mo, base_seed=5)
    if freeze_inner_q_sqrt:
        for i in range(2):
            tr.trainable[(i, "q_sqrt")] = False
    if noise:
        rng = np.random.default_rng(0)

        def hook(flat):
            for i in range(2):
                g = flat[(i, "q_sqrt")]
                sl = np.tril_indices(g.shape[-1], -1)
                g[0][sl] += noise * rng.standard_normal(len(sl[0]))
        tr.grad_hook = hook
    return np.array(tr.optimize_nat_adam(3, 4, lr_adam=0.01, lr_gamma=0.01, beta_1=0.8, beta_2=0.9, ng_all=False))


def test_nglast_trajectory_tolerance_is_amplified_rounding_noise():
    base = _nglast_trajectory()
    # (1) the restatement's own entries there are far below 1e-10: moving the inputs by 1e-14 moves the trajectory by
    #     rounding only
    shifted = _nglast_trajectory(x_shift=1e-14)
    assert np.max(np.abs(shifted - base) / np.abs(base)) < 1e-8
    # (2) noise on exactly those entries is amplified ~4e6 times into the printed ELBOs (linear in the noise): 1e-12 ->
    #     4e-6, 1e-11 -> 4e-5.  The HIP path's entries there are ~3e-11 rms (asserted < 7e-11 in tests/test_gpu_parity.py, i.e.
    #     1e-15 of the largest gradient entry), hence that test's 3e-4; north_star's 1e-5 on THIS trajectory would need
    #     2e-12, below fp64 rounding of the Kuu^-1 L_q product at cond(Kuu) ~ 1e6.
    spreads = {}
    for noise in (1e-12, 1e-11):
        noisy = _nglast_trajectory(noise=noise)
        spreads[noise] = np.max(np.abs(noisy - base) / np.abs(base))
        assert abs(noisy[0] - base[0]) < 1e-12 * abs(base[0])      # the first evaluation precedes any update
    assert 1e-6 < spreads[1e-12] < 2e-5, spreads
    assert 1e-5 < spreads[1e-11] < 2e-4, spreads
    assert 5.0 < spreads[1e-11] / spreads[1e-12] < 20.0, spreads   # linear amplification
    # (3) with those entries out of Adam's reach the same noise is harmless: the tight variant of the GPU test
    frozen = _nglast_trajectory(freeze_inner_q_sqrt=True)
    frozen_noisy = _nglast_trajectory(noise=1e-11, freeze_inner_q_sqrt=True)
    assert np.max(np.abs(frozen_noisy - frozen) / np.abs(frozen)) < 1e-10


# ---------------------------------------------------------------------------------------------------------------
# Where the 1e-6 of tests/test_gpu_parity.py::test_adam_iterations_graph_replay_matches_call_by_call comes from.
# The three ways of running the same nine iterations issue the same kernels with the same seeds; what differs between two
# runs is the order of the fp64 atomic adds of the split-K reductions of this small model.  On the oracle alone: noise of
# relative size r (relative to the largest entry of a gradient block, i.e. also on the inner q_sqrt entries that are zero
# in exact arithmetic) moves the nine printed ELBOs of the Adam-on-everything loop by 1.07e10 * r, linearly from r = 1e-17
# to 1e-14: Adam (epsilon 1e-7, lr 1e-2) turns an entry of size 1e-16 * 1e5 into a step of 1e-6.  So 1e-6 is the response to
# ONE rounding unit (1.1e-16) of the largest entry; the HIP path's own run-to-run spread was measured at 2e-9 (only the last
# bits of a few partial sums move), a wrong seed or step count shows at 1e-2.  With the natural-gradient step owning
# (q_mu, q_sqrt) the same noise moves the ELBOs by < 1e-10.
def _replay_trajectory(rel_noise=0.0, seed=0, natgrad=False):
    from dgp_oracle_train import OracleTrainer
    from helpers import notebook_data
    X, Y, Z = notebook_data()
    mo = O.OracleDGP(X, Y, Z, [O.RBF(1.0, [1.0]) for _ in range(3)], [1, 1], num_samples=10)
    for l in mo.layers[:-1]:
        l.q_sqrt = l.q_sqrt * 1e-2
    tr = OracleTrainer(mo, base_seed=100)
    if natgrad:
        for i in range(3):
            tr.trainable[(i, "q_mu")] = False
            tr.trainable[(i, "q_sqrt")] = False
    if rel_noise:
        rng = np.random.default_rng(seed)

        def hook(flat):
            for k, g in flat.items():
                g = np.asarray(g)
                if g.ndim == 0:
                    continue
                g += rel_noise * (np.abs(g) + np.abs(g).max()) * rng.standard_normal(g.shape)
        tr.grad_hook = hook
    adam = tr.new_adam(0.01, 0.9, 0.999, 1e-7)
    out = []
    for _ in range(9):
        out.append(tr.adam_iteration(adam))
        if natgrad:
            tr.natgrad_iteration(0.01, [0, 1, 2])
    return np.array(out)


@pytest.mark.parametrize("natgrad", [False, True])
def test_graph_replay_tolerance_is_amplified_summation_order_noise(natgrad):
    base = _replay_trajectory(natgrad=natgrad)
    spread = {}
    for noise in (1e-16, 1e-15):
        spread[noise] = max(np.max(np.abs(_replay_trajectory(noise, seed, natgrad) - base) / np.abs(base)) for seed in (0, 1))
    if natgrad:
        assert max(spread.values()) < 1e-9, spread
        return
    amp = spread[1e-15] / 1e-15
    assert 3e9 < amp < 3e10, spread                                   # measured 1.07e10
    assert 5.0 < spread[1e-15] / spread[1e-16] < 20.0, spread        # linear
    assert 0.3e-6 < amp * 1.1e-16 < 3e-6                              # the GPU test's 1e-6 = one rounding unit of the largest entry


# ---------------------------------------------------------------------------------------------------------------
# A known answer that needs neither the reference nor the autograd twin: sparse GP regression's collapsed bound.
@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("shape", [(300, 2, 20, 1), (500, 3, 40, 2)])
def test_one_natural_gradient_step_of_size_one_reaches_the_collapsed_bound(shape, white):
    """A DGP without hidden layers is SVGP regression; with a Gaussian likelihood its ELBO is quadratic in q(u)'s natural parameters,
    so NaturalGradient(gamma=1).minimize lands on the optimal q(u) in one step (dgp.py:312-322,343 with gamma = 1), where the ELBO
    equals Titsias' collapsed bound.  Pins the oracle's natural-gradient step, its non-white KL at q != prior, the conditional and the
    Gaussian variational expectations against closed forms written from the textbook (tests/helpers.py::collapsed_bound)."""
    from dgp_oracle_train import OracleTrainer
    from helpers import collapsed_bound
    N, D, M, Dy = shape
    rng = np.random.default_rng(3)
    X = rng.standard_normal((N, D))
    Y = np.sin(2 * X[:, :1]) @ np.ones((1, Dy)) + 0.3 * rng.standard_normal((N, Dy))
    Z = X[:M].copy()
    ls = np.linspace(0.8, 1.2, D)
    mo = O.OracleDGP(X, Y, Z, [O.RBF(1.3, ls)], [], lik_variance=0.37, white=white, num_samples=3)
    assert len(mo.layers) == 1
    l = mo.layers[0]
    l.q_mu = 0.1 * rng.standard_normal(l.q_mu.shape)                  # start away from the prior
    l.q_sqrt = np.stack([np.tril(0.3 * np.eye(M) + 0.02 * rng.standard_normal((M, M))) for _ in range(Dy)])
    tr = OracleTrainer(mo, base_seed=5)
    e0, _ = tr._grads(tr._next_zs())
    tr.natgrad_iteration(1.0, [0])
    e1, _ = tr._grads(tr._next_zs())
    bound, m_opt, S_opt = collapsed_bound(X, Y, Z, 1.3, ls, 0.37, O.JITTER)
    assert e0 < e1 - 1.0
    assert abs(e1 - bound) < 1e-9 * abs(bound), (e1, bound)
    # (white=True: q is over v = Lu^-1 u, layers.py:238-241 - compare in u's coordinates)
    Lu = np.linalg.cholesky(O.RBF(1.3, ls).K(Z) + O.JITTER * np.eye(M)) if white else np.eye(M)
    assert np.abs(Lu @ l.q_mu - m_opt).max() < 1e-9
    for d in range(Dy):
        Ld = Lu @ np.tril(l.q_sqrt[d])
        assert np.abs(Ld @ Ld.T - S_opt).max() < 1e-9


def test_oracle_hyperparameter_gradients_at_the_optimal_q_are_those_of_the_collapsed_bound():
    """Envelope theorem (see the GPU test of the same name in test_gpu_parity.py): at the optimal q(u) the autograd twin's
    d ELBO / d(variance, lengthscales, noise, Z) must equal central differences of the textbook bound, and d ELBO / d q(u) must vanish."""
    from dgp_oracle_train import OracleTrainer
    from helpers import collapsed_bound
    N, D, M = 300, 2, 20
    rng = np.random.default_rng(3)
    X = rng.standard_normal((N, D))
    Y = np.sin(2 * X[:, :1]) + 0.3 * rng.standard_normal((N, 1))
    Z = X[:M].copy()
    ls = np.array([0.8, 1.2])
    mo = O.OracleDGP(X, Y, Z, [O.RBF(1.3, ls)], [], lik_variance=0.37, num_samples=2)
    tr = OracleTrainer(mo, base_seed=5)
    tr.natgrad_iteration(1.0, [0])
    _, G = tr._grads(tr._next_zs())
    h = 1e-4

    def fd(f):
        return (f(+h) - f(-h)) / (2.0 * h)
    want = {"variance": fd(lambda e: collapsed_bound(X, Y, Z, 1.3 + e, ls, 0.37, O.JITTER)[0]),
            "noise": fd(lambda e: collapsed_bound(X, Y, Z, 1.3, ls, 0.37 + e, O.JITTER)[0]),
            "ls0": fd(lambda e: collapsed_bound(X, Y, Z, 1.3, ls + e * np.array([1.0, 0.0]), 0.37, O.JITTER)[0]),
            "ls1": fd(lambda e: collapsed_bound(X, Y, Z, 1.3, ls + e * np.array([0.0, 1.0]), 0.37, O.JITTER)[0])}
    V = rng.standard_normal(Z.shape)
    V /= np.linalg.norm(V)
    want["Z"] = fd(lambda e: collapsed_bound(X, Y, Z + e * V, 1.3, ls, 0.37, O.JITTER)[0])
    got = {"variance": float(G[(0, "variance")]), "noise": float(G[("lik", "variance")]),
           "ls0": float(np.ravel(G[(0, "lengthscales")])[0]), "ls1": float(np.ravel(G[(0, "lengthscales")])[1]),
           "Z": float((np.asarray(G[(0, "Z")]) * V).sum())}
    scale = max(abs(v) for v in want.values())
    for k in want:
        assert abs(got[k] - want[k]) < 2e-6 * scale, (k, got[k], want[k])
    assert np.abs(G[(0, "q_mu")]).max() < 1e-6 * scale and np.abs(np.tril(np.asarray(G[(0, "q_sqrt")])[0])).max() < 1e-6 * scale


def test_oracle_two_layer_data_path_against_the_collapsed_bound_of_the_sampled_inputs():
    """The oracle's doubly-stochastic data path against the closed form of tests/test_gpu_parity.py::
    test_two_layer_model_with_given_normals_...: hidden layer at the prior (samples F1[s] = X + sqrt(sigma_1^2 + jitter) z[s]), one
    natural-gradient step of size one on the output layer, bound = collapsed bound on the S N sampled inputs with noise S sigma^2 + const."""
    import dgp_oracle_torch as T
    from helpers import collapsed_bound
    N, D, M, S = 400, 2, 30, 3
    rng = np.random.default_rng(9)
    X = rng.standard_normal((N, D))
    Y = np.sin(2 * X[:, :1]) + 0.3 * rng.standard_normal((N, 1))
    Z = X[rng.permutation(N)[:M]].copy()
    ls2 = np.array([0.9, 1.4])
    s1, s2, noise = 0.05, 1.1, 0.25
    mo = O.OracleDGP(X, Y, Z, [O.RBF(s1, np.ones(D)), O.RBF(s2, ls2)], [D], lik_variance=noise, num_samples=S)
    zs = [rng.standard_normal((S, N, D)), np.zeros((S, N, 1))]
    F1 = X[None] + np.sqrt(s1 + O.JITTER) * zs[0]
    Fs = mo.propagate(X, S, zs)[0]
    assert np.abs(np.asarray(Fs[0]) - F1).max() < 1e-12
    _, G = T.elbo_and_grads(mo, zs)
    l = mo.layers[1]
    l.q_mu, l.q_sqrt = O.natgrad_step(l.q_mu, l.q_sqrt, -np.asarray(G["layers"][1]["q_mu"]), -np.asarray(G["layers"][1]["q_sqrt"]), 1.0)
    bound, m_opt, S_opt = collapsed_bound(F1.reshape(S * N, D), np.tile(Y, (S, 1)), Z, s2, ls2, S * noise, O.JITTER)
    bound += S * N * (0.5 * np.log(2 * np.pi * S * noise) - 0.5 / S * np.log(2 * np.pi * noise))
    e = mo.ELBO(zs)
    assert abs(e - bound) < 1e-9 * abs(bound), (e, bound)
    assert np.abs(l.q_mu - m_opt).max() < 1e-9
    Ld = np.tril(l.q_sqrt[0])
    assert np.abs(Ld @ Ld.T - S_opt).max() < 1e-9


@pytest.mark.parametrize("white", [False, True])
def test_oracle_elbo_at_an_arbitrary_q_equals_the_textbook_svgp_bound(white):
    """One layer, random q(u): the oracle's ELBO (conditional_ND, KL, variational expectations: layers.py:227-308, dgp.py:89-100)
    against the SVGP bound written from Hensman et al. 2013 (tests/helpers.py::svgp_elbo)."""
    from helpers import svgp_elbo
    N, D, M, Dy = 250, 2, 18, 2
    rng = np.random.default_rng(4)
    X = rng.standard_normal((N, D))
    Y = np.stack([np.sin(2 * X[:, 0]), np.cos(X[:, 1])], 1) + 0.2 * rng.standard_normal((N, Dy))
    Z = X[:M].copy()
    ls = np.array([0.7, 1.3])
    mo = O.OracleDGP(X, Y, Z, [O.RBF(1.4, ls)], [], lik_variance=0.3, white=white, num_samples=2)
    l = mo.layers[0]
    l.q_mu = 0.4 * rng.standard_normal((M, Dy))
    l.q_sqrt = np.stack([np.tril(0.5 * np.eye(M) + 0.1 * rng.standard_normal((M, M))) for _ in range(Dy)])
    zs = [rng.standard_normal((2, N, Dy))]
    want = svgp_elbo(X, Y, Z, 1.4, ls, 0.3, l.q_mu, l.q_sqrt, O.JITTER, white=white)
    got = mo.ELBO(zs)
    assert abs(got - want) < 1e-10 * abs(want), (got, want)


def test_oracle_two_layer_elbo_at_arbitrary_q_equals_the_bound_written_from_the_paper():
    """Two layers, random q(u) in both, given normals: the oracle's ELBO against tests/helpers.py::dsdgp2_elbo (Salimbeni & Deisenroth
    2017 eq. 13-16 assembled from the SVGP marginals of Hensman et al. 2013)."""
    from helpers import dsdgp2_elbo
    N, D, M, S = 300, 2, 20, 3
    rng = np.random.default_rng(6)
    X = rng.standard_normal((N, D))
    Y = np.sin(2 * X[:, :1]) + 0.3 * rng.standard_normal((N, 1))
    Z = X[rng.permutation(N)[:M]].copy()
    mo = O.OracleDGP(X, Y, Z, [O.RBF(0.6, np.array([0.9, 1.2])), O.RBF(1.1, np.array([1.3, 0.8]))], [D], lik_variance=0.25, num_samples=S)
    lay = []
    for l, dout in zip(mo.layers, (D, 1)):
        l.build_cholesky()
        l.q_mu = l.Lu @ (0.4 * rng.standard_normal((M, dout)))
        l.q_sqrt = np.stack([np.tril(l.Lu @ np.tril(0.5 * np.eye(M) + 0.1 * rng.standard_normal((M, M)))) for _ in range(dout)])
        lay.append(dict(Z=np.asarray(l.Z), variance=l.kern.variance, lengthscales=np.asarray(l.kern.lengthscales), q_mu=l.q_mu, q_sqrt=l.q_sqrt))
    zs = [rng.standard_normal((S, N, D)), np.zeros((S, N, 1))]
    want = dsdgp2_elbo(X, Y, zs[0], lay[0], lay[1], 0.25, O.JITTER)
    got = mo.ELBO(zs)
    assert abs(got - want) < 1e-10 * abs(want), (got, want)


def test_oracle_three_layer_elbo_equals_the_bound_for_any_depth():
    """tests/helpers.py::dsdgp_elbo (the function bench.py's elbo_vs_closed_form and the full-size GPU test use) against the oracle on a
    three-layer model with a random q(u) in every layer and given normals."""
    from helpers import dsdgp_elbo
    N, D, M, S = 200, 2, 15, 3
    rng = np.random.default_rng(7)
    X = rng.standard_normal((N, D))
    Y = np.sin(2 * X[:, :1]) + 0.3 * rng.standard_normal((N, 1))
    Z = X[rng.permutation(N)[:M]].copy()
    kp = [(0.5 + 0.2 * i, np.array([0.9, 1.3]) + 0.05 * i) for i in range(3)]
    mo = O.OracleDGP(X, Y, Z, [O.RBF(v, l) for v, l in kp], [D, D], lik_variance=0.25, num_samples=S)
    lay = []
    for l, dout in zip(mo.layers, (D, D, 1)):
        l.build_cholesky()
        l.q_mu = l.Lu @ (0.3 * rng.standard_normal((M, dout)))
        l.q_sqrt = np.stack([np.tril(l.Lu @ np.tril(0.4 * np.eye(M) + 0.03 * rng.standard_normal((M, M)))) for _ in range(dout)])
        lay.append(dict(Z=np.asarray(l.Z), variance=l.kern.variance, lengthscales=np.asarray(l.kern.lengthscales), q_mu=l.q_mu, q_sqrt=l.q_sqrt))
    zs = [rng.standard_normal((S, N, D)), rng.standard_normal((S, N, D)), np.zeros((S, N, 1))]
    want = dsdgp_elbo(X, Y, zs, lay, 0.25, O.JITTER)
    got = mo.ELBO(zs)
    assert abs(got - want) < 1e-10 * abs(want), (got, want)


@pytest.mark.parametrize("gamma", [0.01, 0.3])
def test_oracle_natural_gradient_step_interpolates_the_natural_parameters(gamma):
    """The oracle's natgrad_step at any step size against the convex combination of natural parameters a conjugate model requires
    (see the GPU test of the same name): theta_new = (1 - gamma) theta_0 + gamma theta_opt, theta_opt from the collapsed bound."""
    from dgp_oracle_train import OracleTrainer
    from helpers import collapsed_bound
    N, D, M = 300, 2, 20
    rng = np.random.default_rng(3)
    X = rng.standard_normal((N, D))
    Y = np.sin(2 * X[:, :1]) + 0.3 * rng.standard_normal((N, 1))
    Z = X[:M].copy()
    ls = np.array([0.8, 1.2])
    mo = O.OracleDGP(X, Y, Z, [O.RBF(1.3, ls)], [], lik_variance=0.37, num_samples=2)
    l = mo.layers[0]
    l.q_mu = 0.1 * rng.standard_normal((M, 1))
    l.q_sqrt = np.tril(0.3 * np.eye(M) + 0.02 * rng.standard_normal((M, M)))[None]
    m0, L0 = l.q_mu.copy(), l.q_sqrt[0].copy()
    OracleTrainer(mo, base_seed=5).natgrad_iteration(gamma, [0])
    _, m_opt, S_opt = collapsed_bound(X, Y, Z, 1.3, ls, 0.37, O.JITTER)
    P0, P_opt = np.linalg.inv(L0 @ L0.T), np.linalg.inv(S_opt)
    S_want = np.linalg.inv((1 - gamma) * P0 + gamma * P_opt)
    m_want = S_want @ ((1 - gamma) * P0 @ m0 + gamma * P_opt @ m_opt)
    Ln = np.tril(l.q_sqrt[0])
    assert np.abs(Ln @ Ln.T - S_want).max() < 1e-9 and np.abs(l.q_mu - m_want).max() < 1e-9
