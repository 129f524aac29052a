"""Reference training loops restated on top of the oracle (TEST INFRASTRUCTURE ONLY).

Follows ``dgp_dace/models/dgp.py:255-345`` (``DGP.optimize_adam`` / ``DGP.optimize_nat_adam``):
gradients by autograd (the reference: ``tf.GradientTape``), Keras-Adam on the *unconstrained*
variables (GPflow ``Parameter`` transforms: Softplus for kernel variance/lengthscales,
Softplus+Shift(1e-6) for the Gaussian variance, FillTriangular for q_sqrt, identity for q_mu/Z)
and the XiNat natural-gradient step on (q_mu, q_sqrt) pairs.  MC normals: every ELBO evaluation
``e`` (0,1,2,…) uses ``draw_zs(model, base_seed + e, …)`` — the convention the product uses.
"""
from __future__ import annotations

import numpy as np

import dgp_oracle as O
import dgp_oracle_torch as T


class OracleTrainer:
    def __init__(self, model, base_seed=0):
        self.model = model
        self.base_seed = int(base_seed)
        self.eval_count = 0
        # set_trainable flags, keyed (layer, name); persistent like gpflow's
        self.trainable = {}
        for i, _ in enumerate(model.layers):
            for k in ("Z", "variance", "lengthscales", "q_mu", "q_sqrt"):
                self.trainable[(i, k)] = True
        self.trainable[("lik", "variance")] = True
        self.grad_hook = None

    # -- helpers -----------------------------------------------------------------------
    def _next_zs(self):
        X = self.model.data[0]
        zs = O.draw_zs(self.model, self.base_seed + self.eval_count, self.model.num_samples, X.shape[0])
        self.eval_count += 1
        return zs

    def _get(self, key):
        i, k = key
        if i == "lik":
            return np.asarray(self.model.lik_variance)
        l = self.model.layers[i]
        return {"Z": l.Z, "variance": np.asarray(l.kern.variance), "lengthscales": l.kern.lengthscales,
                "q_mu": l.q_mu, "q_sqrt": l.q_sqrt}[k]

    def _set(self, key, val):
        i, k = key
        if i == "lik":
            self.model.lik_variance = float(val)
            return
        l = self.model.layers[i]
        if k == "Z":
            l.Z = val
        elif k == "variance":
            l.kern.variance = float(val)
        elif k == "lengthscales":
            l.kern.lengthscales = val
        elif k == "q_mu":
            l.q_mu = val
        else:
            l.q_sqrt = np.tril(val)

    @staticmethod
    def _to_u(key, x):
        if key[1] in ("variance", "lengthscales"):
            shift = O.LIK_VAR_LOWER if key[0] == "lik" else 0.0
            return np.array(O.softplus_inv(np.asarray(x, dtype=np.float64) - shift), dtype=np.float64)
        return np.array(x, dtype=np.float64)

    @staticmethod
    def _to_x(key, u):
        if key[1] in ("variance", "lengthscales"):
            shift = O.LIK_VAR_LOWER if key[0] == "lik" else 0.0
            return O.softplus(u) + shift
        return u

    @staticmethod
    def _dx_du(key, u):
        if key[1] in ("variance", "lengthscales"):
            return 1.0 / (1.0 + np.exp(-u))
        return 1.0

    def _grads(self, zs):
        elbo, G = T.elbo_and_grads(self.model, zs)
        flat = {("lik", "variance"): G["lik_variance"]}
        for i, g in enumerate(G["layers"]):
            for k, v in g.items():
                flat[(i, k)] = v
        if self.grad_hook is not None:      # tests only: e.g. rounding-level noise on chosen entries (tests/test_oracle.py)
            self.grad_hook(flat)
        return elbo, flat

    # -- dgp.py:255-278 ------------------------------------------------------------------
    def scale_inner_q_sqrt(self):
        for layer in self.model.layers[:-1]:
            layer.q_sqrt = layer.q_sqrt * 1e-3                      # dgp.py:268-269,323-324

    def new_adam(self, lr=0.01, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        return {"lr": lr, "beta_1": beta_1, "beta_2": beta_2, "epsilon": epsilon, "t": 0, "m": {}, "v": {}}

    def adam_iteration(self, adam):
        """One loop body of dgp.py:271-276: returns the ELBO that would be printed."""
        zs = self._next_zs()
        elbo, G = self._grads(zs)
        adam["t"] += 1
        for key, on in self.trainable.items():
            if not on:
                continue
            u = self._to_u(key, self._get(key))
            g_u = -np.asarray(G[key]) * self._dx_du(key, u)          # objective = -ELBO
            if key[1] == "q_sqrt":
                g_u = np.tril(g_u)
            m = adam["m"].setdefault(key, np.zeros_like(u))
            v = adam["v"].setdefault(key, np.zeros_like(u))
            O.adam_update(u, m, v, g_u, adam["t"], adam["lr"], adam["beta_1"], adam["beta_2"], adam["epsilon"])
            self._set(key, self._to_x(key, u))
        return elbo

    def natgrad_iteration(self, gamma, layer_ids):
        """optimizer_nat.minimize(objective_nat, var_list)   (dgp.py:343): one fresh ELBO evaluation."""
        zs = self._next_zs()
        _, G = self._grads(zs)
        for i in layer_ids:
            l = self.model.layers[i]
            l.q_mu, l.q_sqrt = O.natgrad_step(l.q_mu, l.q_sqrt, -G[(i, "q_mu")], -G[(i, "q_sqrt")], gamma)

    def optimize_adam(self, iterations, lr=0.01, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.scale_inner_q_sqrt()
        adam = self.new_adam(lr, beta_1, beta_2, epsilon)
        return [self.adam_iteration(adam) for _ in range(iterations)]

    # -- dgp.py:280-345 ------------------------------------------------------------------
    def optimize_nat_adam(self, iterations1, iterations2, lr_adam=0.01, lr_gamma=0.01, beta_1=0.9,
                          beta_2=0.999, epsilon=1e-7, ng_all=True):
        nl = len(self.model.layers)
        ng_layers = list(range(nl)) if ng_all else [nl - 1]
        for i in ng_layers:
            self.trainable[(i, "q_mu")] = False
            self.trainable[(i, "q_sqrt")] = False
        self.scale_inner_q_sqrt()
        adam = self.new_adam(lr_adam, beta_1, beta_2, epsilon)
        out = [self.adam_iteration(adam) for _ in range(iterations1)]
        for _ in range(iterations2):
            out.append(self.adam_iteration(adam))
            self.natgrad_iteration(lr_gamma, ng_layers)
        return out
