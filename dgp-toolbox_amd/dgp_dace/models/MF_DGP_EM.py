"""Multi-fidelity deep GP with input-space mapping (MF-DGP-EM) on the HIP engine.

Mirror of the reference's ``dgp_dace/models/MF_DGP_EM.py`` (Cutajar et al. multi-fidelity DGP plus a GP that maps the
inputs of a fidelity to those of the one below): same classes, constructor arguments, attribute paths and training
phases.  The reference evaluates everything in TensorFlow/GPflow (with a locally patched ``InducingPoints`` that carries
``Z_left`` / ``Z_right``); here every SVGP layer lives in its own single-layer device context of libdgp_hip.so and
this module only walks the model's graph:

* forward   ``dgp_propagate`` per layer call (MF_DGP_EM.py:123-168), the host concatenates ``[H, F]`` between layers;
* backward  ``dgp_vjp_accumulate`` per layer call in reverse order (what ``tape.gradient`` does at :460-464): the
            x-gradient feeds the producing layers, the parameter sums pile up in each layer's accumulator and one
            ``dgp_grad_finish`` per layer adds the KL part;
* the likelihood terms (:205-260) and the optimisers' bookkeeping are host NumPy on [S, N, 1] arrays.

The layer arithmetic itself (kernel matrices incl. the composite ``k_corr (k_prev + Linear) + k_in`` kernel and the
White variance, Cholesky, whitened solves, mean/variance, sampling, KL and all their gradients) runs on the GPU; there
is no CPU fallback.

Random numbers: the reference draws ``tf.random.normal`` inside ``ELBO`` (100 samples per likelihood term, the mean of
50 samples for ``Z_right``, :33-35).  Here they come from a NumPy generator (or are injected through ``normals=`` for
the parity tests); the mean over the 50 samples at identical inputs is taken on the draws (``z̄``), which is the same
random variable.
"""
from __future__ import annotations

import numpy as np

from .. import _native
from ..gpflow_compat import (Gaussian, LinearKernel, Parameter, Product, SquaredExponential, Sum, White, Zero,
                             as_tensor, set_trainable)
from ..utils.utils import BroadcastingLikelihood

RBF = SquaredExponential
JITTER = 1e-6


# ------------------------------------------------------------------------------------------ kernels -> C-ABI
def _kernel_plan(kern, Din):
    """(dgp_kernel_kind, [Parameter or None in the packing order of include/dgp_abi.h], White variance Parameter or None)."""
    terms = list(kern.kernels) if isinstance(kern, Sum) else [kern]
    white = [k for k in terms if isinstance(k, White)]
    rest = [k for k in terms if not isinstance(k, White)]
    if len(white) > 1:
        raise NotImplementedError("more than one White term")
    wv = white[0].variance if white else None
    if len(rest) == 1 and isinstance(rest[0], SquaredExponential) and rest[0].kind == "rbf":
        k = rest[0]
        if k.lengthscales.shape == (1,) and Din > 1:
            k.lengthscales = Parameter(np.full(Din, k.lengthscales._value[0]), "lengthscales", "softplus")
        if k.active_dims not in (None, list(range(Din))):
            raise NotImplementedError("an RBF layer kernel must act on all input dimensions")
        return 0, [k.variance, k.lengthscales], wv
    prods = [k for k in rest if isinstance(k, Product)]
    rbfs = [k for k in rest if isinstance(k, SquaredExponential)]
    if len(rest) == 2 and len(prods) == 1 and len(rbfs) == 1 and len(prods[0].kernels) == 2:
        Dx = Din - 1
        k_corr, inner = prods[0].kernels
        k_in = rbfs[0]
        inner_terms = list(inner.kernels) if isinstance(inner, Sum) else [inner]
        k_prev = [k for k in inner_terms if isinstance(k, SquaredExponential)]
        k_lin = [k for k in inner_terms if isinstance(k, LinearKernel)]
        ok = (isinstance(k_corr, SquaredExponential) and len(k_prev) == 1 and len(k_lin) <= 1 and
              len(inner_terms) == len(k_prev) + len(k_lin) and k_corr.active_dims == list(range(Dx)) and
              k_in.active_dims == list(range(Dx)) and k_prev[0].active_dims == [Dx] and
              all(k.active_dims == [Dx] for k in k_lin) and
              all(k.lengthscales.shape == (1,) for k in (k_corr, k_in, k_prev[0])))
        if ok:
            lin = k_lin[0].variance if k_lin else None
            return 3, [k_corr.variance, k_corr.lengthscales, k_prev[0].variance, k_prev[0].lengthscales, lin, k_in.variance,
                       k_in.lengthscales], wv
    raise NotImplementedError("layer kernel: RBF (+ White) or k_corr * (k_prev [+ Linear]) + k_in (+ White) as "
                              "MF_DGP_EM.make_mf_dgp builds them")


def _rbf_np(X, X2, variance, ls):
    Xs, X2s = X / ls, X2 / ls
    r2 = -2.0 * Xs @ X2s.T + (Xs * Xs).sum(-1)[:, None] + (X2s * X2s).sum(-1)[None, :]
    return variance * np.exp(-0.5 * r2)


def _kernel_matrix_host(kind, vals, white, Z):
    """K(Z) in NumPy for the constructor-time prior initialisation of q_sqrt (layers_red.py:213-216 does it on the host)."""
    if kind == 0:
        K = _rbf_np(Z, Z, vals[0], vals[1])
    else:
        x, f = Z[:, :-1], Z[:, -1:]
        K = _rbf_np(x, x, vals[0], vals[1]) * (_rbf_np(f, f, vals[2], vals[3]) + vals[4] * (f @ f.T)) + \
            _rbf_np(x, x, vals[5], vals[6])
    return K + (white if white is not None else 0.0) * np.eye(Z.shape[0])


# ------------------------------------------------------------------------------------------ layers
class InducingPoints:
    """The reference's patched ``gpflow.inducing_variables.InducingPoints(layers=, layers_red=, Z=)``, restated from its
    call sites (MF_DGP_EM.py:270-271,454-456): ``Z_left`` is the trainable [M, D_x] parameter, ``Z_right`` the [M, 1]
    column sampled through the earlier layers, ``Z`` what Kuu / Kuf see."""

    def __init__(self, Z, augmented=False):
        self.augmented = bool(augmented)
        if augmented:
            self.Z_left = Parameter(np.array(Z, dtype=np.float64), "Z_left")
            self.Z_right = np.zeros((self.Z_left.shape[0], 1))
        else:
            self._Z = Parameter(np.array(Z, dtype=np.float64), "Z")

    @property
    def Z(self):
        if self.augmented:
            return as_tensor(np.concatenate([self.Z_left._value, np.asarray(self.Z_right)], 1))
        return self._Z

    @Z.setter
    def Z(self, value):
        if self.augmented:          # `feature.Z = tf.concat([Z_left, Z_right], 1)` (MF_DGP_EM.py:271): derived, nothing to keep
            return
        self._Z.assign(value)

    def left(self):
        return self.Z_left if self.augmented else self._Z


class SVGP_Layer:
    """State of one layer (layers_red.py:163-222) + its private single-layer device context."""

    def __init__(self, kern, Z, num_outputs, mean_function=None, augmented=False, layers=None, layers_red=None,
                 white=False, **_):
        if white:
            raise NotImplementedError("the multi-fidelity models use white=False (layers_red.py:164)")
        self.kern = kern
        self.num_outputs = int(num_outputs)
        self.mean_function = mean_function if mean_function is not None else Zero()
        self.white = False
        self.feature = InducingPoints(Z.numpy() if hasattr(Z, "numpy") else Z, augmented)
        self.num_inducing = self.feature.left().shape[0]
        self.input_dim = self.feature.left().shape[1] + (1 if augmented else 0)
        self.kind, self._kpars, self._white = _kernel_plan(kern, self.input_dim)
        M = self.num_inducing
        self.q_mu = Parameter(np.zeros((M, self.num_outputs)), "q_mu")
        self.q_sqrt = Parameter(np.tile(np.eye(M)[None], [self.num_outputs, 1, 1]), "q_sqrt", "tril")
        self._ctx = None
        self._shape = None
        self._fresh = True

    # -- packing (include/dgp_abi.h: dgp_model_set) --
    def _kvals(self):
        return [np.zeros(1) if p is None else np.atleast_1d(p._value) for p in self._kpars]

    def _flat(self):
        parts = [np.asarray(self.feature.Z.numpy()).ravel()] + [v.ravel() for v in self._kvals()]
        if self._white is not None:
            parts.append(np.atleast_1d(self._white._value))
        parts += [self.q_mu._value.ravel(), self.q_sqrt._value.ravel(), np.ones(1)]
        return np.concatenate(parts)

    def sync(self):
        """Push the current parameter values to the device context (and start a new gradient accumulation)."""
        if self._ctx is None:
            self._ctx = _native.Context(0)
        desc = (self.input_dim, self.num_outputs, self.num_inducing, 0, self.kind, 0, 1 if self._white is not None else 0)
        flat = self._flat()
        if self._shape != (desc, flat.size):
            self._ctx.model_set([desc], flat, None)
            self._shape = (desc, flat.size)
        else:
            self._ctx.params_set(flat)
        self._fresh = True

    # -- one evaluation at [P, D_in] points with given normals [P, D_out] --
    def forward(self, X, z):
        Fs, Fm, Fv = self._ctx.propagate(X, 1, 0, [z[None]])
        return Fs[0][0], Fm[0][0], Fv[0][0]

    def forward_full_cov(self, X, z):
        """full_cov=True for ONE sample (layers_red.py:63-80 maps the layer over the samples): X [N, D_in], z [N, D_out]
        -> sample mean + chol(V + jitter) z [N, D], mean [N, D], covariance [N, N, D] (dgp_propagate_full_cov)."""
        Fs, Fm, Fv = self._ctx.propagate_full_cov(X, 1, 0, [z[None]])
        return Fs[0][0], Fm[0][0], Fv[0][0]

    def backward(self, X, z, f_bar=None, mean_bar=None, var_bar=None):
        """x-gradient of sum(cotangent * output); the parameter sums are added to this layer's accumulator."""
        bars = [None if b is None else np.ascontiguousarray(b)[None] for b in (f_bar, mean_bar, var_bar)]
        out = self._ctx.propagate_vjp(X, 1, 0, [z[None]], f_bar=bars[0], mean_bar=bars[1], var_bar=bars[2],
                                      accumulate="reset" if self._fresh else "add")
        self._fresh = False
        return out

    def finish(self):
        """(KL, {Parameter-id: gradient of [sum of the accumulated data terms - KL]}, d/dZ_full)."""
        if self._fresh:        # no data term reached this layer: KL part only
            P = 1
            self.backward(np.zeros((P, self.input_dim)), np.zeros((P, self.num_outputs)),
                          mean_bar=np.zeros((P, self.num_outputs)))
        kl = -self._ctx.grad_finish(want_elbo=True)
        g = self._ctx.grad_get()
        M, Din, D = self.num_inducing, self.input_dim, self.num_outputs
        out, off = {}, M * Din
        gZ = g[:off].reshape(M, Din)
        for p in self._kpars:
            n = 1 if p is None else p._value.size
            if p is not None:
                out[id(p)] = out.get(id(p), 0.0) + g[off:off + n].reshape(p._value.shape)
            off += n
        if self._white is not None:
            out[id(self._white)] = g[off:off + 1].reshape(())
            off += 1
        out[id(self.q_mu)] = g[off:off + M * D].reshape(M, D)
        off += M * D
        out[id(self.q_sqrt)] = g[off:off + D * M * M].reshape(D, M, M)
        return kl, out, gZ

    def KL(self):
        self.sync()
        return self.finish()[0]

    def natgrad_step(self, gamma):
        """gpflow NaturalGradient on this layer's (q_mu, q_sqrt) from the gradient of the last `finish` (XiNat closed
        form, dgp_natgrad_step); the new values are read back into the host parameters."""
        self._ctx.natgrad_step(gamma, [1])
        flat = self._ctx.params_get()
        M, D = self.num_inducing, self.num_outputs
        n_q = M * D + D * M * M
        q = flat[-1 - n_q:-1]
        self.q_mu._value = q[:M * D].reshape(M, D).copy()
        self.q_sqrt._value = np.tril(q[M * D:].reshape(D, M, M))

    def parameters(self):
        ps = [self.feature.left()] + [p for p in self._kpars if p is not None]
        if self._white is not None:
            ps.append(self._white)
        return ps + [self.q_mu, self.q_sqrt]


def init_layers_mf(X, Z, W, kernels, kernels_red, num_outputs=None, Layer=SVGP_Layer):
    """MF_DGP_EM.py:60-86."""
    num_outputs = num_outputs or 1
    layers, layers_red = [], []
    for i in range(1, len(X)):
        layers_red.append(Layer(kernels_red[i - 1], W[i - 1], X[-(1 + i)].shape[1], Zero()))
    L = len(layers_red)
    layers.append(Layer(kernels[0], Z[0], num_outputs, Zero()))
    for i in range(1, len(Z)):
        layers.append(Layer(kernels[i], Z[i], num_outputs, Zero(), augmented=True, layers=layers[:i],
                            layers_red=layers_red[L - i:]))
    return layers, layers_red


# ------------------------------------------------------------------------------------------ the model
def _ve(mean, var, Y, s2):
    return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(s2) - 0.5 * ((Y[None] - mean) ** 2 + var) / s2


class DGP_Base:
    """MF_DGP_EM.py:89-381."""

    def __init__(self, likelihood, layers, layers_red, minibatch_size=None, num_samples=1, seed=0, **kwargs):
        self.minibatch_size = minibatch_size
        self.num_samples = num_samples
        self._train_upto_fidelity = -1
        self.num_layers = len(layers)
        self.layers = layers
        self.layers_red = layers_red
        self.likelihood = BroadcastingLikelihood(likelihood)
        self.likelihood_projection = BroadcastingLikelihood(Gaussian())
        self.rng = np.random.default_rng(seed)
        self.L = self.KL = self.L_red = self.KL_red = 0.0

    # ---- random draws of one ELBO evaluation (one dict per evaluation; tests inject the same layout) ----
    def _draw_zright(self, n_zright=50):
        n, L, rng = self.num_layers, len(self.layers_red), self.rng
        out = [None]
        for i in range(1, n):
            Mi = self.layers[i].num_inducing
            out.append({
                "red": [rng.standard_normal((n_zright, Mi, lr.num_outputs)).mean(0) for lr in self.layers_red[L - i:]],
                "layers": [rng.standard_normal((n_zright, Mi, 1)).mean(0) for _ in range(i)]})
        return out

    def draw_normals(self, X, S, n_zright=50):
        n, L, rng = self.num_layers, len(self.layers_red), self.rng
        out = {"zright": self._draw_zright(n_zright), "zs": [], "ws": [], "ws_proj": []}
        for f in range(n):
            N = X[f].shape[0]
            out["ws"].append([rng.standard_normal((S, N, lr.num_outputs)) for lr in self.layers_red[L - f:]])
            out["zs"].append([rng.standard_normal((S, N, 1)) for _ in range(f + 1)])
            if f < n - 1:
                Nn = X[f + 1].shape[0]
                out["ws_proj"].append([rng.standard_normal((S, Nn, lr.num_outputs)) for lr in self.layers_red[L - (f + 1):]])
        return out

    # ---- Z_right (MF_DGP_EM.py:38-58): forward records what the backward needs ----
    def _z_right_forward(self, i, zbar):
        L = len(self.layers_red)
        H = self.layers[i].feature.Z_left._value
        rec = {"red": [], "lay": []}
        Hs = [H]
        for j, lr in enumerate(self.layers_red[L - i:]):
            Hn = lr.forward(H, zbar["red"][j])[0]
            rec["red"].append((lr, H, zbar["red"][j]))
            H = Hn
            Hs.append(H)
        Zr = None
        for j in range(i):
            inp = Hs[-1] if j == 0 else np.concatenate([Hs[-(j + 1)], Zr], 1)
            Zr = self.layers[j].forward(inp, zbar["layers"][j])[0]
            rec["lay"].append((self.layers[j], inp, zbar["layers"][j]))
        rec["nH"] = len(Hs)
        return Zr, rec

    def _z_right_backward(self, rec, Zr_bar):
        """Cotangent of Z_right back through the chain; returns the cotangent of Z_left (the chain's input)."""
        nH = rec["nH"]
        Hbar = [0.0] * nH
        fbar = Zr_bar
        for j in range(len(rec["lay"]) - 1, -1, -1):
            lay, inp, z = rec["lay"][j]
            xb = lay.backward(inp, z, f_bar=fbar)
            if j == 0:
                Hbar[nH - 1] = Hbar[nH - 1] + xb
            else:
                Dx = inp.shape[1] - 1
                Hbar[nH - (j + 1)] = Hbar[nH - (j + 1)] + xb[:, :Dx]
                fbar = xb[:, Dx:]
        for j in range(len(rec["red"]) - 1, -1, -1):
            lr, Hin, w = rec["red"][j]
            hb = Hbar[j + 1]
            if isinstance(hb, float):
                continue
            Hbar[j] = Hbar[j] + lr.backward(Hin, w, f_bar=hb)
        return Hbar[0]

    def update_Z_right(self, normals=None, keep=False):
        """`layers[i].feature.Z_right = sample_Z_right(...)` for every augmented layer (MF_DGP_EM.py:269-271)."""
        recs = [None]
        for lay in self.layers_red:
            lay.sync()
        self.layers[0].sync()
        for i in range(1, self.num_layers):
            zb = normals["zright"][i] if normals is not None else self._draw_zright()[i]
            Zr, rec = self._z_right_forward(i, zb)
            self.layers[i].feature.Z_right = Zr
            self.layers[i].sync()
            recs.append(rec)
        return recs if keep else None

    def _sync_all(self):
        """Push every layer's current state (with the stored Z_right) to its device context."""
        for lay in list(self.layers_red) + list(self.layers):
            lay.sync()

    # ---- propagate (MF_DGP_EM.py:123-168) on flattened [S*N, D] arrays, recording the calls ----
    def _chain(self, X, S, zs, ws, fidelity_dim, project=False):
        N = X.shape[0]
        L = len(self.layers_red)
        H = np.tile(X[None], [S, 1, 1]).reshape(S * N, -1)
        rec = {"red": [], "lay": [], "S": S, "N": N}
        Hs, Hm, Hv = [H], [], []
        for j, lr in enumerate(self.layers_red[L - fidelity_dim:]):
            w = ws[j].reshape(S * N, -1)
            Hn, m, v = lr.forward(H, w)
            rec["red"].append((lr, H, w))
            H = Hn
            Hs.append(H); Hm.append(m); Hv.append(v)
        rec["nH"] = len(Hs)
        if project:
            return Hs, Hm, Hv, rec
        Fs, Fm, Fv = [], [], []
        F = None
        for i in range(fidelity_dim + 1):
            inp = Hs[-1] if i == 0 else np.concatenate([Hs[-(i + 1)], F], 1)
            z = zs[i].reshape(S * N, -1)
            F, m, v = self.layers[i].forward(inp, z)
            rec["lay"].append((self.layers[i], inp, z))
            Fs.append(F); Fm.append(m); Fv.append(v)
        return Fs, Fm, Fv, rec

    def _chain_backward(self, rec, mean_bar, var_bar, project=False):
        nH = rec["nH"]
        Hbar = [0.0] * nH
        if project:
            lr, Hin, w = rec["red"][-1]
            Hbar[nH - 2] = Hbar[nH - 2] + lr.backward(Hin, w, mean_bar=mean_bar, var_bar=var_bar)
            start = len(rec["red"]) - 2
        else:
            fbar = None
            for i in range(len(rec["lay"]) - 1, -1, -1):
                lay, inp, z = rec["lay"][i]
                top = i == len(rec["lay"]) - 1
                xb = lay.backward(inp, z, f_bar=fbar, mean_bar=mean_bar if top else None, var_bar=var_bar if top else None)
                if i == 0:
                    Hbar[nH - 1] = Hbar[nH - 1] + xb
                else:
                    Dx = inp.shape[1] - 1
                    Hbar[nH - (i + 1)] = Hbar[nH - (i + 1)] + xb[:, :Dx]
                    fbar = xb[:, Dx:]
            start = len(rec["red"]) - 1
        for j in range(start, -1, -1):
            hb = Hbar[j + 1]
            if isinstance(hb, float):
                continue
            lr, Hin, w = rec["red"][j]
            Hbar[j] = Hbar[j] + lr.backward(Hin, w, f_bar=hb)

    def _chain_full_cov(self, X, S, zs, ws, fidelity_dim, project=False):
        """propagate(full_cov=True) (MF_DGP_EM.py:123-168): every layer is evaluated sample by sample on its own N
        inputs (conditional_SND, layers_red.py:63-80); covariances are [S, N, N, D]."""
        L = len(self.layers_red)
        parts = None
        for s in range(S):
            H = X
            Hs, Hm, Hv = [H], [], []
            for j, lr in enumerate(self.layers_red[L - fidelity_dim:]):
                H, m, v = lr.forward_full_cov(H, np.ascontiguousarray(ws[j][s]))
                Hs.append(H); Hm.append(m); Hv.append(v)
            if project:
                res = (Hs, Hm, Hv)
            else:
                Fs, Fm, Fv = [], [], []
                F = None
                for i in range(fidelity_dim + 1):
                    inp = Hs[-1] if i == 0 else np.concatenate([Hs[-(i + 1)], F], 1)
                    F, m, v = self.layers[i].forward_full_cov(inp, np.ascontiguousarray(zs[i][s]))
                    Fs.append(F); Fm.append(m); Fv.append(v)
                res = (Fs, Fm, Fv)
            if parts is None:
                parts = tuple([[t] for t in part] for part in res)
            else:
                for part, acc in zip(res, parts):
                    for t, a in zip(part, acc):
                        a.append(t)
        return tuple([as_tensor(np.stack(a)) for a in part] for part in parts)

    def propagate(self, X, full_cov=False, S=1, zs=None, ws=None, fidelity_dim=None, project=False):
        X = np.ascontiguousarray(X, dtype=np.float64)
        if full_cov:
            L = len(self.layers_red)
            fd = L if fidelity_dim is None else fidelity_dim
            N = X.shape[0]
            ws = ws if ws is not None else [self.rng.standard_normal((S, N, lr.num_outputs)) for lr in self.layers_red[L - fd:]]
            zs = zs if zs is not None else [self.rng.standard_normal((S, N, 1)) for _ in range(fd + 1)]
            self._sync_all()
            return self._chain_full_cov(X, S, [np.asarray(z) for z in zs], [np.asarray(w) for w in ws], fd, project)
        L = len(self.layers_red)
        fd = L if fidelity_dim is None else fidelity_dim
        N = X.shape[0]
        rng = self.rng
        ws = ws if ws is not None else [rng.standard_normal((S, N, lr.num_outputs)) for lr in self.layers_red[L - fd:]]
        zs = zs if zs is not None else [rng.standard_normal((S, N, 1)) for _ in range(fd + 1)]
        self._sync_all()        # Z_right keeps the value of the last bound evaluation, as in the reference
        a, b, c, _ = self._chain(X, S, zs, ws, fd, project)
        shape = lambda t: as_tensor(np.reshape(t, (S, N, -1)))
        return [shape(t) for t in a], [shape(t) for t in b], [shape(t) for t in c]

    def predict_f(self, X, full_cov=False, S=1, fidelity=None, fidelity_dim=None):
        _, Fm, Fv = self.propagate(X, full_cov=full_cov, S=S, fidelity_dim=fidelity_dim)
        f = -1 if fidelity is None else fidelity
        return Fm[f], Fv[f]

    def project(self, X, full_cov=False, S=1, fidelity=None, fidelity_dim=None):
        _, Hm, Hv = self.propagate(X, full_cov=full_cov, S=S, fidelity_dim=fidelity_dim, project=True)
        f = -1 if fidelity is None else fidelity
        return Hm[f], Hv[f]

    def predict_all_layers(self, Xnew, num_samples):
        return self.propagate(Xnew, full_cov=False, S=num_samples)

    def predict_y(self, Xnew, num_samples, full_cov=False):
        Fmean, Fvar = self.predict_f(Xnew, full_cov=full_cov, S=num_samples)
        return Fmean, as_tensor(Fvar + float(self.likelihood.likelihood.variance._value))

    def _likelihood_at_fidelity(self, Fmu, Fvar, Y, variance):
        """Gaussian log-likelihood term of one fidelity (MF_DGP_EM.py:205-216)."""
        return as_tensor(_ve(np.asarray(Fmu), np.asarray(Fvar), np.asarray(Y, dtype=np.float64), float(variance)))

    def E_log_p_Y(self, X_f, Y_f, fidelity=None, fidelity_dim=None, project=False):
        """Expected data log-likelihood per point, averaged over the Monte-Carlo samples (MF_DGP_EM.py:218-260): [N, D]."""
        if project:
            mean, var = self.project(X_f, S=self.num_samples, fidelity=fidelity, fidelity_dim=fidelity_dim)
            s2 = float(self.likelihood_projection.likelihood.variance._value)
        else:
            mean, var = self.predict_f(X_f, S=self.num_samples, fidelity=fidelity, fidelity_dim=fidelity_dim)
            s2 = self._noise(self.num_layers - 1 if fidelity is None or fidelity == -1 else fidelity)[0]
        return as_tensor(np.mean(_ve(np.asarray(mean), np.asarray(var), np.asarray(Y_f, dtype=np.float64), s2), 0))

    def predict_density(self, Xnew, Ynew, num_samples):
        """log of the mixture predictive density, averaged over the samples (MF_DGP_EM.py:318-322)."""
        Fmean, Fvar = self.predict_f(Xnew, full_cov=False, S=num_samples)
        v = np.asarray(Fvar) + float(self.likelihood.likelihood.variance._value)
        l = -0.5 * np.log(2 * np.pi * v) - 0.5 * (np.asarray(Ynew)[None] - np.asarray(Fmean)) ** 2 / v
        m = l.max(0)
        return as_tensor(m + np.log(np.exp(l - m).sum(0)) - np.log(num_samples))

    # ---- the bound and its gradient ----
    def _noise(self, f):
        """(value, Parameter) of the Gaussian noise of fidelity f: the likelihood's variance at the top, the White
        variance of the layer's kernel below (MF_DGP_EM.py:239-259)."""
        if f == self.num_layers - 1:
            p = self.likelihood.likelihood.variance
        else:
            p = self.layers[f].kern.kernels[-1].variance
        return float(p._value), p

    def _elbo(self, data, normals, want_grad):
        X, Y, X_red = data
        n, S = self.num_layers, self.num_samples
        X = [np.ascontiguousarray(x, dtype=np.float64) for x in X]
        if normals is None:
            normals = self.draw_normals(X, S)
        zr_recs = self.update_Z_right(normals, keep=True)
        grads = {}

        def add(p, g):
            grads[id(p)] = grads.get(id(p), 0.0) + g

        Lt = L_red = 0.0
        for f in range(n):
            if self._train_upto_fidelity != -1 and f > self._train_upto_fidelity:
                continue
            Nf = X[f].shape[0]
            _, Fm, Fv, rec = self._chain(X[f], S, normals["zs"][f], normals["ws"][f], f)
            s2, p_noise = self._noise(f)
            mean, var = Fm[f].reshape(S, Nf, -1), Fv[f].reshape(S, Nf, -1)
            Yf = np.asarray(Y[f], dtype=np.float64)
            Lt += _ve(mean, var, Yf, s2).sum() / S
            if want_grad:
                r2 = (Yf[None] - mean) ** 2 + var
                add(p_noise, (-0.5 / s2 + 0.5 * r2 / s2 ** 2).sum() / S)
                self._chain_backward(rec, ((Yf[None] - mean) / s2 / S).reshape(S * Nf, -1),
                                     np.full((S * Nf, mean.shape[2]), -0.5 / s2 / S))
            if f < n - 1:
                Xn, Yn = X[f + 1], np.asarray(X_red[f], dtype=np.float64)
                Nn = Xn.shape[0]
                _, Hm, Hv, rec = self._chain(Xn, S, None, normals["ws_proj"][f], f + 1, project=True)
                scale = Nn / Nf                                   # MF_DGP_EM.py:292-294, literally
                pv = self.likelihood_projection.likelihood.variance
                s2p = float(pv._value)
                mean, var = Hm[f].reshape(S, Nn, -1), Hv[f].reshape(S, Nn, -1)
                L_red += _ve(mean, var, Yn, s2p).sum() / S * scale
                if want_grad:
                    r2 = (Yn[None] - mean) ** 2 + var
                    add(pv, (-0.5 / s2p + 0.5 * r2 / s2p ** 2).sum() / S * scale)
                    self._chain_backward(rec, ((Yn[None] - mean) / s2p / S * scale).reshape(S * Nn, -1),
                                         np.full((S * Nn, mean.shape[2]), -0.5 / s2p / S * scale), project=True)
        KL = KL_red = 0.0
        if want_grad:
            # augmented layers from the top: their d/dZ_right seeds the Z_right chain, which reaches the layers below
            for i in range(n - 1, -1, -1):
                lay = self.layers[i]
                kl, g, gZ = lay.finish()
                KL += kl if (self._train_upto_fidelity == -1 or i <= self._train_upto_fidelity) else 0.0
                for k, v in g.items():
                    grads[k] = grads.get(k, 0.0) + v
                if lay.feature.augmented:
                    Dx = lay.input_dim - 1
                    zl = gZ[:, :Dx] + self._z_right_backward(zr_recs[i], np.ascontiguousarray(gZ[:, Dx:]))
                    add(lay.feature.Z_left, zl)
                else:
                    add(lay.feature.left(), gZ)
            for j, lr in enumerate(self.layers_red):
                kl, g, gZ = lr.finish()
                KL_red += kl
                for k, v in g.items():
                    grads[k] = grads.get(k, 0.0) + v
                add(lr.feature.left(), gZ)
        else:
            KL = sum(self.layers[i].finish()[0] for i in range(n)
                     if self._train_upto_fidelity == -1 or i <= self._train_upto_fidelity)
            KL_red = sum(lr.finish()[0] for lr in self.layers_red)
        self.L, self.KL, self.L_red, self.KL_red = Lt, KL, L_red, KL_red
        return Lt + L_red - KL - KL_red, grads

    def ELBO(self, data, normals=None):
        """MF_DGP_EM.py:262-301."""
        return self._elbo(data, normals, False)[0]

    ELBO_closure = ELBO

    def ELBO_and_grad(self, data, normals=None):
        """(ELBO, {id(Parameter): d ELBO / d constrained value}) - what `tape.gradient` yields at MF_DGP_EM.py:460-464."""
        return self._elbo(data, normals, True)

    def parameters(self):
        ps = []
        for lay in list(self.layers) + list(self.layers_red):
            ps += lay.parameters()
        return ps + [self.likelihood.likelihood.variance, self.likelihood_projection.likelihood.variance]

    def fix_inducing_point_locations(self):
        for layer in self.layers:
            set_trainable(layer.feature.left(), False)

    @classmethod
    def make_mf_dgp(cls, X, Z, W, add_linear=True, minibatch_size=None, seed=0):
        """MF_DGP_EM.py:324-374."""
        n_fidelities = len(Z)
        Din, Dout = X[0].shape[1], 1
        kernels = [RBF(active_dims=list(range(Din)), variance=1.0, lengthscales=[1.0] * Din)]
        for l in range(1, n_fidelities):
            Din = X[l].shape[1]
            D_range = list(range(Din + Dout))
            k_corr = RBF(active_dims=D_range[:Din], variance=1.0)
            k_prev = RBF(active_dims=D_range[Din:], variance=1.0)
            k_in = RBF(active_dims=D_range[:Din], variance=1.0)
            if add_linear:
                k_l = k_corr * (k_prev + LinearKernel(active_dims=D_range[Din:], variance=1.0)) + k_in
            else:
                k_l = k_corr * k_prev + k_in
            kernels.append(k_l)
        kernels_red = [RBF(variance=1.0, lengthscales=[1.0] * X[-(l + 1)].shape[1]) for l in range(n_fidelities - 1)]
        for i in range(n_fidelities - 1):
            kernels[i] = kernels[i] + White(variance=1e-6)
        layers, layers_red = init_layers_mf(X, Z, W, kernels, kernels_red, num_outputs=Dout)
        model = cls(Gaussian(), layers, layers_red, num_samples=100, minibatch_size=minibatch_size, seed=seed)
        model._init_q_sqrt_to_prior()
        return model

    def _init_q_sqrt_to_prior(self):
        """q_sqrt = chol(K(Z) + jitter I) with Z_right sampled through the layers below (layers_red.py:208-216)."""
        for lay in self.layers_red + [self.layers[0]]:
            self._prior(lay)
        for i in range(1, self.num_layers):
            for l in self.layers_red:
                l.sync()
            for l in self.layers[:i]:
                l.sync()
            zb = self._draw_zright(100)[i]
            self.layers[i].feature.Z_right = self._z_right_forward(i, zb)[0]
            self._prior(self.layers[i])

    @staticmethod
    def _prior(lay):
        Z = np.asarray(lay.feature.Z.numpy())
        K = _kernel_matrix_host(lay.kind, [float(v[0]) if v.size == 1 else v for v in lay._kvals()],
                                None if lay._white is None else float(lay._white._value), Z)
        Lu = np.linalg.cholesky(K + JITTER * np.eye(Z.shape[0]))
        lay.q_sqrt._value = np.tile(Lu[None], [lay.num_outputs, 1, 1])


# ------------------------------------------------------------------------------------------ host optimiser
def _to_unconstrained(p):
    x = p._value
    if p.transform == "softplus":
        return np.log(np.expm1(x)) if np.all(x < 30) else x + np.log1p(-np.exp(-x))
    if p.transform == "softplus_shift":
        y = x - 1e-6
        return np.log(np.expm1(y))
    return x.copy()


def _from_unconstrained(p, u):
    if p.transform == "softplus":
        return np.logaddexp(0.0, u)
    if p.transform == "softplus_shift":
        return np.logaddexp(0.0, u) + 1e-6
    if p.transform == "tril":
        return np.tril(u)
    return u


def _chain_to_unconstrained(p, g):
    """d/du from d/dx: softplus' = 1 - exp(-x) (optimisers act on the unconstrained variables)."""
    if p.transform == "softplus":
        return g * (-np.expm1(-p._value))
    if p.transform == "softplus_shift":
        return g * (-np.expm1(-(p._value - 1e-6)))
    if p.transform == "tril":
        return np.tril(g)
    return g


class _Adam:
    """tf.optimizers.Adam (TF 2.x Keras), one state per parameter, on the unconstrained values."""

    def __init__(self, lr, beta_1, beta_2, epsilon):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.t = 0
        self.state = {}

    def apply(self, params, grads_elbo):
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for p in params:
            g = grads_elbo.get(id(p))
            if g is None or not p.trainable:
                continue
            g = -_chain_to_unconstrained(p, np.asarray(g, dtype=np.float64).reshape(p._value.shape))   # loss = -ELBO
            m, v = self.state.setdefault(id(p), (np.zeros_like(p._value), np.zeros_like(p._value)))
            m[...] = self.b1 * m + (1 - self.b1) * g
            v[...] = self.b2 * v + (1 - self.b2) * g * g
            u = _to_unconstrained(p) - lr_t * m / (np.sqrt(v) + self.eps)
            p._value = np.asarray(_from_unconstrained(p, u), dtype=np.float64)


class MultiFidelityDeepGP_EM:
    """MF_DGP_EM.py:383-596: inducing points fixed for the first part of the optimisation, then freed."""

    def __init__(self, X, Y, X_red, Z=None, W=None, n_iter=5000, fix_inducing=True, minibatch_size=None, seed=0):
        self.name = "mf_dgp_EM"
        self._Y, self._X, self._X_red = Y, X, X_red
        self.minibatch_size = minibatch_size
        self.Z = self._make_inducing_points(X, Y) if Z is None else Z
        if W is None:
            self.W = [X[-1].copy()]
            for i in range(1, len(X) - 1):
                self.W.append(X[-(1 + i)])
        else:
            self.W = W
        self.model = DGP_Base.make_mf_dgp(X, self.Z, self.W, minibatch_size=minibatch_size, seed=seed)
        self.n_fidelities = len(X)
        self.n_iter = n_iter
        self.fix_inducing = fix_inducing

    def predict(self, X_test, full_cov=False):
        y_m, y_v = self.model.predict_y(X_test, 250, full_cov=full_cov)
        y_m, y_v = np.asarray(y_m), np.asarray(y_v)
        mean = np.mean(y_m, axis=0).flatten()
        if full_cov:
            # MF_DGP_EM.py:419-425 adds np.mean(y_v, 0).flatten() (N*N values) to np.var(y_m, 0).flatten() (N values):
            # that only broadcasts for N = 1.  The same two terms for any N (law of total covariance): the mean of the
            # per-sample [N, N] covariances + the covariance of the per-sample means -> ([N, 1], [N, N]); for N = 1 it is
            # the reference's value
            dm = y_m[:, :, 0] - mean[None]
            return mean[:, None], np.mean(y_v[..., 0], axis=0) + dm.T @ dm / y_m.shape[0]
        var = np.mean(y_v, axis=0).flatten() + np.var(y_m, axis=0).flatten()
        return mean[:, None], var[:, None]

    def objective(self):
        return self.model.ELBO((self._X, self._Y, self._X_red))

    def _data(self):
        return (self._X, self._Y, self._X_red)

    def _run(self, optimizer, iterations, messages, natgrad=None):
        params = self.model.parameters()
        for it in range(iterations):
            elbo, grads = self.model.ELBO_and_grad(self._data())
            optimizer.apply(params, grads)
            if natgrad is not None:
                gamma, layers = natgrad
                self.model.ELBO_and_grad(self._data())             # `optimizer_nat.minimize` evaluates the loss again
                for lay in layers:
                    lay.natgrad_step(gamma)
            if messages and it % messages == 0:
                print(f"ELBO: {elbo}")

    def _initialise(self, q_scale, red_scale):
        m = self.model
        for i, layer in enumerate(m.layers[:-1]):
            layer.q_mu.assign(self._Y[i])
            set_trainable(layer.q_mu, False)
            layer.q_sqrt.assign(layer.q_sqrt.numpy() * q_scale * self._Y[i].var())
            set_trainable(layer.q_sqrt, False)
        m.layers[-1].q_sqrt.assign(m.layers[-1].q_sqrt.numpy() * self._Y[-1].var() * q_scale)
        set_trainable(m.layers[-1].q_sqrt, False)
        set_trainable(m.layers[-1].q_mu, False)
        m.layers[-1].q_mu.assign(self._Y[-1])
        for layer in m.layers_red:
            layer.q_sqrt.assign(layer.q_sqrt.numpy() * red_scale)
            set_trainable(layer.q_sqrt, False)
        for i, layer in enumerate(m.layers_red):
            layer.q_mu.assign(self._X_red[-(i + 1)])
            set_trainable(layer.q_mu, False)

    def optimize_adam(self, lr=0.01, iterations1=2000, iterations2=5000, iterations3=7500, beta_1=0.9, beta_2=0.999,
                      epsilon=1e-07, messages=500):
        """MF_DGP_EM.py:429-499."""
        m = self.model
        optimizer = _Adam(lr, beta_1, beta_2, epsilon)
        self._initialise(1e-2, 1e-2)
        m.likelihood.likelihood.variance.assign(self._Y[-1].var() * 1e-2)
        set_trainable(m.likelihood.likelihood.variance, False)
        set_trainable(m.layers[0].feature.left(), False)
        for layer in m.layers[1:]:
            set_trainable(layer.feature.Z_left, False)
        self._run(optimizer, iterations1, messages)
        set_trainable(m.layers[0].feature.left(), True)
        for layer in m.layers[1:]:
            set_trainable(layer.feature.Z_left, True)
        self._run(optimizer, iterations2, messages)
        m.update_Z_right()
        set_trainable(m.likelihood.likelihood.variance, True)
        for layer in m.layers:
            set_trainable(layer.q_mu, True)
            set_trainable(layer.q_sqrt, True)
        self._run(optimizer, iterations3, messages)
        m.update_Z_right()

    def optimize_nat_adam(self, lr_adam=0.01, lr_gamma=0.01, iterations1=2000, iterations2=5000, iterations3=7500,
                          beta_1=0.9, beta_2=0.999, epsilon=1e-07, messages=500):
        """MF_DGP_EM.py:501-578."""
        m = self.model
        optimizer = _Adam(lr_adam, beta_1, beta_2, epsilon)
        self._initialise(1e-3, 1e-5)
        m.likelihood_projection.likelihood.variance.assign(self._X_red[-1].var() * 1e-3)
        set_trainable(m.likelihood_projection.likelihood.variance, False)
        m.likelihood.likelihood.variance.assign(self._Y[-1].var() * 1e-3)
        set_trainable(m.likelihood.likelihood.variance, False)
        set_trainable(m.layers[0].feature.left(), False)
        for layer in m.layers[1:]:
            set_trainable(layer.feature.Z_left, False)
        self._run(optimizer, iterations1, messages)
        set_trainable(m.layers[0].feature.left(), True)
        for layer in m.layers[1:]:
            set_trainable(layer.feature.Z_left, True)
        self._run(optimizer, iterations2, messages)
        m.update_Z_right()
        set_trainable(m.likelihood.likelihood.variance, False)
        self._run(optimizer, iterations3, messages, natgrad=(lr_gamma, list(m.layers) + list(m.layers_red)))
        m.update_Z_right()

    @staticmethod
    def _make_inducing_points(X, Y):
        """MF_DGP_EM.py:579-596: the training inputs of every fidelity."""
        return [x.copy() for x in X]
