"""Infill (acquisition) criteria over a DGP surrogate, TensorFlow-free.

Host-side mirror of the reference's ``dgp_dace/Infill_criteria.py`` for ``model.name == 'dgp'``: same class names,
constructor arguments, ``run`` / ``optimize`` signatures, sign conventions (``run`` returns MINUS the criterion) and
search-space transform (``x = lw + (up-lw)/(1+exp(u))``, Infill_criteria.py:63).  What changes is the execution:

* ``run`` evaluates a whole population of candidates in one call (P = num_samples x population points through
  ``predict_f`` / ``predict_y`` / ``propagate`` on the device) -- the reference does the same through a tf.function;
* the Adam branch (Infill_criteria.py:69-85) gets its gradient from ``DGP.propagate_vjp`` (the C-ABI's
  ``dgp_propagate_vjp``) and the closed-form derivatives of the criterion with respect to the predictive moments,
  where the reference runs ``tf.GradientTape``;
* differential evolution (``tfp.optimizer.differential_evolution_minimize``, Infill_criteria.py:65-67) is restated in
  NumPy with TFP's defaults (rand/1/bin, differential_weight 0.5, crossover_prob 0.9).

The exact-GP branch (``model.name == 'gpr'``, ``dgp_dace.models.gpr.GPR``) is evaluated from ``predict_y`` as in the
reference; its Adam branch takes the input gradient from ``GPR.predict_vjp``.
"""
import numpy as np
from scipy.special import ndtr

from .gpflow_compat import as_tensor

_SQRT_2PI = np.sqrt(2.0 * np.pi)


def _np(x):
    return np.asarray(x.numpy() if hasattr(x, "numpy") else x, dtype=np.float64)


def _moments(Fm, Fv):
    """Moment-matched mixture over the S samples (Infill_criteria.py:40-41): mean, variance, each [N, D]."""
    mean = Fm.mean(0)
    return mean, (Fv + Fm ** 2).mean(0) - mean ** 2


def _ei(y_min, mean, var):
    """(y_min - mu) Phi(z) + var * N(y_min; mu, sigma)  (Infill_criteria.py:43-47), and its partials in mu, var."""
    sd = np.sqrt(var)
    z = (y_min - mean) / sd
    cdf, pdf = ndtr(z), np.exp(-0.5 * z * z) / _SQRT_2PI
    return (y_min - mean) * cdf + sd * pdf, -cdf, pdf / (2.0 * sd)


def _moment_cotangents(Fm, mean, d_mean, d_var):
    """Cotangents of the per-sample (Fmean_s, Fvar_s) given those of the moment-matched (mean, var)."""
    S = Fm.shape[0]
    mean_bar = d_mean[None] / S + d_var[None] * 2.0 * (Fm - mean[None]) / S
    var_bar = np.broadcast_to(d_var[None] / S, Fm.shape).copy()
    return mean_bar, var_bar


class Infill_criteria(object):
    def __init__(self):
        self.name = 'Infill criteria'

    def run(self, model, x):
        raise NotImplementedError("method not implemented")

    # ---- shared machinery -------------------------------------------------------------------------------
    def _value_and_grad(self, model, x, **kw):
        """(MINUS criterion [N,1], d sum(MINUS criterion) / dx [N,d]); rows of x are independent candidates."""
        raise NotImplementedError("method not implemented")

    @staticmethod
    def _check(model, gradient=False):
        name = getattr(model, "name", None)
        if name not in ('dgp', 'gpr'):
            raise NotImplementedError("model.name must be 'dgp' or 'gpr'")
        if gradient and name == 'gpr' and not hasattr(model, "predict_vjp"):
            raise NotImplementedError("the Adam branch needs d prediction / dx (dgp_dace.models.gpr.GPR.predict_vjp)")

    @staticmethod
    def _gpr_moments(model, x):
        mean, var = model.predict_y(x)                    # Infill_criteria.py:28-29
        return _np(mean), _np(var)

    def loss(self, model, x, *args, **kw):          # the reference wraps run in a tf.function; nothing to trace here
        return self.run(model, x, *args, **kw)

    def optimize(self, model, bounds, popsize_DE=300, popstd_DE=1.5, iterations_DE=400, init_adam=None,
                 iterations_adam=1000, method='DE', seed=None, **run_kw):
        """Minimise MINUS the criterion over the box `bounds` (Infill_criteria.py:61-87): differential evolution on the
        unconstrained variable u (x = lw + (up-lw)/(1+exp(u))), then optionally Adam(lr 0.01) from its optimum."""
        self._check(model)
        lw, up = (np.asarray(b, dtype=np.float64).reshape(-1) for b in bounds)
        to_x = lambda u: lw + (up - lw) / (1.0 + np.exp(u))
        rng = np.random.default_rng(seed)
        self.x_opt = getattr(self, "x_opt", None)
        if method in ('DE', 'DE+Adam'):
            f = lambda U: _np(self.run(model, to_x(U), **run_kw)).reshape(U.shape[0], -1).sum(1)
            u_best = _differential_evolution(f, np.zeros(self.d), popstd_DE, popsize_DE, iterations_DE, rng)
            self.x_opt = to_x(u_best).reshape(self.d, 1)
            self.IC_optimized = self.run(model, self.x_opt.reshape(1, self.d), **run_kw)
        if method in ('Adam', 'DE+Adam'):
            self._check(model, gradient=True)
            if init_adam is None:
                init_adam = np.zeros(self.d) if self.x_opt is None else self.x_opt
            x0 = np.asarray(init_adam, dtype=np.float64).reshape(-1)
            u = np.log((up - x0 + 1e-3) / (x0 - lw + 1e-3))
            m, v = np.zeros_like(u), np.zeros_like(u)
            objective = None
            for step in range(1, iterations_adam + 1):
                e = np.exp(u)
                x = lw + (up - lw) / (1.0 + e)
                val, gx = self._value_and_grad(model, x.reshape(1, self.d), **run_kw)
                objective = val
                g = gx.reshape(-1) * (-(up - lw) * e / (1.0 + e) ** 2)          # chain rule through the transform
                m = 0.9 * m + 0.1 * g
                v = 0.999 * v + 0.001 * g * g
                lr_t = 0.01 * np.sqrt(1.0 - 0.999 ** step) / (1.0 - 0.9 ** step)
                u = u - lr_t * m / (np.sqrt(v) + 1e-7)                           # tf.optimizers.Adam, epsilon 1e-7
            self.x_opt = to_x(u).reshape(self.d, 1)
            self.IC_optimized = objective
        return self.x_opt


def _differential_evolution(f, x0, pop_std, pop_size, iterations, rng, weight=0.5, crossover=0.9):
    """rand/1/bin differential evolution over R^d, whole population evaluated per call of `f` ([pop, d] -> [pop])."""
    d = x0.size
    pop = x0[None] + pop_std * rng.standard_normal((pop_size, d))
    pop[0] = x0
    val = f(pop)
    idx = np.arange(pop_size)
    for _ in range(iterations):
        # three mutually distinct partners, none equal to the target
        r = np.argsort(rng.random((pop_size, pop_size - 1)), axis=1)[:, :3]
        r = r + (r >= idx[:, None])
        mutant = pop[r[:, 0]] + weight * (pop[r[:, 1]] - pop[r[:, 2]])
        cross = rng.random((pop_size, d)) < crossover
        cross[idx, rng.integers(0, d, pop_size)] = True
        trial = np.where(cross, mutant, pop)
        tv = f(trial)
        better = tv <= val
        pop[better], val[better] = trial[better], tv[better]
    return pop[np.argmin(val)]


class EI(Infill_criteria):
    """Expected improvement below y_min (Infill_criteria.py:20-52)."""

    def __init__(self, y_min, d):
        self.name = 'Expected Improvement'
        self.y_min = y_min
        self.d = d
        self.IC_optimized = None
        self.x_opt = None

    def run(self, model, x, analytic=True, num_samples=1000):
        self._check(model)
        x = _np(x)
        y_min = np.asarray(self.y_min, dtype=np.float64)
        if model.name == 'gpr':
            return as_tensor(-_ei(y_min, *self._gpr_moments(model, x))[0])
        if analytic:
            Fm, Fv = (_np(a) for a in model.predict_f(x, S=num_samples))
            ei, _, _ = _ei(y_min, *_moments(Fm, Fv))
        else:
            F, _, _ = model.propagate(x, S=num_samples)
            FL = _np(F[-1])
            ei = np.where(FL - y_min < 0, y_min - FL, 0.0).mean(0)
        return as_tensor(-ei)

    def _value_and_grad(self, model, x, analytic=True, num_samples=1000):
        y_min = np.asarray(self.y_min, dtype=np.float64)
        if model.name == 'gpr':
            ei, d_mean, d_var = _ei(y_min, *self._gpr_moments(model, x))
            return -ei, _np(model.predict_vjp(x, -d_mean, -d_var))
        if analytic:
            Fm, Fv = (_np(a) for a in model.predict_f(x, S=num_samples))
            mean, var = _moments(Fm, Fv)
            ei, d_mean, d_var = _ei(y_min, mean, var)
            mean_bar, var_bar = _moment_cotangents(Fm, mean, -d_mean, -d_var)
            return -ei, _np(model.propagate_vjp(x, S=num_samples, mean_bar=mean_bar, var_bar=var_bar))
        F, _, _ = model.propagate(x, S=num_samples)
        FL = _np(F[-1])
        below = FL - y_min < 0
        f_bar = np.where(below, 1.0, 0.0) / FL.shape[0]                 # d(-EI)/dF_s
        return -np.where(below, y_min - FL, 0.0).mean(0), _np(model.propagate_vjp(x, S=num_samples, f_bar=f_bar))


class WB2(Infill_criteria):
    """Watson-Barnes criterion EI - mean on predict_y with 500 samples (Infill_criteria.py:104-131)."""

    num_samples = 500

    def __init__(self, y_min, d):
        self.name = 'WB2 criterion'
        self.y_min = y_min
        self.d = d
        self.IC_optimized = None
        self.x_opt = None

    def _scale(self, x):
        return 1.0

    def run(self, model, x):
        self._check(model)
        x = _np(x)
        if model.name == 'gpr':
            mean, var = self._gpr_moments(model, x)
        else:
            Fm, Fv = (_np(a) for a in model.predict_y(x, num_samples=self.num_samples))
            mean, var = _moments(Fm, Fv)
        ei, _, _ = _ei(np.asarray(self.y_min, dtype=np.float64), mean, var)
        return as_tensor(-(self._scale(x) * ei - mean))

    def _value_and_grad(self, model, x):
        gpr = model.name == 'gpr'
        if gpr:
            mean, var = self._gpr_moments(model, x)
        else:
            Fm, Fv = (_np(a) for a in model.predict_y(x, num_samples=self.num_samples))
            mean, var = _moments(Fm, Fv)
        ei, d_mean, d_var = _ei(np.asarray(self.y_min, dtype=np.float64), mean, var)
        s = self._scale(x)
        s_col = s if np.ndim(s) == 0 else s.sum(1, keepdims=True)      # WB2S broadcasts [N,d] * [N,1]: sum over d
        k = 1.0 if np.ndim(s) == 0 else float(x.shape[1])              # ... and "- mean" is then counted d times
        if gpr:
            gx = _np(model.predict_vjp(x, -(s_col * d_mean) + k, -(s_col * d_var)))
        else:
            mean_bar, var_bar = _moment_cotangents(Fm, mean, -(s_col * d_mean) + k, -(s_col * d_var))
            gx = _np(model.propagate_vjp(x, S=self.num_samples, mean_bar=mean_bar, var_bar=var_bar))
        if np.ndim(s) != 0:
            gx = gx - self._dscale(x) * ei                              # explicit dependence of the scale on x
        return -(s * ei - mean), gx


class WB2S(WB2):
    """WB2 with the EI term scaled by sigmoid(x) (Infill_criteria.py:175-199; element-wise in x as written there)."""

    def __init__(self, y_min, d):
        super().__init__(y_min, d)
        self.name = 'WB2S criterion'

    def _scale(self, x):
        return 1.0 / (1.0 + 1.0 / np.exp(x))

    def _dscale(self, x):
        s = self._scale(x)
        return s * (1.0 - s)


def _ev(zero_c, mean, var):
    """Expected violation E[max(g - zero_c, 0)] for g ~ N(mean, var) as Infill_criteria.py:243-257 writes it
    ((mu - c) Phi(z) + var * N(-c; -mu, sigma), z = (mu - c)/sigma), with its partials in mu and var."""
    sd = np.sqrt(var)
    z = (mean - zero_c) / sd
    cdf, pdf = ndtr(z), np.exp(-0.5 * z * z) / _SQRT_2PI
    return (mean - zero_c) * cdf + sd * pdf, cdf, pdf / (2.0 * sd)


class EV_one_constraint(Infill_criteria):
    """Expected violation of one constraint model (Infill_criteria.py:234-262); `run` returns +EV."""

    num_samples_analytic = 500

    def __init__(self, zero_c, d):
        self.name = 'Expected Violation'
        self.zero_c = zero_c
        self.d = d
        self.IC_optimized = None
        self.x_opt = None

    def _moments(self, model, x):
        if model.name == 'gpr':
            mean, var = self._gpr_moments(model, x)
            return mean, var, None
        Fm, Fv = (_np(a) for a in model.predict_y(x, num_samples=self.num_samples_analytic))
        mean, var = _moments(Fm, Fv)
        return mean, var, Fm

    def run(self, model, x, analytic=True, num_samples=100):
        self._check(model)
        x = _np(x)
        c = np.asarray(self.zero_c, dtype=np.float64)
        if analytic:
            mean, var, _ = self._moments(model, x)
            return as_tensor(_ev(c, mean, var)[0])
        F, _, _ = model.propagate(x, S=num_samples)
        FL = _np(F[-1])
        return as_tensor(np.where(FL - c < 0, 0.0, FL - c).mean(0))

    def _value_and_grad(self, model, x, analytic=True, num_samples=100):
        """(+EV [N,1], d sum(EV)/dx)"""
        self._check(model, gradient=True)
        c = np.asarray(self.zero_c, dtype=np.float64)
        if analytic:
            mean, var, Fm = self._moments(model, x)
            ev, d_mean, d_var = _ev(c, mean, var)
            if Fm is None:
                return ev, _np(model.predict_vjp(x, d_mean, d_var))
            mean_bar, var_bar = _moment_cotangents(Fm, mean, d_mean, d_var)
            return ev, _np(model.propagate_vjp(x, S=self.num_samples_analytic, mean_bar=mean_bar, var_bar=var_bar))
        F, _, _ = model.propagate(x, S=num_samples)
        FL = _np(F[-1])
        above = FL - c >= 0
        return np.where(above, FL - c, 0.0).mean(0), _np(model.propagate_vjp(x, S=num_samples, f_bar=np.where(above, 1.0, 0.0) / FL.shape[0]))


class EV(Infill_criteria):
    """Expected violation over a list of constraint models and its combination with an unconstrained criterion
    (Infill_criteria.py:264-316): a candidate whose largest expected violation exceeds `threshold` scores
    sum(EV) + 10000, any other scores the criterion IC (minus EI, ...) of the objective model."""

    def __init__(self, zero_c, d):
        self.name = 'Expected Violation'
        self.zero_c = zero_c
        self.d = d
        self.IC_optimized = None
        self.x_opt = None

    def run(self, model_C, x, analytic=True, num_samples=100):
        cols = [_np(EV_one_constraint(self.zero_c[i], self.d).run(model_C[i], x, analytic=analytic, num_samples=num_samples))
                for i in range(len(model_C))]
        return as_tensor(np.concatenate(cols, 1))

    def run_with_IC(self, IC, model_Y, model_C, x, threshold=0.1, analytic=True, num_samples=100):
        ev = _np(self.run(model_C, x, analytic=analytic, num_samples=num_samples))
        ic = _np(IC.run(model_Y, x)).reshape(ev.shape[0], -1)[:, :1]
        bad = ev.max(1, keepdims=True) > threshold
        return as_tensor(np.where(bad, ev.sum(1, keepdims=True) + 10000.0, ic))

    def _value_and_grad_with_IC(self, IC, model_Y, model_C, x, threshold, analytic, num_samples):
        parts = [EV_one_constraint(self.zero_c[i], self.d)._value_and_grad(model_C[i], x, analytic=analytic, num_samples=num_samples)
                 for i in range(len(model_C))]
        ev = np.concatenate([v for v, _ in parts], 1)
        g_ev = sum(g for _, g in parts)
        ic, g_ic = IC._value_and_grad(model_Y, x)
        bad = ev.max(1, keepdims=True) > threshold
        return np.where(bad, ev.sum(1, keepdims=True) + 10000.0, _np(ic).reshape(ev.shape[0], -1)[:, :1]), np.where(bad, g_ev, g_ic)

    def optimize_with_IC(self, IC, model_Y, model_C, bounds, threshold=0.1, analytic=True, num_samples=100, popsize_DE=300,
                         popstd_DE=1.5, iterations_DE=400, init_adam=None, iterations_adam=1000, method='DE', seed=None):
        """Infill_criteria.py:287-316 (the reference hard-codes threshold 0.1 inside its objective; the argument is
        honoured here)."""
        lw, up = (np.asarray(b, dtype=np.float64).reshape(-1) for b in bounds)
        to_x = lambda u: lw + (up - lw) / (1.0 + np.exp(u))
        rng = np.random.default_rng(seed)
        if method in ('DE', 'DE+Adam'):
            f = lambda U: _np(self.run_with_IC(IC, model_Y, model_C, to_x(U), threshold, analytic, num_samples)).reshape(U.shape[0], -1).sum(1)
            u_best = _differential_evolution(f, np.zeros(self.d), popstd_DE, popsize_DE, iterations_DE, rng)
            self.x_opt = to_x(u_best).reshape(self.d, 1)
            self.IC_optimized = self.run_with_IC(IC, model_Y, model_C, self.x_opt.reshape(1, self.d), threshold, analytic, num_samples)
        if method in ('Adam', 'DE+Adam'):
            if init_adam is None:
                init_adam = np.zeros(self.d) if self.x_opt is None else self.x_opt
            x0 = np.asarray(init_adam, dtype=np.float64).reshape(-1)
            u = np.log((up - x0 + 1e-3) / (x0 - lw + 1e-3))
            m, v = np.zeros_like(u), np.zeros_like(u)
            objective = None
            for step in range(1, iterations_adam + 1):
                e = np.exp(u)
                x = lw + (up - lw) / (1.0 + e)
                objective, gx = self._value_and_grad_with_IC(IC, model_Y, model_C, x.reshape(1, self.d), threshold, analytic, num_samples)
                g = gx.reshape(-1) * (-(up - lw) * e / (1.0 + e) ** 2)
                m = 0.9 * m + 0.1 * g
                v = 0.999 * v + 0.001 * g * g
                u = u - 0.01 * np.sqrt(1.0 - 0.999 ** step) / (1.0 - 0.9 ** step) * m / (np.sqrt(v) + 1e-7)
            self.x_opt = to_x(u).reshape(self.d, 1)
            self.IC_optimized = objective
        return self.x_opt

