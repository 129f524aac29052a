"""Duck-typed stand-ins for the handful of GPflow objects the DGP constructor is handed.

GPflow/TensorFlow are not a dependency of this engine.  The reference's callers build kernels,
likelihood and mean functions with GPflow (nb_DGP_regression cell 17: ``RBF(lengthscales=[1]*d,
variance=1.0)``; SO_BO.py:240 ``gpflow.kernels.SquaredExponential``; ``gpflow.likelihoods.Gaussian()``)
and touch parameters through ``.numpy()`` / ``.assign()`` / ``gpflow.set_trainable`` (dgp.py:268-269,
316-322).  These classes offer exactly that surface; real GPflow objects are accepted too (they are
read by attribute name, see ``kernel_from_any`` / ``likelihood_variance_from_any``).
"""
from __future__ import annotations

import numpy as np


class TensorLike(np.ndarray):
    """ndarray that also answers ``.numpy()`` like the tf.Tensor the reference returns."""

    def numpy(self):
        return np.asarray(self)


def as_tensor(a):
    return np.asarray(a, dtype=np.float64).view(TensorLike)


class Parameter:
    """Value holder with the gpflow.Parameter surface the reference's code paths use."""

    def __init__(self, value, name=None, transform="identity", trainable=True):
        self._value = np.array(value, dtype=np.float64)
        self.name = name
        self.transform = transform          # "identity" | "softplus" | "softplus_shift" | "tril"
        self.trainable = bool(trainable)
        self._owner = None                  # set by DGP: object with _before_read() / _after_write()

    # -- gpflow surface --
    def numpy(self):
        if self._owner is not None:
            self._owner._before_read()
        return self._value.copy()

    def assign(self, value):
        value = np.asarray(value.numpy() if hasattr(value, "numpy") else value, dtype=np.float64)
        if self._owner is not None:
            self._owner._before_read()
        if value.shape != self._value.shape:
            value = np.broadcast_to(value, self._value.shape)
        self._value = np.array(value, dtype=np.float64)
        if self.transform == "tril":
            self._value = np.tril(self._value)
        if self._owner is not None:
            self._owner._after_write()
        return self

    @property
    def shape(self):
        return self._value.shape

    def __array__(self, dtype=None, copy=None):
        v = self.numpy()
        return v.astype(dtype) if dtype is not None else v

    def __mul__(self, o):
        return self.numpy() * np.asarray(o)

    __rmul__ = __mul__

    def __add__(self, o):
        return self.numpy() + np.asarray(o)

    __radd__ = __add__

    def __sub__(self, o):
        return self.numpy() - np.asarray(o)

    def __truediv__(self, o):
        return self.numpy() / np.asarray(o)

    def __repr__(self):
        return f"Parameter(name={self.name!r}, shape={self._value.shape}, transform={self.transform}, trainable={self.trainable})"


def set_trainable(obj, flag):
    """gpflow.set_trainable for a Parameter or any object holding Parameters."""
    if isinstance(obj, Parameter):
        obj.trainable = bool(flag)
        return
    for v in vars(obj).values():
        if isinstance(v, Parameter):
            v.trainable = bool(flag)


def _val(p):
    return np.asarray(p.numpy() if hasattr(p, "numpy") else p, dtype=np.float64)


class _KernelOps:
    """`k1 + k2` / `k1 * k2` as gpflow builds Sum / Product kernels (MF_DGP_EM.py:350-352,367)."""

    def __add__(self, other):
        return Sum([self, other])

    def __mul__(self, other):
        return Product([self, other])


class _Combination(_KernelOps):
    def __init__(self, kernels):
        self.kernels = []
        for k in kernels:                      # gpflow flattens nested combinations of the same type
            self.kernels.extend(k.kernels if type(k) is type(self) else [k])


class Sum(_Combination):
    kind = "sum"


class Product(_Combination):
    kind = "product"


class SquaredExponential(_KernelOps):
    """gpflow.kernels.SquaredExponential: K = variance * exp(-0.5 * |x/l - x'/l|^2), K_diag = variance."""

    kind = "rbf"

    def __init__(self, variance=1.0, lengthscales=1.0, active_dims=None, **_):
        self.variance = Parameter(np.asarray(variance, dtype=np.float64).reshape(()), "variance", "softplus")
        self.lengthscales = Parameter(np.atleast_1d(np.asarray(lengthscales, dtype=np.float64)), "lengthscales",
                                      "softplus")
        self.active_dims = None if active_dims is None else list(active_dims)


class LinearKernel(_KernelOps):
    """gpflow.kernels.Linear: K = variance * x x'^T on its active dimension(s)."""

    kind = "linear"

    def __init__(self, variance=1.0, active_dims=None, **_):
        self.variance = Parameter(np.asarray(variance, dtype=np.float64).reshape(()), "variance", "softplus")
        self.active_dims = None if active_dims is None else list(active_dims)


class White(_KernelOps):
    """gpflow.kernels.White: variance * I on K(X) and K_diag, zero cross-covariance."""

    kind = "white"

    def __init__(self, variance=1.0, active_dims=None, **_):
        self.variance = Parameter(np.asarray(variance, dtype=np.float64).reshape(()), "variance", "softplus")
        self.active_dims = None if active_dims is None else list(active_dims)


RBF = SquaredExponential


class Matern32(SquaredExponential):
    """gpflow.kernels.Matern32: variance (1 + sqrt3 r) exp(-sqrt3 r), r = sqrt(max(|x/l - x'/l|^2, 1e-36))."""

    kind = "matern32"


class Matern52(SquaredExponential):
    """gpflow.kernels.Matern52: variance (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)."""

    kind = "matern52"


KERNEL_KINDS = {"rbf": 0, "matern32": 1, "matern52": 2}          # dgp_kernel_kind of include/dgp_abi.h
_KERNEL_CLASSES = {"SquaredExponential": SquaredExponential, "RBF": SquaredExponential, "Matern32": Matern32,
                   "Matern52": Matern52}


def split_white(kern):
    """(stationary kernel, White kernel or None) of `k` or `k + White(...)` - the two forms a DGP layer kernel may take."""
    if isinstance(kern, Sum):
        base = [k for k in kern.kernels if not isinstance(k, White)]
        white = [k for k in kern.kernels if isinstance(k, White)]
        if len(base) == 1 and len(white) == 1 and isinstance(base[0], SquaredExponential):
            return base[0], white[0]
        raise NotImplementedError("layer kernel: a stationary kernel, optionally plus one White kernel")
    return kern, None


def kernel_matrix_host(kern, Z):
    """K(Z, Z) in NumPy for the constructor-time prior initialisation (layers.py:220-223 does this on the host too)."""
    kern, white = split_white(kern)
    if white is not None:
        return kernel_matrix_host(kern, Z) + float(white.variance._value) * np.eye(Z.shape[0])
    Zs = Z / kern.lengthscales._value
    sq = np.sum(Zs * Zs, -1)
    r2 = -2.0 * Zs @ Zs.T + sq[:, None] + sq[None, :]
    v = kern.variance._value
    if kern.kind == "rbf":
        return v * np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, 1e-36))
    if kern.kind == "matern32":
        a = np.sqrt(3.0) * r
        return v * (1.0 + a) * np.exp(-a)
    a = np.sqrt(5.0) * r
    return v * (1.0 + a + 5.0 / 3.0 * r * r) * np.exp(-a)


class Gaussian:
    """gpflow.likelihoods.Gaussian: variance with the Softplus + Shift(1e-6) transform."""

    def __init__(self, variance=1.0, **_):
        self.variance = Parameter(np.asarray(variance, dtype=np.float64).reshape(()), "variance", "softplus_shift")


class Zero:
    kind = "zero"


class Identity:
    kind = "identity"


class Linear:
    """gpflow.mean_functions.Linear(A, b): X @ A + b, b defaults to zeros(1)."""

    kind = "linear"

    def __init__(self, A=None, b=None):
        A = np.ones((1, 1)) if A is None else np.asarray(A, dtype=np.float64)
        b = np.zeros(1) if b is None else np.asarray(b, dtype=np.float64)
        self.A = Parameter(A, "A")
        self.b = Parameter(b, "b")


class kernels:           # `from dgp_dace.gpflow_compat import kernels; kernels.RBF(...)`
    SquaredExponential = SquaredExponential
    RBF = SquaredExponential
    Matern32, Matern52 = Matern32, Matern52
    Linear, White, Sum, Product = LinearKernel, White, Sum, Product


class likelihoods:
    Gaussian = Gaussian


class mean_functions:
    Zero, Identity, Linear = Zero, Identity, Linear


def kernel_from_any(k, input_dim):
    """Accept our stand-ins or real gpflow kernels (read by attribute): SquaredExponential/RBF, Matern32, Matern52
    with scalar or ARD lengthscales -- the kernels SO_BO.py:192-197,239-244 constructs."""
    if isinstance(k, SquaredExponential):
        if k.lengthscales.shape == (1,) and input_dim > 1:       # isotropic -> ARD storage
            k.lengthscales = Parameter(np.full(input_dim, k.lengthscales._value[0]), "lengthscales", "softplus")
        return k
    if isinstance(k, White):
        return k
    name = type(k).__name__
    if name == "Sum" and hasattr(k, "kernels"):                   # stationary + White (ours or gpflow's)
        out = Sum([White(variance=_val(q.variance)) if type(q).__name__ == "White" and not isinstance(q, White)
                   else kernel_from_any(q, input_dim) for q in k.kernels])
        split_white(out)
        return out
    if name not in _KERNEL_CLASSES:
        raise NotImplementedError(
            f"kernel {name}: the HIP path implements the stationary kernels the reference's DGP callers use "
            "(SquaredExponential/RBF, Matern32, Matern52); composite kernels belong to MF-DGP (DESIGN.md §9)")
    ls = np.atleast_1d(_val(k.lengthscales))
    if ls.size == 1 and input_dim > 1:
        ls = np.full(input_dim, ls[0])
    return _KERNEL_CLASSES[name](variance=_val(k.variance), lengthscales=ls)


def likelihood_from_any(lik):
    if isinstance(lik, Gaussian):
        return lik
    if type(lik).__name__ != "Gaussian":
        raise NotImplementedError("only the Gaussian likelihood is implemented (the reference's DGP callers use no other)")
    return Gaussian(variance=_val(lik.variance))


def mean_function_from_any(mf):
    if mf is None:
        return Zero()
    kind = getattr(mf, "kind", None) or type(mf).__name__.lower()
    if kind == "zero":
        return mf if isinstance(mf, Zero) else Zero()
    if kind == "identity":
        return mf if isinstance(mf, Identity) else Identity()
    if kind == "linear":
        return mf if isinstance(mf, Linear) else Linear(_val(mf.A), _val(mf.b))
    raise NotImplementedError(f"mean function {type(mf).__name__}")
