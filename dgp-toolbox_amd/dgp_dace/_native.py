"""ctypes binding of libdgp_hip.so (include/dgp_abi.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is usable, the
first compute call raises ``NativeUnavailable`` — the reference would have run TensorFlow on the
CPU here (dgp.py:102-109); this engine deliberately does not.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGP_HIP_LIB", os.path.join(os.path.dirname(_HERE), "libdgp_hip.so"))

DGP_OK, ERR_INVALID, ERR_HIP, ERR_NOT_PD, ERR_NO_DEVICE, ERR_NONFINITE = 0, -1, -2, -3, -4, -5

# every symbol include/dgp_abi.h declares
SYMBOLS = [
    "dgp_create", "dgp_destroy", "dgp_last_error", "dgp_sync", "dgp_device_info", "dgp_hip_runtimes", "dgp_model_set", "dgp_param_count",
    "dgp_params_get", "dgp_params_set", "dgp_data_set", "dgp_set_workspace_limit", "dgp_batch_set", "dgp_elbo", "dgp_propagate",
    "dgp_propagate_vjp", "dgp_vjp_accumulate", "dgp_propagate_full_cov", "dgp_gpr_lml", "dgp_gpr_predict", "dgp_gpr_predict_vjp",
    "dgp_grad_partial", "dgp_acc_info", "dgp_acc_bind", "dgp_grad_finish", "dgp_grad_get", "dgp_last_elbo", "dgp_grad_step",
    "dgp_comm_available", "dgp_comm_unique_id", "dgp_comm_init", "dgp_comm_destroy", "dgp_comm_allreduce",
    "dgp_adam_reset", "dgp_adam_step", "dgp_adam_iterations", "dgp_natgrad_step", "dgp_prof_enable", "dgp_prof_read", "dgp_prof_mark", "dgp_prof_marks_read", "dgp_dev_gemm", "dgp_dev_gram",
    "dgp_dev_layer_products", "dgp_dev_rbf_contract", "dgp_dev_g_panel",
    "dgp_dev_chol", "dgp_dev_trinv", "dgp_dev_normals", "dgp_dev_mfma_peak",
]


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdgp_hip error {code}: {msg}")
        self.code = code


class NativeUnavailable(RuntimeError):
    pass


class NotPositiveDefinite(NativeError):
    """Cholesky met a non-positive pivot (the reference surfaces tf InvalidArgumentError)."""


class LayerDesc(C.Structure):
    _fields_ = [("D_in", C.c_int32), ("D_out", C.c_int32), ("M", C.c_int32), ("white", C.c_int32),
                ("kernel_kind", C.c_int32), ("mean_kind", C.c_int32), ("kernel_white", C.c_int32)]


_dp = C.POINTER(C.c_double)
_dpp = C.POINTER(_dp)
_lib = None


def _one_hip_runtime():
    """One ROCm stack per process.  torch's wheels ship their own libamdhip64.so / libhsa-runtime64.so / librccl.so with the
    SAME sonames as the system ROCm's.  The dynamic loader resolves this library's `libamdhip64.so.7` to whichever copy is
    already in the process - but torch asks for its copies by file name, so `import torch` AFTER this library brings a
    second stack in.  Seen on the GPU box when that happened and the process also created an RCCL communicator or touched
    torch.cuda: `ncclCommInitRank: unhandled cuda error` (RCCL of one stack handed the other's streams), `double free or
    corruption` / `free(): invalid pointer` at exit.  Mapping only torch's libamdhip64 first was not enough (same aborts
    at exit).  The order that every multi-rank launch has anyway - torch first, then this library, which then binds to
    torch's stack, RCCL included (csrc/dgp_abi.hip: nccl_load) - is the one that runs clean, so when torch is installed
    it is imported before the library is loaded.  DGP_HIP_RUNTIME=system skips this (a process that never imports
    torch and wants the system ROCm)."""
    if os.environ.get("DGP_HIP_RUNTIME", "shared") == "system" or "torch" in sys.modules:
        return
    try:
        import torch  # noqa: F401
    except Exception:       # noqa: BLE001   (no torch: the system runtime is the only one)
        pass


def hip_runtimes_in(mapped_paths):
    """Distinct copies of libamdhip64 among the paths of a process's mapped objects (the lines of /proc/self/maps, or what
    dl_iterate_phdr reports): one entry per real file.  More than one = two ROCm stacks in the process (see
    `_one_hip_runtime`)."""
    seen = []
    for line in mapped_paths:
        path = line.split()[-1] if line.split() else ""
        if not path.startswith("/") or not os.path.basename(path).startswith("libamdhip64.so"):
            continue
        real = os.path.realpath(path) if os.path.exists(path) else path
        if real not in seen:
            seen.append(real)
    return seen


def check_one_hip_runtime(paths=None):
    """Raise NativeUnavailable naming the copies when more than one libamdhip64 is mapped (DGP_ALLOW_TWO_RUNTIMES=1 turns
    the refusal into a warning; `dgp_comm_init` refuses regardless).  `paths`: a list to judge instead of this process."""
    if paths is None:
        buf = C.create_string_buffer(4096)
        n = _lib.dgp_hip_runtimes(buf, len(buf)) if _lib is not None else 0
        paths = buf.value.decode().split("\n") if n else []
    paths = hip_runtimes_in(paths)
    if len(paths) <= 1:
        return paths
    msg = (f"{len(paths)} copies of libamdhip64 are mapped in this process ({', '.join(paths)}): two ROCm stacks. "
           "`import torch` BEFORE the first use of dgp_dace (or never import it, with DGP_HIP_RUNTIME=system) so that "
           "libdgp_hip.so and torch share one; mixing them ends in `ncclCommInitRank: unhandled cuda error` or an abort at exit.")
    if os.environ.get("DGP_ALLOW_TWO_RUNTIMES", "0") == "1":
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
        return paths
    raise NativeUnavailable(msg)


def load():
    """Load the shared library (no device is touched)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeUnavailable(
            f"{LIB_PATH} not found: build it with `make -C dgp-toolbox_amd/csrc` (hipcc, gfx950). "
            "There is no CPU fallback for the DGP hot path.")
    _one_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    sig = {
        "dgp_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
        "dgp_destroy": (None, [vp]),
        "dgp_last_error": (C.c_char_p, [vp]),
        "dgp_sync": (C.c_int, [vp]),
        "dgp_device_info": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(i64)]),
        "dgp_hip_runtimes": (C.c_int, [C.c_char_p, C.c_int]),
        "dgp_model_set": (C.c_int, [vp, C.c_int, C.POINTER(LayerDesc), _dp, i64, _dp, i64]),
        "dgp_param_count": (i64, [vp]),
        "dgp_params_get": (C.c_int, [vp, _dp]),
        "dgp_params_set": (C.c_int, [vp, _dp]),
        "dgp_data_set": (C.c_int, [vp, _dp, _dp, i64, i32, i32, i64]),
        "dgp_set_workspace_limit": (C.c_int, [vp, i64]),
        "dgp_batch_set": (C.c_int, [vp, i64, i64, dbl]),
        "dgp_elbo": (C.c_int, [vp, i32, u64, _dpp, _dp, _dp]),
        "dgp_propagate": (C.c_int, [vp, _dp, i64, i32, u64, _dpp, _dpp, _dpp, _dpp, i32]),
        "dgp_propagate_vjp": (C.c_int, [vp, _dp, i64, i32, u64, _dpp, _dp, _dp, _dp, _dp]),
        "dgp_vjp_accumulate": (C.c_int, [vp, _dp, i64, i32, u64, _dpp, _dp, _dp, _dp, _dp, i32]),
        "dgp_propagate_full_cov": (C.c_int, [vp, _dp, i64, i32, u64, _dpp, _dpp, _dpp, _dpp]),
        "dgp_gpr_lml": (C.c_int, [vp, i32, _dp, _dp, i64, i32, i32, C.c_double, _dp, C.c_double, _dp, _dp]),
        "dgp_gpr_predict": (C.c_int, [vp, i32, _dp, _dp, i64, i32, i32, C.c_double, _dp, C.c_double, _dp, i64, i32, _dp, _dp]),
        "dgp_gpr_predict_vjp": (C.c_int, [vp, i32, _dp, _dp, i64, i32, i32, C.c_double, _dp, C.c_double, _dp, i64, _dp, _dp, _dp]),
        "dgp_grad_partial": (C.c_int, [vp, i32, u64, _dpp]),
        "dgp_acc_info": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64)]),
        "dgp_acc_bind": (C.c_int, [vp, vp]),
        "dgp_grad_finish": (C.c_int, [vp, _dp]),
        "dgp_grad_get": (C.c_int, [vp, _dp]),
        "dgp_last_elbo": (C.c_int, [vp, _dp]),
        "dgp_grad_step": (C.c_int, [vp, i32, u64, _dpp, _dp]),
        "dgp_comm_available": (C.c_int, []),
        "dgp_comm_unique_id": (C.c_int, [vp]),
        "dgp_comm_init": (C.c_int, [vp, i32, i32, vp]),
        "dgp_comm_destroy": (C.c_int, [vp]),
        "dgp_comm_allreduce": (C.c_int, [vp, vp, i64]),
        "dgp_adam_reset": (C.c_int, [vp]),
        "dgp_adam_step": (C.c_int, [vp, dbl, dbl, dbl, dbl, C.POINTER(C.c_uint8)]),
        "dgp_natgrad_step": (C.c_int, [vp, dbl, C.POINTER(C.c_uint8)]),
        "dgp_adam_iterations": (C.c_int, [vp, i32, i32, u64, dbl, dbl, dbl, dbl, C.POINTER(C.c_uint8), dbl, C.POINTER(C.c_uint8),
                                          i32, _dp]),
        "dgp_prof_enable": (C.c_int, [vp, i32]),
        "dgp_prof_read": (C.c_int, [vp, i32, _dp, C.POINTER(i64), _dp, _dp]),
        "dgp_prof_mark": (C.c_int, [vp]),
        "dgp_prof_marks_read": (C.c_int, [vp, i32, _dp, C.POINTER(i32)]),
        "dgp_dev_gemm": (C.c_int, [vp, i32, i64, i64, i64, _dp, i64, _dp, i64, _dp, i64, dbl, i32, i32, i32, i64, i32, _dp]),
        "dgp_dev_gram": (C.c_int, [vp, _dp, _dp, i64, i32, _dp, _dp, _dp]),
        "dgp_dev_layer_products": (C.c_int, [vp, i64, i32, i32] + [_dp] * 15 + [C.POINTER(i32)]),
        "dgp_dev_rbf_contract": (C.c_int, [vp, _dp, _dp, _dp, i64, i32, i32, _dp, _dp, C.POINTER(i32)]),
        "dgp_dev_g_panel": (C.c_int, [vp, _dp, _dp, _dp, _dp, _dp, i64, i32, _dp, _dp, C.POINTER(i32)]),
        "dgp_dev_chol": (C.c_int, [vp, _dp, i32, i32]),
        "dgp_dev_trinv": (C.c_int, [vp, _dp, _dp, i32, i32]),
        "dgp_dev_normals": (C.c_int, [vp, u64, i32, i32, i64, i64, i32, _dp]),
        "dgp_dev_mfma_peak": (C.c_int, [vp, i32, _dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    check_one_hip_runtime()
    return lib


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr_array(arrs):
    """host list of arrays (or None entries) -> (double*[]), keeping the arrays alive"""
    T = _dp * len(arrs)
    return T(*[(_ptr(a) if a is not None else None) for a in arrs])


class Context:
    """Owns one dgp_ctx (one device, one stream)."""

    def __init__(self, device=0, stream=None):
        lib = load()
        check_one_hip_runtime()        # a later `import torch` may have brought a second ROCm stack in since load()
        h = C.c_void_p()
        rc = lib.dgp_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc == ERR_NO_DEVICE:
            raise NativeUnavailable("no usable AMD GPU (gfx950) visible to HIP: the DGP hot path has no CPU fallback")
        if rc != DGP_OK:
            raise NativeError(rc, "dgp_create failed")
        self._h, self._lib, self.device = h, lib, int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dgp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc == DGP_OK:
            return
        msg = self._lib.dgp_last_error(self._h).decode()
        if rc == ERR_NOT_PD:
            raise NotPositiveDefinite(rc, msg)
        raise NativeError(rc, msg)

    # ---- model / data ---------------------------------------------------------------------
    def model_set(self, descs, flat_params, mean_params):
        arr = (LayerDesc * len(descs))(*[LayerDesc(*d) for d in descs])
        fp = _c(flat_params)
        mp = _c(mean_params) if mean_params is not None and len(mean_params) else np.zeros(0)
        self._chk(self._lib.dgp_model_set(self._h, len(descs), arr, _ptr(fp), fp.size,
                                          _ptr(mp) if mp.size else None, mp.size))
        self.n_layers = len(descs)
        self.douts = [d[1] for d in descs]

    def param_count(self):
        return int(self._lib.dgp_param_count(self._h))

    def params_get(self):
        out = np.empty(self.param_count())
        self._chk(self._lib.dgp_params_get(self._h, _ptr(out)))
        return out

    def params_set(self, flat):
        flat = _c(flat)
        assert flat.size == self.param_count()
        self._chk(self._lib.dgp_params_set(self._h, _ptr(flat)))

    def data_set(self, X, Y, n_global_offset=0):
        X, Y = _c(X), _c(Y)
        self._chk(self._lib.dgp_data_set(self._h, _ptr(X), _ptr(Y), X.shape[0], X.shape[1], Y.shape[1],
                                         int(n_global_offset)))
        self.n_data = X.shape[0]

    def batch_set(self, start=0, count=0, scale=1.0):
        """Evaluate the bound on the resident points [start, start+count) with the data term times `scale`
        (count == 0: all points)."""
        self._chk(self._lib.dgp_batch_set(self._h, int(start), int(count), float(scale)))

    def set_workspace_limit(self, nbytes):
        self._chk(self._lib.dgp_set_workspace_limit(self._h, int(nbytes)))

    # ---- forward --------------------------------------------------------------------------
    def _zs(self, zs, S=None, N=None):
        """Injected normals: one [S, N, D_l] array per layer.  The library copies S*N*D_l doubles from each pointer, so
        a wrongly shaped array would be an out-of-bounds read: reject it here (the reference would raise a shape error)."""
        if zs is None:
            return None, None
        keep = [_c(z) for z in zs]
        if len(keep) != self.n_layers:
            raise ValueError(f"zs: expected {self.n_layers} arrays (one per layer), got {len(keep)}")
        for l, z in enumerate(keep):
            want = (S, N, self.douts[l])
            if z.ndim != 3 or z.shape[2] != want[2] or (S is not None and z.shape[0] != S) or (N is not None and z.shape[1] != N):
                raise ValueError(f"zs[{l}]: expected shape {want}, got {z.shape}")
        return _ptr_array(keep), keep

    def elbo(self, S, seed=0, zs=None):
        zp, keep = self._zs(zs, S, getattr(self, "n_data", None))
        a, b = C.c_double(), C.c_double()
        self._chk(self._lib.dgp_elbo(self._h, int(S), int(seed) & (2 ** 64 - 1), zp, C.byref(a), C.byref(b)))
        return a.value, b.value

    def propagate(self, Xnew, S, seed=0, zs=None, want=(True, True, True), add_lik_var=False):
        Xnew = _c(Xnew)
        Nn = Xnew.shape[0]
        zp, keep = self._zs(zs, S, Nn)
        outs = []
        for w in want:
            outs.append([np.empty((S, Nn, d)) if w else None for d in self.douts])
        if Nn == 0:          # no points: empty [S, 0, D_l] results, as the reference's tensors would be
            return outs
        ptrs = [_ptr_array(o) for o in outs]
        self._chk(self._lib.dgp_propagate(self._h, _ptr(Xnew), Nn, int(S), int(seed) & (2 ** 64 - 1), zp, ptrs[0],
                                          ptrs[1], ptrs[2], 1 if add_lik_var else 0))
        return outs

    def propagate_full_cov(self, Xnew, S, seed=0, zs=None):
        """full_cov=True propagation (dgp_propagate_full_cov): Fvars are [S, Nn, Nn, D_l]."""
        Xnew = _c(Xnew)
        Nn = Xnew.shape[0]
        zp, keep = self._zs(zs, S, Nn)
        Fs = [np.empty((S, Nn, d)) for d in self.douts]
        Fm = [np.empty((S, Nn, d)) for d in self.douts]
        Fv = [np.empty((S, Nn, Nn, d)) for d in self.douts]
        if Nn == 0:          # no points: empty results, as propagate() returns
            return Fs, Fm, Fv
        self._chk(self._lib.dgp_propagate_full_cov(self._h, _ptr(Xnew), Nn, int(S), int(seed) & (2 ** 64 - 1), zp,
                                                   _ptr_array(Fs), _ptr_array(Fm), _ptr_array(Fv)))
        return Fs, Fm, Fv

    # ---- exact GP regression (stateless calls) ---------------------------------------------------
    def gpr_lml(self, kind, X, Y, variance, lengthscales, noise, want_grad=True):
        X, Y, ls = _c(X), _c(Y), _c(lengthscales)
        lml = C.c_double()
        g = np.empty(2 + ls.size) if want_grad else None
        self._chk(self._lib.dgp_gpr_lml(self._h, int(kind), _ptr(X), _ptr(Y), X.shape[0], X.shape[1], Y.shape[1],
                                        float(variance), _ptr(ls), float(noise), C.byref(lml),
                                        _ptr(g) if want_grad else None))
        return lml.value, g

    def gpr_predict(self, kind, X, Y, variance, lengthscales, noise, Xnew, add_noise):
        X, Y, ls, Xnew = _c(X), _c(Y), _c(lengthscales), _c(Xnew)
        mean, var = np.empty((Xnew.shape[0], Y.shape[1])), np.empty((Xnew.shape[0], Y.shape[1]))
        self._chk(self._lib.dgp_gpr_predict(self._h, int(kind), _ptr(X), _ptr(Y), X.shape[0], X.shape[1], Y.shape[1],
                                            float(variance), _ptr(ls), float(noise), _ptr(Xnew), Xnew.shape[0],
                                            1 if add_noise else 0, _ptr(mean), _ptr(var)))
        return mean, var

    def gpr_predict_vjp(self, kind, X, Y, variance, lengthscales, noise, Xnew, mean_bar, var_bar):
        X, Y, ls, Xnew, mb, vb = _c(X), _c(Y), _c(lengthscales), _c(Xnew), _c(mean_bar), _c(var_bar)
        if mb.shape != (Xnew.shape[0], Y.shape[1]) or vb.shape != mb.shape:
            raise ValueError("cotangents must be [Nn, Dy]")
        out = np.empty(Xnew.shape)
        self._chk(self._lib.dgp_gpr_predict_vjp(self._h, int(kind), _ptr(X), _ptr(Y), X.shape[0], X.shape[1], Y.shape[1],
                                                float(variance), _ptr(ls), float(noise), _ptr(Xnew), Xnew.shape[0],
                                                _ptr(mb), _ptr(vb), _ptr(out)))
        return out

    def propagate_vjp(self, Xnew, S, seed=0, zs=None, f_bar=None, mean_bar=None, var_bar=None, accumulate=None):
        """d(sum of cotangent * last-layer output)/dXnew, [Nn, D_in] (dgp_propagate_vjp).

        accumulate: None = inputs only; "reset" / "add" = also add the parameter sums of this call to the gradient
        accumulator, cleared first or not (dgp_vjp_accumulate; finish with grad_finish / grad_get)."""
        Xnew = _c(Xnew)
        Nn = Xnew.shape[0]
        zp, keep = self._zs(zs, S, Nn)
        shape = (S, Nn, self.douts[-1])
        bars = []
        for b in (f_bar, mean_bar, var_bar):
            if b is None:
                bars.append(None)
                continue
            b = _c(b)
            if b.shape != shape:
                raise ValueError(f"cotangent of shape {b.shape}, expected {shape}")
            bars.append(b)
        out = np.empty((Nn, Xnew.shape[1]))
        if accumulate is None:
            self._chk(self._lib.dgp_propagate_vjp(self._h, _ptr(Xnew), Nn, int(S), int(seed) & (2 ** 64 - 1), zp,
                                                  *[_ptr(b) if b is not None else None for b in bars], _ptr(out)))
        else:
            self._chk(self._lib.dgp_vjp_accumulate(self._h, _ptr(Xnew), Nn, int(S), int(seed) & (2 ** 64 - 1), zp,
                                                   *[_ptr(b) if b is not None else None for b in bars], _ptr(out),
                                                   1 if accumulate == "reset" else 0))
        return out

    # ---- backward + optimisers --------------------------------------------------------------
    def grad_partial(self, S, seed=0, zs=None):
        zp, keep = self._zs(zs, S, getattr(self, "n_data", None))
        self._chk(self._lib.dgp_grad_partial(self._h, int(S), int(seed) & (2 ** 64 - 1), zp))

    def acc_info(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self._lib.dgp_acc_info(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def acc_bind(self, dev_ptr):
        self._chk(self._lib.dgp_acc_bind(self._h, C.c_void_p(dev_ptr) if dev_ptr else None))

    def grad_step(self, S, seed=0, zs=None, want_elbo=False):
        """grad_partial + all-reduce (when comm_init was called) + grad_finish in one overlapped call (dgp_grad_step)."""
        zp, keep = self._zs(zs, S, getattr(self, "n_data", None))
        if want_elbo:
            e = C.c_double()
            self._chk(self._lib.dgp_grad_step(self._h, int(S), int(seed) & (2 ** 64 - 1), zp, C.byref(e)))
            return e.value
        self._chk(self._lib.dgp_grad_step(self._h, int(S), int(seed) & (2 ** 64 - 1), zp, None))
        return None

    # ---- multi-GPU: library-owned RCCL communicator ---------------------------------------------
    @staticmethod
    def comm_available():
        """True when librccl.so loads with every entry point the library binds (no communicator is created).  The library
        takes the librccl that sits next to the HIP runtime it is bound to (csrc/dgp_abi.hip: nccl_load)."""
        return load().dgp_comm_available() == DGP_OK

    @staticmethod
    def comm_unique_id():
        """128 bytes (ncclUniqueId) from rank 0, to be broadcast to the other ranks by the host."""
        lib = load()
        buf = C.create_string_buffer(128)
        rc = lib.dgp_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != DGP_OK:
            raise NativeError(rc, "dgp_comm_unique_id failed (librccl.so not loadable?)")
        return bytes(buf.raw)

    def comm_init(self, rank, world, unique_id=None):
        buf = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        self._chk(self._lib.dgp_comm_init(self._h, int(rank), int(world), C.cast(buf, C.c_void_p) if buf is not None else None))
        self.comm_world = int(world)

    def comm_destroy(self):
        self._chk(self._lib.dgp_comm_destroy(self._h))
        self.comm_world = 1

    def comm_allreduce(self, dev_ptr, n):
        self._chk(self._lib.dgp_comm_allreduce(self._h, C.c_void_p(dev_ptr), int(n)))

    def grad_finish(self, want_elbo=False):
        if want_elbo:
            e = C.c_double()
            self._chk(self._lib.dgp_grad_finish(self._h, C.byref(e)))
            return e.value
        self._chk(self._lib.dgp_grad_finish(self._h, None))
        return None

    def grad_get(self):
        out = np.empty(self.param_count())
        self._chk(self._lib.dgp_grad_get(self._h, _ptr(out)))
        return out

    def last_elbo(self):
        e = C.c_double()
        self._chk(self._lib.dgp_last_elbo(self._h, C.byref(e)))
        return e.value

    def adam_reset(self):
        self._chk(self._lib.dgp_adam_reset(self._h))

    def adam_step(self, lr, beta_1, beta_2, epsilon, trainable):
        t = (C.c_uint8 * len(trainable))(*[1 if x else 0 for x in trainable])
        self._chk(self._lib.dgp_adam_step(self._h, lr, beta_1, beta_2, epsilon, t))

    def adam_iterations(self, n_iter, S, seed0, lr, beta_1, beta_2, epsilon, trainable, gamma=0.0, layer_mask=None,
                        use_graph=-1, want_elbo=True):
        """n_iter loop bodies of optimize_adam (gamma == 0) or of part 2 of optimize_nat_adam (gamma > 0) in one call
        (dgp_adam_iterations; launch-bound models replay a captured hipGraph).  Returns the ELBOs the iterations would
        print (or None)."""
        t = (C.c_uint8 * len(trainable))(*[1 if x else 0 for x in trainable])
        m = (C.c_uint8 * len(layer_mask))(*[1 if x else 0 for x in layer_mask]) if layer_mask is not None else None
        out = np.empty(int(n_iter)) if want_elbo else None
        self._chk(self._lib.dgp_adam_iterations(self._h, int(n_iter), int(S), int(seed0) & (2 ** 64 - 1), lr, beta_1, beta_2,
                                                epsilon, t, float(gamma), m, int(use_graph), _ptr(out) if want_elbo else None))
        return out

    def natgrad_step(self, gamma, layer_mask):
        t = (C.c_uint8 * len(layer_mask))(*[1 if x else 0 for x in layer_mask])
        self._chk(self._lib.dgp_natgrad_step(self._h, gamma, t))

    def sync(self):
        self._chk(self._lib.dgp_sync(self._h))

    # ---- measurement ------------------------------------------------------------------------
    def device_info(self):
        buf = C.create_string_buffer(256)
        cu, mem = C.c_int(), C.c_int64()
        self._chk(self._lib.dgp_device_info(self._h, buf, 256, C.byref(cu), C.byref(mem)))
        return buf.value.decode(), cu.value, mem.value

    PROF_CATEGORIES = ("mfma_contractions", "per_point_streaming", "small_matrix_chain", "adam")

    def prof_enable(self, on=True, categories=None):
        """HIP-event timing of the launches by category; `categories` (names) restricts the event pairs to those - the others
        still count launches and algorithmic flops, their milliseconds read 0."""
        code = 1 if on else 0
        if on and categories is not None:
            code = 0x100 | sum(1 << self.PROF_CATEGORIES.index(c) for c in categories)
        self._chk(self._lib.dgp_prof_enable(self._h, code))

    def prof_read(self):
        ms, fl, by = np.zeros(4), np.zeros(4), np.zeros(4)
        ln = np.zeros(4, dtype=np.int64)
        self._chk(self._lib.dgp_prof_read(self._h, 4, _ptr(ms), ln.ctypes.data_as(C.POINTER(C.c_int64)), _ptr(fl), _ptr(by)))
        names = self.PROF_CATEGORIES
        return {n: {"ms": ms[i], "launches": int(ln[i]), "alg_flops": fl[i], "alg_bytes": by[i]} for i, n in enumerate(names)}

    def prof_mark(self):
        """Record a step boundary (HIP event) on the context's stream."""
        self._chk(self._lib.dgp_prof_mark(self._h))

    def prof_marks_read(self, n_max=65536):
        """Milliseconds between consecutive marks (synchronises on the last mark, clears the marks)."""
        out = np.zeros(int(n_max))
        n = C.c_int32()
        self._chk(self._lib.dgp_prof_marks_read(self._h, int(n_max), _ptr(out), C.byref(n)))
        return out[:n.value].copy()

    # ---- unit-level hooks -------------------------------------------------------------------
    def dev_gemm(self, op, A, B, C0=None, alpha=1.0, beta=0, splits=1, tri=0, triblk=0, repeats=0):
        A, B = _c(A), _c(B)
        opi = {"NN": 0, "NT": 1, "TN": 2}[op]
        M = A.shape[1] if opi == 2 else A.shape[0]
        K = A.shape[0] if opi == 2 else A.shape[1]
        N = B.shape[0] if opi == 1 else B.shape[1]
        Cm = np.zeros((M, N)) if C0 is None else _c(C0).copy()
        ms = C.c_double()
        self._chk(self._lib.dgp_dev_gemm(self._h, opi, M, N, K, _ptr(A), A.shape[1], _ptr(B), B.shape[1], _ptr(Cm), N,
                                         alpha, beta, splits, tri, triblk, repeats, C.byref(ms)))
        return (Cm, ms.value) if repeats else Cm

    def dev_gram(self, Cmat, s=None, G0=None, mb=None, du0=None):
        """G[d] (+)= sum_p s[p, d] c_p c_p^T (lower triangles) by the library's dispatcher: Cmat [P, 256], s [P, D] or None.
        With mb [P, D]: also du (+)= Cmat^T mb, inside the same launch where the Gram kernel runs it; returns (G, du)."""
        Cmat = _c(Cmat)
        D = 1 if s is None else s.shape[1]
        G = np.zeros((D, 256, 256)) if G0 is None else _c(G0).copy()
        sp = None if s is None else _c(s)
        if mb is None:
            self._chk(self._lib.dgp_dev_gram(self._h, _ptr(Cmat), None if sp is None else _ptr(sp), Cmat.shape[0], D, _ptr(G), None, None))
            return G
        mbp = _c(mb)
        du = np.zeros((256, D)) if du0 is None else _c(du0).copy()
        self._chk(self._lib.dgp_dev_gram(self._h, _ptr(Cmat), _ptr(sp), Cmat.shape[0], D, _ptr(G), _ptr(mbp), _ptr(du)))
        return G, du

    ENGINES = {0: "engine128x64", 1: "wide", 2: "tall", 3: "tallu", 4: "gram", 5: "small", 6: "mid", 7: "dcpanel"}

    def dev_layer_products(self, Kt, Linv, Wcat, u, vbar, mbar):
        """The point contractions of one SVGP layer (dgp_dev_layer_products) on explicit operands: Kt [P, Mp], Linv [Mp, Mp]
        lower, Wcat [Mp, D*Mp], u [Mp, D], vbar / mbar [P, D].  Returns a dict of Ct, cn, T, tn, mean0, Cbar, g, du, Gd (lower
        triangles of [D, Mp, Mp]) and `engines`: the kernel family that ran (Ct, T, Cbar, g, du, Gd)."""
        Kt, Linv, Wcat, u, vbar, mbar = (_c(a) for a in (Kt, Linv, Wcat, u, vbar, mbar))
        P, Mp = Kt.shape
        D = u.shape[1]
        assert Linv.shape == (Mp, Mp) and Wcat.shape == (Mp, D * Mp) and vbar.shape == (P, D) and mbar.shape == (P, D)
        out = {"Ct": np.empty((P, Mp)), "cn": np.empty(P), "T": np.empty((P, D * Mp)), "tn": np.empty((P, D)),
               "mean0": np.empty((P, D)), "Cbar": np.empty((P, Mp)), "g": np.empty((P, Mp)), "du": np.empty((Mp, D)),
               "Gd": np.empty((D, Mp, Mp))}
        eng = (C.c_int32 * 7)()
        self._chk(self._lib.dgp_dev_layer_products(self._h, P, Mp, D, _ptr(Kt), _ptr(Linv), _ptr(Wcat), _ptr(u), _ptr(vbar),
                                                   _ptr(mbar), *[_ptr(out[k]) for k in ("Ct", "cn", "T", "tn", "mean0", "Cbar", "g", "du", "Gd")],
                                                   eng))
        out["engines"] = [self.ENGINES[int(e)] for e in eng[:6]]
        out["mean_in_ct"] = bool(eng[6])
        return out

    def dev_rbf_contract(self, G, Z1, X1, GX0=None):
        """R1 = G Z1 and GX (+)= G^T X1 through the backward pass's launcher (dgp_dev_rbf_contract): G [P, Mp], Z1 [Mp, w1],
        X1 [P, w1].  Returns (R1, GX, fused): fused = both in one pass over G (points.hip), else two engine products."""
        G, Z1, X1 = _c(G), _c(Z1), _c(X1)
        P, Mp = G.shape
        w1 = Z1.shape[1]
        assert Z1.shape == (Mp, w1) and X1.shape == (P, w1)
        R1 = np.empty((P, w1))
        GX = np.zeros((Mp, w1)) if GX0 is None else _c(GX0).copy()
        f = C.c_int32(0)
        self._chk(self._lib.dgp_dev_rbf_contract(self._h, _ptr(G), _ptr(Z1), _ptr(X1), P, Mp, w1, _ptr(R1), _ptr(GX), C.byref(f)))
        return R1, GX, bool(f.value)

    def dev_g_panel(self, Cbar, Linv, E, Z1, X1, GX0=None):
        """R1 = g Z1 and GX (+)= g^T X1 for g = (Cbar Linv) .* E through the launch the backward pass uses at Mp = 256
        (dgp_dev_g_panel; g is never stored).  Returns (R1, GX, used); used False: the size goes down the stored-g path."""
        Cbar, Linv, E, Z1, X1 = _c(Cbar), _c(Linv), _c(E), _c(Z1), _c(X1)
        P, w1 = Cbar.shape[0], Z1.shape[1]
        assert Cbar.shape == (P, 256) and E.shape == (P, 256) and Linv.shape == (256, 256) and Z1.shape == (256, w1) and X1.shape == (P, w1)
        R1 = np.empty((P, w1))
        GX = np.zeros((256, w1)) if GX0 is None else _c(GX0).copy()
        u = C.c_int32(0)
        self._chk(self._lib.dgp_dev_g_panel(self._h, _ptr(Cbar), _ptr(Linv), _ptr(E), _ptr(Z1), _ptr(X1), P, w1, _ptr(R1), _ptr(GX), C.byref(u)))
        return R1, GX, bool(u.value)

    def dev_chol(self, A):
        A = _c(A).copy()
        b = 1 if A.ndim == 2 else A.shape[0]
        self._chk(self._lib.dgp_dev_chol(self._h, _ptr(A), A.shape[-1], b))
        return A

    def dev_trinv(self, L):
        L = _c(L)
        X = np.empty_like(L)
        b = 1 if L.ndim == 2 else L.shape[0]
        self._chk(self._lib.dgp_dev_trinv(self._h, _ptr(L), _ptr(X), L.shape[-1], b))
        return X

    def dev_normals(self, seed, layer, S, n0, N, D):
        out = np.empty((S, N, D))
        self._chk(self._lib.dgp_dev_normals(self._h, int(seed) & (2 ** 64 - 1), layer, S, n0, N, D, _ptr(out)))
        return out

    def dev_mfma_peak(self, iters=20000):
        t = C.c_double()
        self._chk(self._lib.dgp_dev_mfma_peak(self._h, iters, C.byref(t)))
        return t.value
