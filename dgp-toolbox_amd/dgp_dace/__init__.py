"""dgp_dace — MI355X-native drop-in for the doubly-stochastic DGP path of Hebbalali/dgp-toolbox.

Same import path and class names as the reference package (``dgp_dace.models.dgp.DGP``); the
numerics run in hand-written HIP kernels (libdgp_hip.so) reached through a ctypes C-ABI.
"""
__all__ = ["models", "utils", "gpflow_compat"]
