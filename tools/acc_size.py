import sys, os, io, contextlib
sys.path.insert(0, "/root/repo/dgp-toolbox_amd"); sys.path.insert(0, "/root/repo")
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
X, Y, Z = synthetic(2000, 8, 256)
with contextlib.redirect_stdout(io.StringIO()):
    m = DGP(X, Y, Z, [RBF(1.0, [1.0] * 8) for _ in range(3)], [8, 8], Gaussian(), num_samples=10)
ctx = m._sync_model()
print("n_acc doubles", ctx.acc_info()[1], "MB", ctx.acc_info()[1] * 8 / 1e6)
