"""Config-2 iteration time with each stationary kernel (RBF / Matern32 / Matern52 in every layer)."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
import numpy as np
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Matern32, Matern52, Gaussian
from dgp_dace.models.dgp import DGP
X, Y, Z = synthetic(100_000, 8, 256)
for name, K in (("rbf", RBF), ("matern32", Matern32), ("matern52", Matern52)):
    with contextlib.redirect_stdout(io.StringIO()):
        m = DGP(X, Y, Z, [K(1.0, [1.0] * 8) for _ in range(3)], [8, 8], Gaussian(), num_samples=10)
    for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
    ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
    def it():
        c = m._grad_step(m.data); c.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
    for _ in range(3): it()
    m.sync(); t0 = time.perf_counter()
    for _ in range(10): it()
    m.sync(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt*1e3:.2f} ms/iteration, ELBO {ctx.last_elbo():.3f}", flush=True)
    del m, ctx
