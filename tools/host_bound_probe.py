"""Is a small-model iteration host-bound?  Enqueue time of n iterations (no synchronisation) against their total time."""
import os, sys, io, contextlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
X, Y, Z = synthetic(1000, 1, 32)
with contextlib.redirect_stdout(io.StringIO()):
    m = DGP(X, Y, Z, [RBF(1.0, [1.0]), RBF(1.0, [1.0])], [1], Gaussian(), num_samples=10)
for l in m.layers[:-1]:
    l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model(); m._sync_data(m.data); ctx.adam_reset(); fl = m._trainable_flags()
for i in range(20):
    ctx.grad_step(10, i, None); ctx.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
ctx.sync()
n = 200
t0 = time.perf_counter()
for i in range(n):
    ctx.grad_step(10, i, None); ctx.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / n:.3f} ms per iteration, enqueue + drain {1e3 * (t2 - t0) / n:.3f} ms per iteration")
