"""Small model (config 1): time of one optimize_nat_adam part-2 iteration, call by call (2 ELBO evaluations + Adam + natural gradient)."""
import os, sys, time, io, contextlib
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
X, Y, Z = synthetic(1000, 1, 32)
with contextlib.redirect_stdout(io.StringIO()):
    m = DGP(X, Y, Z, [RBF(1.0, [1.0]), RBF(1.0, [1.0])], [1], Gaussian(), num_samples=10)
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
mask = m._natgrad_setup(True); fl = m._trainable_flags()
ctx = m._sync_model(); m._sync_data(m.data); ctx.adam_reset()
def it(i):
    ctx.grad_step(10, 2*i, None); ctx.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
    ctx.grad_step(10, 2*i+1, None); ctx.natgrad_step(0.01, mask)
for i in range(20): it(i)
ctx.sync(); n = 300; t0 = time.perf_counter()
for i in range(n): it(100+i)
ctx.sync(); print("nat-adam part-2 iteration call by call: %.3f ms" % (1e3*(time.perf_counter()-t0)/n))
