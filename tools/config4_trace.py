"""Config-4 shape, a minibatch of 10 000 points: a few Adam + natural-gradient iterations for rocprofv3 --kernel-trace."""
import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
X, Y, Z = synthetic(10_000, 16, 512)
with contextlib.redirect_stdout(io.StringIO()):
    m = DGP(X, Y, Z, [RBF(1.0, [1.0] * 16) for _ in range(4)], [16, 16, 16], Gaussian(), num_samples=10)
mask = m._natgrad_setup(True)
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
for i in range(4):
    c = m._grad_step(m.data); c.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
    c = m._grad_step(m.data); c.natgrad_step(0.01, mask)
m.sync()
