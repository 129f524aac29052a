import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
X, Y, Z = synthetic(20_000, 8, 256)
with contextlib.redirect_stdout(io.StringIO()):
    m = DGP(X, Y, Z, [RBF(1.0, [1.0] * 8) for _ in range(3)], [8, 8], Gaussian(), num_samples=10)
mask = m._natgrad_setup(True)
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
c = m._grad_step(m.data)
c.prof_enable(True)
for _ in range(5):
    c.natgrad_step(0.01, mask)
print({k: (round(v["ms"] / 5, 3), v["launches"] // 5) for k, v in c.prof_read().items()})
