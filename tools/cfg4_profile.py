"""Config 4's shard shape (a rank's 125 000 of 10^6 points, D = 16, M = 512, three hidden layers): two gradient evaluations, for
rocprofv3 --kernel-trace --stats (tools/r4_cfg4.sh)."""
import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
X, Y, Z = synthetic(N, 16, 512)
with contextlib.redirect_stdout(io.StringIO()):
    m = DGP(X, Y, Z, [RBF(1.0, [1.0] * 16) for _ in range(4)], [16, 16, 16], Gaussian(), num_samples=10)
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model()
for _ in range(3):
    m._grad_step(m.data)
m.sync()
print("ELBO", ctx.last_elbo())
