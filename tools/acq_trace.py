"""One EI value + gradient evaluation for a single candidate (S = 1000, config-2 model) repeated: run under
rocprofv3 --kernel-trace to see where the 3-4 ms of an acquisition Adam step go (tools/acq_bench.py times it)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
from dgp_dace import Infill_criteria as IC
rng = np.random.default_rng(0)
N, D, M, units, S = 5000, 8, 256, [8, 8], 1000
X = rng.uniform(0, 1, (N, D)); X = (X - X.mean(0)) / X.std(0)
Y = np.sin(X @ rng.standard_normal((D, 1))); Y = (Y - Y.mean(0)) / Y.std(0)
Z = X[rng.permutation(N)[:M]]
m = DGP(X, Y, Z, [RBF(1.0, np.ones(d)) for d in [D] + units], units, Gaussian(), num_samples=10)
for l in m.layers[:-1]:
    l.q_sqrt.assign(l.q_sqrt.numpy() * 1e-3)
c = IC.EI(float(Y.min()), D)
x1 = rng.uniform(X.min(0), X.max(0), (1, D))
for _ in range(6):
    c._value_and_grad(m, x1, num_samples=S)
