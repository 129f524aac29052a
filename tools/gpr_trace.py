"""A few exact-GP training-loss + gradient evaluations (N = 1000, D = 8) for rocprofv3 --kernel-trace."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))
from dgp_dace.models.gpr import GPR
from dgp_dace.gpflow_compat import Matern52
rng = np.random.default_rng(0)
Ng, D = 1000, 8
Xg = rng.uniform(-1, 1, (Ng, D)); Yg = np.sin(Xg @ rng.standard_normal((D, 1)))
gp = GPR((Xg, Yg), Matern52(1.0, np.ones(D)), noise_variance=1e-5)
for _ in range(5):
    gp.loss_and_grad()
