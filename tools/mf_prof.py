import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))
import numpy as np
from dgp_dace.models.MF_DGP_EM import MultiFidelityDeepGP_EM
rng = np.random.default_rng(0)
n_lf, n_hf, S = 2048, 512, 100
X0, X1 = rng.uniform(0, 1, (n_lf, 4)), rng.uniform(0, 1, (n_hf, 2))
lf = lambda x: np.sin(4 * x[:, :1]) + x[:, 1:2] * x[:, 2:3] - 0.5 * x[:, 3:4]
X_red = [np.concatenate([X1, 0.5 * np.ones((n_hf, 2))], 1)]
Y = [lf(X0), 1.3 * lf(X_red[0]) + 0.2 * X1[:, :1]]
mf = MultiFidelityDeepGP_EM([X0, X1], Y, X_red, seed=0)
mf.model.num_samples = S
mf._initialise(1e-2, 1e-2)
for _ in range(3): mf.model.ELBO_and_grad(mf._data())
