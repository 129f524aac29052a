"""Two-fidelity model with different input spaces (the setting of Notebooks_dgp/nb_mfdgpem.ipynb) on the HIP engine.

Low fidelity: 40 points of a 2-D function; high fidelity: 12 points of a related 1-D function whose nominal mapping to
the low-fidelity inputs is x -> (x, 0.5).  Run from the repository root:  python examples/mf_dgp_em.py
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgp-toolbox_amd"))
import numpy as np
from dgp_dace.models.MF_DGP_EM import MultiFidelityDeepGP_EM

rng = np.random.default_rng(0)
lf = lambda x: np.sin(6 * x[:, :1]) + 0.3 * x[:, 1:2]
hf = lambda x: 1.5 * lf(np.concatenate([x, 0.5 * np.ones_like(x)], 1)) + 0.2 * x
X_lf, X_hf = rng.uniform(0, 1, (40, 2)), rng.uniform(0, 1, (12, 1))
X, Y = [X_lf, X_hf], [lf(X_lf), hf(X_hf)]
X_red = [np.concatenate([X_hf, 0.5 * np.ones_like(X_hf)], 1)]      # nominal low-fidelity inputs of the high-fidelity points

model = MultiFidelityDeepGP_EM(X, Y, X_red, seed=0)
model.model.num_samples = 20
model.optimize_adam(iterations1=300, iterations2=300, iterations3=600, messages=200)
Xt = np.linspace(0, 1, 11)[:, None]
mean, var = model.predict(Xt)
print("rmse on the high-fidelity function:", float(np.sqrt(np.mean((mean - hf(Xt)) ** 2))))
for x, m, v, t in zip(Xt[:, 0], mean[:, 0], var[:, 0], hf(Xt)[:, 0]):
    print(f"x = {x:4.2f}   mean {m:7.3f}   sd {np.sqrt(v):6.3f}   truth {t:7.3f}")
