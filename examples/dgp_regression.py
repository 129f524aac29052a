"""The flow of the reference's Notebooks_dgp/nb_DGP_regression.ipynb on the HIP engine (needs an MI355X).

Step-function data (cells 2-10), a DGP with two hidden layers of width 1 (cell 18), Adam then Adam + natural gradients
(cells 22, 26), prediction (cells 34-41).  Run from the repository root:  python examples/dgp_regression.py
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgp-toolbox_amd"))
import numpy as np
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP

np.random.seed(0)
f_step = lambda x: 0.0 if x < 0.5 else 1.0
X = np.random.uniform(0, 1, 50)[:, None]
Z = np.random.uniform(0, 1, 25)[:, None]
Y = np.array([f_step(x) for x in X[:, 0]])[:, None] + np.random.randn(50, 1) * 1e-2

kernels = [RBF(lengthscales=[1.0], variance=1.0) for _ in range(3)]
model = DGP(X, Y, Z, kernels, num_units=[1, 1], likelihood=Gaussian(), num_samples=10)
print("trainable parameters:", model.number_parameters(trainable=True))
model.optimize_adam(iterations=300, lr=0.01, messages=100)
model.optimize_nat_adam(iterations1=100, iterations2=300, lr_adam=0.01, lr_gamma=0.01, messages=100)

Xs = np.linspace(-0.1, 1.1, 13)[:, None]
mean, var = model.predict(Xs, num_samples=50)
for x, m, v in zip(Xs[:, 0], mean[:, 0], var[:, 0]):
    print(f"x = {x:5.2f}   mean {m:7.3f}   sd {np.sqrt(v):6.3f}")
samples, Fmeans, Fvars = model.propagate(Xs, S=5)
print("layer outputs:", [tuple(s.shape) for s in samples])
