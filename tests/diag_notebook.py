"""Diagnostic: per-parameter gradient error of the HIP path vs the oracle on the notebook model."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("dgp-toolbox_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import dgp_oracle as O, dgp_oracle_torch as T
from helpers import notebook_data, split_flat
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP

X, Y, Z = notebook_data()
m = DGP(X, Y, Z, [RBF(1.0, [1.0]) for _ in range(3)], [1, 1], Gaussian(), num_samples=10, seed=5)
mo = O.OracleDGP(X, Y, Z, [O.RBF(1.0, [1.0]) for _ in range(3)], [1, 1], num_samples=10)
for l in m.layers[:-1]:
    l.q_sqrt.assign(l.q_sqrt * 1e-3)
for l in mo.layers[:-1]:
    l.q_sqrt = l.q_sqrt * 1e-3
ctx = m._sync_model(); m._sync_data(m.data)
ctx.grad_partial(10, 5, None)
e = ctx.grad_finish(want_elbo=True)
zs = O.draw_zs(mo, 5, 10, 50)
eo, G = T.elbo_and_grads(mo, zs)
print("elbo", e, eo, abs(e - eo) / abs(eo))
Gp = split_flat(m, ctx.grad_get())
for i in range(3):
    for k in ("Z", "variance", "lengthscales", "q_mu", "q_sqrt"):
        a, b = np.asarray(Gp[(i, k)]), np.asarray(G["layers"][i][k])
        print(i, k, "max|g|=%.3e maxabs=%.3e rel_to_max=%.2e" % (np.abs(b).max(), np.abs(a - b).max(), np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)))
print("lik", Gp[("lik", "variance")], G["lik_variance"])
# forward pieces
Fs, Fm, Fv = m.propagate(X, S=10, zs=zs)
Fso, Fmo, Fvo = mo.propagate(X, 10, zs)
for i in range(3):
    print("layer", i, "mean err %.2e var err %.2e F err %.2e" % (np.abs(Fm[i] - Fmo[i]).max(), np.abs(Fv[i] - Fvo[i]).max(), np.abs(Fs[i] - Fso[i]).max()))
for i in range(2):
    a, b = np.asarray(Gp[(i, "q_sqrt")])[0], np.asarray(G["layers"][i]["q_sqrt"])[0]
    sl = np.tril_indices(25, -1)
    print("layer", i, "strictly-lower q_sqrt grad: gpu max %.3e  oracle max %.3e" % (np.abs(a[sl]).max(), np.abs(b[sl]).max()))
    print("   gpu rms %.3e oracle rms %.3e" % (np.sqrt((a[sl]**2).mean()), np.sqrt((b[sl]**2).mean())))
