#!/usr/bin/env python3
"""Acquisition-side timing: one differential-evolution generation of EI (population x S samples through
predict_f) and one Adam step on a single candidate (predict_f + propagate_vjp), on the config-2 model.
usage: python tools/acq_bench.py [population] [S]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
from dgp_dace import Infill_criteria as IC

pop = int(sys.argv[1]) if len(sys.argv) > 1 else 300
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(0)
N, D, M, units = 5000, 8, 256, [8, 8]          # the surrogate's training set size does not enter prediction cost
X = rng.uniform(0, 1, (N, D)); X = (X - X.mean(0)) / X.std(0)
Y = np.sin(X @ rng.standard_normal((D, 1))); Y = (Y - Y.mean(0)) / Y.std(0)
Z = X[rng.permutation(N)[:M]]
m = DGP(X, Y, Z, [RBF(1.0, np.ones(d)) for d in [D] + units], units, Gaussian(), num_samples=10)
for l in m.layers[:-1]:
    l.q_sqrt.assign(l.q_sqrt.numpy() * 1e-3)
c = IC.EI(float(Y.min()), D)
cand = rng.uniform(X.min(0), X.max(0), (pop, D))
c.run(m, cand, num_samples=S)                   # warm-up (workspace allocation)
t = time.perf_counter(); reps = 5
for _ in range(reps):
    c.run(m, cand, num_samples=S)
dt = (time.perf_counter() - t) / reps
print(f"EI over a population of {pop} x S={S} (P={pop*S} points, 3 layers M={M}): {dt*1e3:.2f} ms per generation "
      f"= {pop*S/dt/1e6:.2f} M points/s")
x1 = cand[:1]
c._value_and_grad(m, x1, num_samples=S)
t = time.perf_counter(); reps = 20
for _ in range(reps):
    c._value_and_grad(m, x1, num_samples=S)
dt = (time.perf_counter() - t) / reps
print(f"EI value + gradient for one candidate, S={S}: {dt*1e3:.2f} ms per Adam step")
cb = cand[:64]
c._value_and_grad(m, cb, num_samples=S)
t = time.perf_counter(); reps = 5
for _ in range(reps):
    c._value_and_grad(m, cb, num_samples=S)
dt = (time.perf_counter() - t) / reps
print(f"EI value + gradient for 64 candidates, S={S} (P={64*S}): {dt*1e3:.2f} ms")

# exact GP (num_layers == 0): one training-loss + gradient evaluation and one EI population evaluation
from dgp_dace.models.gpr import GPR
from dgp_dace.gpflow_compat import Matern52
Ng = 1000
Xg = rng.uniform(-1, 1, (Ng, D)); Yg = np.sin(Xg @ rng.standard_normal((D, 1)))
gp = GPR((Xg, Yg), Matern52(1.0, np.ones(D)), noise_variance=1e-5)
gp.loss_and_grad()
t = time.perf_counter(); reps = 10
for _ in range(reps):
    gp.loss_and_grad()
dt = (time.perf_counter() - t) / reps
print(f"exact GP, N={Ng}, D={D}: training loss + gradient {dt*1e3:.2f} ms per Adam iteration")
cg = IC.EI(float(Yg.min()), D)
candg = rng.uniform(-1, 1, (pop, D))
cg.run(gp, candg)
t = time.perf_counter()
for _ in range(reps):
    cg.run(gp, candg)
dt = (time.perf_counter() - t) / reps
print(f"exact GP: EI over a population of {pop}: {dt*1e3:.2f} ms per generation")
