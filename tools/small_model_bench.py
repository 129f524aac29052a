"""Iteration time of the launch-bound models (the reference's Bayesian-optimisation workloads, SO_BO.py:248-258):
call-by-call, the library's loop without and with the captured hipGraph.  Config 1 of BASELINE.json (N=1k, D=1, M=32,
`[1]`) and the notebook model (N=50, M=25, `[1,1]`)."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP


def model(kind):
    with contextlib.redirect_stdout(io.StringIO()):
        if kind == "config1":
            X, Y, Z = synthetic(1000, 1, 32)
            m = DGP(X, Y, Z, [RBF(1.0, [1.0]), RBF(1.0, [1.0])], [1], Gaussian(), num_samples=10)
        else:
            from helpers import notebook_data
            X, Y, Z = notebook_data()
            m = DGP(X, Y, Z, [RBF(1.0, [1.0]) for _ in range(3)], [1, 1], Gaussian(), num_samples=10)
    for l in m.layers[:-1]:
        l.q_sqrt.assign(l.q_sqrt * 1e-3)
    ctx = m._sync_model(); m._sync_data(m.data); ctx.adam_reset()
    return m, ctx


for kind in ("config1", "notebook"):
    n = 400
    m, ctx = model(kind); fl = m._trainable_flags()
    for i in range(20):
        ctx.grad_step(10, i, None); ctx.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
    ctx.sync(); t0 = time.perf_counter()
    for i in range(n):
        ctx.grad_step(10, 100 + i, None); ctx.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
    ctx.sync(); t_calls = (time.perf_counter() - t0) / n
    out = {}
    for g in (0, 1, -1):
        m, ctx = model(kind); fl = m._trainable_flags()
        ctx.adam_iterations(20, 10, 0, 0.01, 0.9, 0.999, 1e-7, fl, use_graph=g, want_elbo=False)
        ctx.sync(); t0 = time.perf_counter()
        ctx.adam_iterations(n, 10, 100, 0.01, 0.9, 0.999, 1e-7, fl, use_graph=g, want_elbo=False)
        ctx.sync(); out[g] = (time.perf_counter() - t0) / n
    mask = None
    m, ctx = model(kind); fl = m._trainable_flags(); mask = m._natgrad_setup(True); fl = m._trainable_flags()
    ctx.adam_iterations(20, 10, 0, 0.01, 0.9, 0.999, 1e-7, fl, 0.01, mask, use_graph=1, want_elbo=False)
    ctx.sync(); t0 = time.perf_counter()
    ctx.adam_iterations(n, 10, 100, 0.01, 0.9, 0.999, 1e-7, fl, 0.01, mask, use_graph=1, want_elbo=False)
    ctx.sync(); t_nat = (time.perf_counter() - t0) / n
    print(f"{kind}: Adam iteration call by call {1e3 * t_calls:.3f} ms | library loop {1e3 * out[0]:.3f} ms | captured graph "
          f"{1e3 * out[1]:.3f} ms | the library's choice (optimize_adam) {1e3 * out[-1]:.3f} ms | nat-adam part-2 iteration, "
          f"captured graph {1e3 * t_nat:.3f} ms", flush=True)
