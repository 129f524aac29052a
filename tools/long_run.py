"""Stability check: 300 training iterations at the headline configuration (ELBO trend, timing drift, device memory)."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
X, Y, Z = synthetic(100_000, 8, 256)
with contextlib.redirect_stdout(io.StringIO()):
    m = DGP(X, Y, Z, [RBF(1.0, [1.0] * 8) for _ in range(3)], [8, 8], Gaussian(), num_samples=10)
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
free0 = None
for blk in range(6):
    t0 = time.perf_counter()
    for _ in range(50):
        c = m._grad_step(m.data); c.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
    e = ctx.last_elbo(); dt = (time.perf_counter() - t0) / 50
    free, total = torch.cuda.mem_get_info()
    free0 = free0 or free
    print(f"iterations {50*(blk+1):4d}: ELBO {e:.2f}  {dt*1e3:.2f} ms/iteration  device memory in use {(total-free)/2**30:.1f} GiB", flush=True)
assert np.isfinite(e) and abs(free - free0) < 2**30
