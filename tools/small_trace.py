"""Config 1 (or `notebook`), Adam iterations call by call - to be run under rocprofv3 --kernel-trace and read with
tools/iter_timeline.py <csv> 0.  usage: python tools/small_trace.py [config1|notebook]"""
import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
kind = sys.argv[1] if len(sys.argv) > 1 else "config1"
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
ctx = m._sync_model(); m._sync_data(m.data); ctx.adam_reset(); fl = m._trainable_flags()
for i in range(40):
    ctx.grad_step(10, i, None); ctx.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
ctx.sync()
