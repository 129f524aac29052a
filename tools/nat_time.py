import os, sys, io, contextlib, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP
def run(N, D, M, units):
    X, Y, Z = synthetic(N, D, M)
    with contextlib.redirect_stdout(io.StringIO()):
        m = DGP(X, Y, Z, [RBF(1.0, [1.0] * d) for d in [D] + units], units, Gaussian(), num_samples=10)
    mask = m._natgrad_setup(True)
    for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
    ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
    def t(f, n=5):
        f(); m.sync(); t0 = time.perf_counter()
        for _ in range(n): f()
        m.sync(); return (time.perf_counter() - t0) / n * 1e3
    g = lambda: m._grad_step(m.data)
    a = lambda: (m._grad_step(m.data), ctx.adam_step(0.01, 0.9, 0.999, 1e-7, fl))
    ng = lambda: ctx.natgrad_step(0.01, mask)
    gn = lambda: (m._grad_step(m.data), ctx.natgrad_step(0.01, mask))
    both = lambda: (a(), gn())
    print("  Part-2 iteration (grad+adam, grad+natgrad): %.2f ms, again %.2f ms" % (t(both), t(both)), flush=True)
    print(N, D, M, units, "grad %.2f  grad+adam %.2f  natgrad alone %.2f  grad+natgrad %.2f ms" % (t(g), t(a), t(ng), t(gn)), flush=True)
run(100_000, 8, 256, [8, 8])
run(10_000, 16, 512, [16, 16, 16])
