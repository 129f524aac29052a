"""Time of the row-parallel point products against the number of points (fixed cost per launch vs per-point cost)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))
import numpy as np
from dgp_dace import _native
ctx = _native.Context(0)
rng = np.random.default_rng(0)
for (K, N, tri, name) in ((256, 256, 1, "c = Kt Lu^-T (tri)"), (256, 256, 0, "dense 256x256"), (256, 2048, 0, "T = Ct Wcat (dense)")):
    B = rng.standard_normal((K, N))
    out = []
    for P in (15616, 31232, 62464, 124928, 249984, 499968):
        A = rng.standard_normal((P, K))
        _, ms = ctx.dev_gemm("NN", A, B, tri=tri, triblk=0, repeats=20)
        out.append((P, ms))
    base = out[-1][1] / out[-1][0]
    print(name, " ".join(f"P={p}: {ms*1e3:.0f} us ({ms/(base*p):.2f}x of linear)" for p, ms in out), flush=True)
