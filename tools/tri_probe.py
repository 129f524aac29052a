"""Unscaled triangular products through the engine's unit hook: executed-flop rate of the tile-level + half-tile skipping."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))
import numpy as np
from dgp_dace import _native
ctx = _native.Context(0)
rng = np.random.default_rng(0)
P, M, D = 249984, 256, 8
U = np.concatenate([np.triu(rng.standard_normal((M, M))) for _ in range(D)], 0)      # [D*M, M], each block upper-triangular
T = rng.standard_normal((P, D * M))
C, ms = ctx.dev_gemm("NN", T, U, tri=1, triblk=M, repeats=10)
print(f"multi-block upper-triangular B, K={D*M}, N={M}: {ms:.3f} ms  executed {0.5625*2.0*P*D*M*M/ms/1e9:.1f} TFLOP/s  (dense-equivalent {2.0*P*D*M*M/ms/1e9:.1f})  err {np.abs(C[:40]-T[:40]@U).max():.1e}", flush=True)
C, ms = ctx.dev_gemm("NN", T, U, repeats=10)
print(f"same product without the hint: {ms:.3f} ms  {2.0*P*D*M*M/ms/1e9:.1f} TFLOP/s", flush=True)
L = np.tril(rng.standard_normal((M, M)))
Wc = np.concatenate([np.tril(rng.standard_normal((M, M))) for _ in range(D)], 1)       # [M, D*M]
Ct = rng.standard_normal((P, M))
_, ms = ctx.dev_gemm("NN", Ct, Wc, tri=2, triblk=M, repeats=10)
print(f"single-block lower-triangular B, K={M}, N={D*M} (T product, no store epilogue extras): {ms:.3f} ms  executed {0.5625*2.0*P*D*M*M/ms/1e9:.1f} TFLOP/s", flush=True)
_, ms = ctx.dev_gemm("NN", Ct, Wc, repeats=10)
print(f"same without the hint: {ms:.3f} ms  {2.0*P*D*M*M/ms/1e9:.1f} TFLOP/s", flush=True)
