"""Time of the chains' Mp x Mp products (dgp_dev_gemm, back-to-back repeats on one stream): 32 x 32-tile kernel of gemm_mid.hip
(default) against the 128 x 64 engine (DGP_MID_GEMM=0).  usage: python tools/mid_gemm_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgp-toolbox_amd"))
from dgp_dace import _native
c = _native.Context(0)
rng = np.random.default_rng(0)
for (M, N, K) in ((256, 256, 256), (128, 128, 128), (256, 8, 256), (512, 512, 256)):
    for op in ("NN", "NT", "TN"):
        A = rng.standard_normal((K, M) if op == "TN" else (M, K))
        B = rng.standard_normal((N, K) if op == "NT" else (K, N))
        _, ms = c.dev_gemm(op, A, B, repeats=200)
        print(f"{os.environ.get('DGP_MID_GEMM', '1')} {op} {M}x{N}x{K}: {ms * 1000:.1f} us per call", flush=True)
c.close()
