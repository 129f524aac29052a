"""Child process of tests/test_gpu_units.py::test_gram_kernel_is_exact_for_any_grid: the weighted Gram kernel's XCD-grouped
work split (csrc/gemm_gram.h: gram_range) at the workgroup count given by DGP_GRAM_GRID (read once per process), against
NumPy:  G_d += sum_p s[p, d] c_p c_p^T  (SURVEY App. C step 2), lower triangles, with du += C^T mbar riding on the same launch
(form DU: App. C step 1), and the two-source form Q' = A^T C."""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "dgp-toolbox_amd"))
from dgp_dace import _native  # noqa: E402

c = _native.Context(0)
lower = np.tril(np.ones((256, 256), dtype=bool))
for P, D in ((8192, 8), (20000, 8), (30000 - 30000 % 16, 3), (8192 + 16, 1), (65536 + 48, 5)):
    rng = np.random.default_rng(P + D)
    Cm = rng.standard_normal((P, 256)); s = rng.standard_normal((P, D)); G0 = rng.standard_normal((D, 256, 256))
    mb = rng.standard_normal((P, D)); du0 = rng.standard_normal((256, D))
    got, du = c.dev_gram(Cm, s, G0, mb=mb, du0=du0)
    ref = du0 + Cm.T @ mb
    assert np.abs(du - ref).max() <= 1e-12 * np.abs(ref).max(), (P, D, "du", np.abs(du - ref).max())
    assert np.array_equal(c.dev_gram(Cm, s, G0), got), (P, D, "the triangles do not depend on the du form")
    for d in range(D):
        ref = G0[d] + (Cm * s[:, d:d + 1]).T @ Cm
        err = np.abs(got[d] - ref)[lower].max()
        assert err <= 1e-12 * np.abs(ref).max(), (P, D, d, err)
    A = rng.standard_normal((P, 256)); C0 = rng.standard_normal((256, 256))
    q = c.dev_gemm("TN", A, Cm, C0=C0, beta=1, splits=8, tri=3, triblk=256)
    ref = C0 + A.T @ Cm
    assert np.abs(q - ref)[lower].max() <= 1e-12 * np.abs(ref).max(), (P, "two sources")
    print("ok", P, D, flush=True)
c.close()
