"""Timing of the MF-DGP-EM bound + gradient on one GPU (BASELINE.json config 5 at the sizes the engine takes).

The reference assigns q_mu = Y per fidelity (MF_DGP_EM.py:435-447), i.e. M = N for every layer; the device
library accepts M <= 4096, so the low-fidelity set is capped at 4096 points here (config 5 as written,
N_lf = 50 000, would need M = 50 000)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))
import numpy as np
from dgp_dace.models.MF_DGP_EM import MultiFidelityDeepGP_EM

rng = np.random.default_rng(0)
for (n_lf, n_hf, S) in ((200, 50, 100), (1000, 100, 100), (1024, 256, 100), (2048, 512, 100), (4096, 512, 100)):
    X0, X1 = rng.uniform(0, 1, (n_lf, 4)), rng.uniform(0, 1, (n_hf, 2))
    lf = lambda x: np.sin(4 * x[:, :1]) + x[:, 1:2] * x[:, 2:3] - 0.5 * x[:, 3:4]
    X_red = [np.concatenate([X1, 0.5 * np.ones((n_hf, 2))], 1)]
    Y = [lf(X0), 1.3 * lf(X_red[0]) + 0.2 * X1[:, :1]]
    t0 = time.perf_counter()
    mf = MultiFidelityDeepGP_EM([X0, X1], Y, X_red, seed=0)
    mf.model.num_samples = S
    t_build = time.perf_counter() - t0
    data = mf._data()
    mf._initialise(1e-2, 1e-2)
    mf.model.ELBO_and_grad(data)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        val, _ = mf.model.ELBO_and_grad(data)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        mf.model.ELBO(data)
    dtf = (time.perf_counter() - t0) / n
    print(f"N_lf={n_lf} N_hf={n_hf} S={S}: bound+gradient {dt*1e3:.1f} ms ({1/dt:.2f} it/s), bound only {dtf*1e3:.1f} ms, "
          f"construction {t_build*1e3:.0f} ms, ELBO {val:.3f}", flush=True)
