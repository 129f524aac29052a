"""GPU unit tests of single kernels through the C-ABI (dgp_dev_* entry points) against NumPy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dgp_dace import _native
    c = _native.Context(0)
    yield c
    c.close()


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("op", ["NN", "NT", "TN"])
@pytest.mark.parametrize("shape", [(300, 256, 64), (128, 128, 16), (1000, 48, 48), (37, 130, 50), (256, 8, 640),
                                   (513, 3, 77), (64, 272, 272), (64, 64, 64), (37, 50, 61), (5, 64, 33), (64, 3, 64),
                                   (1, 1, 1), (256, 256, 256), (128, 128, 128), (256, 8, 256), (512, 512, 256), (100, 70, 130),
                                   (97, 33, 255), (1024, 16, 250)])
def test_gemm_matches_numpy(ctx, op, shape):
    """MFMA operand/accumulator lane maps, LDS images, tails: asymmetric random operands.  The shapes with M, N, K <= 64
    run in the one-workgroup kernel of gemm_small.hip; those with K <= 256, M, N <= 1024 and few 128 x 64 tiles (the chains'
    Mp x Mp products, the last seven and several of the others) in the 32 x 32-tile kernel of gemm_mid.hip, edges and k tails
    (K % 4 != 0) included; the rest on the 128 x 64 engine."""
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((K, M) if op == "TN" else (M, K))
    B = rng.standard_normal((N, K) if op == "NT" else (K, N))
    ref = (A.T if op == "TN" else A) @ (B.T if op == "NT" else B)
    C = ctx.dev_gemm(op, A, B)
    assert _rel(C, ref) < 1e-13
    C0 = rng.standard_normal((M, N))
    C1 = ctx.dev_gemm(op, A, B, C0=C0, alpha=0.5, beta=1)
    assert _rel(C1, C0 + 0.5 * ref) < 1e-13


@pytest.mark.parametrize("shape", [(256, 128, 20003), (128, 64, 4099), (384, 192, 1000)])
def test_gemm_reduction_over_points_interior_plus_tails(ctx, shape):
    """TN with M % 128 == 0, N % 64 == 0 and K not a multiple of 16: FAST interior kernel + generic K tail,
    with and without XCD-grouped split-K (splits multiple of 8)."""
    M, N, K = shape
    rng = np.random.default_rng(K)
    A = rng.standard_normal((K, M)); B = rng.standard_normal((K, N)); C0 = rng.standard_normal((M, N))
    for splits in (1, 8, 24):
        C = ctx.dev_gemm("TN", A, B, C0=C0, beta=1, splits=splits)
        assert _rel(C, C0 + A.T @ B) < 1e-12, splits


def test_gemm_split_k_atomic_accumulation(ctx):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((20000, 96))
    B = rng.standard_normal((20000, 160))
    C0 = rng.standard_normal((96, 160))
    C = ctx.dev_gemm("TN", A, B, C0=C0, beta=1, splits=13)
    assert _rel(C, C0 + A.T @ B) < 1e-12


@pytest.mark.parametrize("K", [8192, 20000, 70000 - 70000 % 16 + 16])
def test_gram_kernel_reductions_over_the_points(ctx, K):
    """The reductions over the points with a 256 x 256 lower-triangular output (G_d = Ct^T diag(v) Ct, Q = Kbar^T Ct;
    SURVEY App. C step 2) run in the weighted Gram kernel (gemm_gram.h) from 8192 points on: the two-source form
    (A != B) and the one-source form (the same operand twice) against NumPy, added to a pre-filled C; one point short of
    a multiple of 16 the same call takes the 128 x 64 engine - same answer."""
    rng = np.random.default_rng(K)
    A = rng.standard_normal((K, 256)); B = rng.standard_normal((K, 256)); C0 = rng.standard_normal((256, 256))
    lower = np.tril(np.ones((256, 256), dtype=bool))
    for a, b in ((A, B), (A, A)):
        want = C0 + a.T @ b
        for k in (K, K - 1):
            got = ctx.dev_gemm("TN", a[:k], a[:k] if b is a else b[:k], C0=C0, beta=1, splits=8, tri=3, triblk=256)
            ref = C0 + a[:k].T @ (a[:k] if b is a else b[:k])
            assert np.abs(got - ref)[lower].max() <= 1e-12 * np.abs(ref).max(), (k, b is a)
        del want
    # fixed summation order: the kernel has no atomics, two runs agree to the bit
    g1 = ctx.dev_gemm("TN", A, A, C0=C0, beta=1, splits=8, tri=3, triblk=256)
    g2 = ctx.dev_gemm("TN", A, A, C0=C0, beta=1, splits=8, tri=3, triblk=256)
    assert np.array_equal(g1[lower], g2[lower])


@pytest.mark.parametrize("P,D", [(8192, 1), (8192 + 48, 8), (30000 - 30000 % 16, 3), (65536 + 16, 8), (20000, 5), (4096, 8), (10001, 2)])
def test_weighted_gram_kernel_against_numpy(ctx, P, D):
    """G_d += sum_p s[p, d] c_p c_p^T exactly as the backward pass calls it: on the weighted Gram
    kernel for >= 8192 points in multiples of 16, on the 128 x 64 engine otherwise (4096 points; 10 001 points) - lower
    triangles against NumPy, added to a pre-filled G, signed weights."""
    rng = np.random.default_rng(P + D)
    Cm = rng.standard_normal((P, 256)); s = rng.standard_normal((P, D)); G0 = rng.standard_normal((D, 256, 256))
    got = ctx.dev_gram(Cm, s, G0)
    lower = np.tril(np.ones((256, 256), dtype=bool))
    for d in range(D):
        ref = G0[d] + (Cm * s[:, d:d + 1]).T @ Cm
        assert np.abs(got[d] - ref)[lower].max() <= 1e-12 * np.abs(ref).max(), d
    if D == 1:
        g1 = ctx.dev_gram(Cm, None, G0)
        ref = G0[0] + Cm.T @ Cm
        assert np.abs(g1[0] - ref)[lower].max() <= 1e-12 * np.abs(ref).max()


# (P, Mp, D, kernel families expected for Ct / T / dC [/ du / G_d]).  The wide-tile kernel takes a row-panel product from 192
# tiles of 128 x 256 on (csrc/gemm_wide.hip): `Ct` (256 columns) from 24 576 rows, `T` from 24 576 / D rows; the tall-tile kernel
# (Mp = 256) replaces its `T` mode; `dC` at Mp = 256 runs on the row-panel kernel (csrc/gemm_dcpanel.h, from 4096 rows on) from a
# ROW-MAJOR T - the tall kernel then writes T row-major.  98 432 = 769 x 128 rows: the last 256-row tile of T is half empty.
PRODUCTION_CASES = [
    (98304 + 128, 256, 8, ("wide", "tall", "dcpanel", "gram", "gram")),        # (du rides on G_d's Gram launch)
    (98304, 256, 1, ("wide", "tall", "dcpanel", "gram", "gram")),
    (98304 + 384, 256, 3, ("wide", "tall", "dcpanel", "gram", "gram")),
    (300032 + 128, 256, 2, ("wide", "tall", "tallu", "gram", "gram")),         # above the row-panel kernel's window: blocked T, gemm_tallu.h
    (49152 + 128, 512, 2, ("wide", "wide", "wide")),          # BASELINE config 4's inducing count
    (20000, 256, 8, ("engine128x64", "tall", "dcpanel")),     # below the limit of Ct (157 tiles): the 128 x 64 engine, padded rows; T (1256 tiles) is eligible on its own
    (25008, 256, 8, ("wide", "tall", "dcpanel", "gram", "gram")),   # one rank's share of 4: the first layer's shape (196 tiles of Ct)
    (12496, 256, 8, ("engine128x64", "tall", "dcpanel")),     # one rank's share of 8: the first layer's shape (98 panels of dC for 256 workgroups)
    (12496, 256, 1, ("engine128x64", "engine128x64", "dcpanel")),
    (12496 + 37, 256, 5, ("engine128x64", "tall", "dcpanel")),      # a ragged last panel: rows past P read as zeros and are not stored
    (3001, 192, 2, ("engine128x64", "engine128x64", "engine128x64")),
]


@pytest.mark.parametrize("P,Mp,D,expect", PRODUCTION_CASES)
def test_layer_products_on_the_production_kernels_against_numpy(ctx, P, Mp, D, expect):
    """Every element of Ct (+ |c|^2), T (blocked output, + |t_d|^2), mean0, dC (scaled A, the `- c` epilogue term, the
    rank-D term mbar u^T), g = (dC Linv) .* Kt, du and G_d (du inside G_d's Gram launch where that kernel runs) against NumPy, through the launches forward_chunk / backward_chunk
    make (dgp_dev_layer_products), at row counts where the headline configuration's kernels are the ones selected -
    asserted, not assumed.  Reference: R/dgp_dace/utils/layers.py:243-263 (whitened) and its adjoint."""
    from layer_products_check import check
    check(ctx, P, Mp, D, expect)


def test_layer_products_on_the_wide_tile_kernel_modes_against_numpy():
    """The same check with DGP_TALL = DGP_TALLU = 0: `T` (lower form, blocked output + row sums) and `dC` (upper form, scaled
    A, `- c`, rank term) then run as modes of the wide-tile kernel.  The switches are read once per process: child process."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    specs = [f"{P}:{Mp}:{D}:wide,wide,wide,gram,gram" for P, Mp, D, e in PRODUCTION_CASES[:3]]
    env = dict(os.environ, DGP_TALL="0", DGP_TALLU="0", DGP_DCPANEL="0")
    p = subprocess.run([sys.executable, os.path.join(here, "layer_products_check.py")] + specs, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert p.returncode == 0 and p.stdout.count("ok ") == len(specs), p.stdout


def test_layer_products_on_the_blocked_t_tall_tile_pair_against_numpy():
    """With DGP_DCPANEL = 0 the dC product goes back to csrc/gemm_tallu.h and T to the blocked layout that kernel reads: the
    round-2/3 pair stays correct (it is the fallback when the row-panel kernel is switched off).  Child process."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    specs = [f"{P}:{Mp}:{D}:wide,tall,tallu,gram,gram" for P, Mp, D, e in PRODUCTION_CASES[:3]]
    p = subprocess.run([sys.executable, os.path.join(here, "layer_products_check.py")] + specs, env=dict(os.environ, DGP_DCPANEL="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert p.returncode == 0 and p.stdout.count("ok ") == len(specs), p.stdout


@pytest.mark.parametrize("grid", ["256", "248", "232", "200", "72"])
def test_gram_kernel_is_exact_for_any_grid(grid):
    """The Gram kernel's work split keeps the D workgroups that stream the same points on one XCD for ANY workgroup count
    (a collective or the side-stream kernels of the backward pass take CUs away from it): both forms against NumPy at
    grids that are and are not multiples of 8 D.  The grid is read once per process: child process."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "gram_grid_check.py")], env=dict(os.environ, DGP_GRAM_GRID=grid),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.count("ok ") == 5, p.stdout


@pytest.mark.parametrize("Mp", [128, 256, 272])
def test_gemm_triangular_hints_are_exact(ctx, Mp):
    """TRI_* only skip structurally-zero work: results equal the dense product."""
    rng = np.random.default_rng(Mp)
    D, P = 3, 700
    L = np.tril(rng.standard_normal((Mp, Mp)))
    Kt = rng.standard_normal((P, Mp))
    C = ctx.dev_gemm("NT", Kt, L, tri=1, triblk=Mp)                 # Kt @ L^T, L lower
    assert _rel(C, Kt @ L.T) < 1e-13
    C = ctx.dev_gemm("NN", Kt, L, tri=2, triblk=Mp)                 # Kt @ L
    assert _rel(C, Kt @ L) < 1e-13
    W = [np.tril(rng.standard_normal((Mp, Mp))) for _ in range(D)]
    Wcat = np.concatenate(W, axis=1)                                 # [Mp, D*Mp]
    T = ctx.dev_gemm("NN", Kt, Wcat, tri=2, triblk=Mp)
    assert _rel(T, Kt @ Wcat) < 1e-13
    Cb = ctx.dev_gemm("NT", T, Wcat, tri=1, triblk=Mp)               # sum_d T_d W_d^T
    assert _rel(Cb, T @ Wcat.T) < 1e-13
    G = ctx.dev_gemm("TN", Kt, T, tri=3, triblk=Mp, beta=1, splits=3)
    ref = Kt.T @ T
    for d in range(D):                                               # only the lower triangles are defined
        blk = slice(d * Mp, (d + 1) * Mp)
        assert _rel(np.tril(G[:, blk]), np.tril(ref[:, blk])) < 1e-12


@pytest.mark.parametrize("M", [16, 48, 256, 512])
def test_cholesky_and_triangular_inverse(ctx, M):
    rng = np.random.default_rng(M)
    B = 3
    A = rng.standard_normal((B, M, M))
    A = A @ A.transpose(0, 2, 1) + M * np.eye(M)[None]
    L = ctx.dev_chol(A)
    ref = np.linalg.cholesky(A)
    assert _rel(L, ref) < 1e-12
    assert np.all(np.triu(L, 1) == 0)
    X = ctx.dev_trinv(ref)
    assert _rel(X @ ref, np.tile(np.eye(M)[None], [B, 1, 1])) < 1e-11
    assert np.all(np.triu(X, 1) == 0)


def test_cholesky_reports_non_positive_definite(ctx):
    from dgp_dace._native import NotPositiveDefinite
    A = np.eye(32)
    A[5, 5] = -1.0
    with pytest.raises(NotPositiveDefinite):
        ctx.dev_chol(A)


def test_philox_normals_match_oracle(ctx):
    import dgp_oracle as O
    z = ctx.dev_normals(seed=0x1234567890ABCDEF, layer=2, S=3, n0=10 ** 10, N=257, D=5)
    ref = O.philox_normal(0x1234567890ABCDEF, 2, 3, np.arange(10 ** 10, 10 ** 10 + 257), 5)
    np.testing.assert_allclose(z, ref, rtol=1e-13, atol=1e-13)


def test_fp64_mfma_issue_rate(ctx):
    """v_mfma_f64_16x16x4_f64 micro-benchmark: the roofline denominator (spec 78.6 TFLOP/s)."""
    t = ctx.dev_mfma_peak(20000)
    print("fp64 MFMA issue rate: %.1f TFLOP/s" % t)
    assert 30.0 < t < 120.0


def test_half_row_launches_forced_at_small_sizes():
    """The two-part launch of lower-triangular outputs (full tiles + half-row tiles on the ROWSEL instantiation) only
    triggers by itself at production sizes; DGP_HALF_MIN_WG=1 forces it for every such product, and the gradient /
    natural-gradient parity tests must still hold (run in a child process: the switch is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DGP_HALF_MIN_WG="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-k",
                        "gradient_matches_autograd_golden or natural_gradient_step_matches_golden or medium_size"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("P,Mp,w1,fused", [(101760, 256, 9, True), (12496, 256, 9, True), (2048 + 5, 256, 2, True), (40000, 256, 4, True),
                                           (2000, 256, 9, False), (30000, 512, 9, False), (5000, 256, 12, False)])
def test_rbf_backward_contractions_against_numpy(ctx, P, Mp, w1, fused):
    """R1 = g [Z | 1] (every row: the x-gradient of a stationary kernel) and GX += g^T [X | 1] (the Z / lengthscale gradients),
    through the launcher the backward pass uses, against NumPy: the one-pass matrix-core kernel of points.hip where it applies
    (asserted), the two engine products elsewhere; a ragged last 16-point tile and a pre-filled GX included.
    Reference: what tf.GradientTape derives through gpflow's K(Z, X) (R/dgp_dace/utils/layers.py:243, R/dgp_dace/models/dgp.py:272-275)."""
    rng = np.random.default_rng(P + Mp + w1)
    G = rng.standard_normal((P, Mp)); Z1 = rng.standard_normal((Mp, w1)); X1 = rng.standard_normal((P, w1))
    GX0 = rng.standard_normal((Mp, w1))
    R1, GX, f = ctx.dev_rbf_contract(G, Z1, X1, GX0)
    assert f == fused
    ref = G @ Z1
    assert np.abs(R1 - ref).max() <= 1e-12 * np.abs(ref).max()
    ref = GX0 + G.T @ X1
    assert np.abs(GX - ref).max() <= 1e-12 * np.abs(ref).max()
    if fused:                                                            # no atomics there: two runs agree to the bit
        R1b, GXb, _ = ctx.dev_rbf_contract(G, Z1, X1, GX0)
        assert np.array_equal(R1, R1b) and np.array_equal(GX, GXb)


@pytest.mark.parametrize("rows", [12544, 8192, 32768])
def test_gemm_triangular_row_panels_at_shard_sizes(ctx, rows):
    """Row-panel products with a triangular 256-column B at the row counts of a rank's share of 8 GPUs / the deduplicated first
    layer (below the wide- and tall-tile kernels' limits: the 128 x 64 engine with its triangular k-range skips): both triangle
    orientations, one and several k blocks, against NumPy."""
    rng = np.random.default_rng(rows)
    L = np.tril(rng.standard_normal((256, 256)))
    Kt = rng.standard_normal((rows, 256))
    assert _rel(ctx.dev_gemm("NT", Kt, L, tri=1, triblk=256), Kt @ L.T) < 1e-13          # B = L^T upper
    assert _rel(ctx.dev_gemm("NN", Kt, L, tri=2, triblk=256), Kt @ L) < 1e-13            # B lower
    W = [np.tril(rng.standard_normal((256, 256))) for _ in range(3)]
    T = rng.standard_normal((rows, 3 * 256))
    Wcat = np.concatenate(W, axis=1)
    assert _rel(ctx.dev_gemm("NT", T, Wcat, tri=1, triblk=256), T @ Wcat.T) < 1e-13      # sum_d T_d W_d^T: K = 3 blocks


@pytest.mark.parametrize("white", [False, True])
def test_layer_api_conditionals_samples_and_kl_against_the_oracle_layer(white):
    """The per-layer surface of the reference (`SVGP_Layer.conditional_ND / conditional_SND / sample_from_conditional / KL`,
    layers.py:63-130,237-308) evaluated through a one-layer device context, against `OracleLayer` in the same state;
    `sample_from_conditional(full_cov=True)` must draw through chol(var + jitter I) per (sample, output) (utils.py:43-51)."""
    import dgp_oracle as O
    from dgp_dace.gpflow_compat import RBF, Identity
    from dgp_dace.utils.layers import SVGP_Layer
    rng = np.random.default_rng(12)
    M, Din, Dout, S, N = 20, 3, 3, 2, 17
    Z = rng.standard_normal((M, Din))
    ls, var = np.array([0.9, 1.2, 0.7]), 1.4
    lay = SVGP_Layer(RBF(var, ls), Z, Dout, Identity(), white=white)
    ora = O.OracleLayer(O.RBF(var, ls), Z, Dout, O.MeanFunction("identity"), white=white)
    q_mu = 0.5 * rng.standard_normal((M, Dout))
    q_sqrt = np.tril(0.3 * np.asarray(lay.q_sqrt.numpy()) + 0.05 * rng.standard_normal((Dout, M, M)))
    lay.q_mu.assign(q_mu); lay.q_sqrt.assign(q_sqrt)
    ora.q_mu, ora.q_sqrt = q_mu.copy(), q_sqrt.copy()
    X = rng.standard_normal((S, N, Din))
    z = rng.standard_normal((S, N, Dout))
    for full_cov in (False, True):
        f, mean, v = lay.sample_from_conditional(X, z=z, full_cov=full_cov)
        fo, mo, vo = ora.sample_from_conditional(X, z, full_cov=full_cov)
        assert np.asarray(v).shape == ((S, N, N, Dout) if full_cov else (S, N, Dout))
        np.testing.assert_allclose(np.asarray(mean), mo, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(np.asarray(v), vo, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(np.asarray(f), fo, rtol=1e-7, atol=1e-8)
    # the full-covariance draw is not the diagonal formula (what this method computed before round 4)
    mo, vo = ora.conditional_SND(X, full_cov=True)
    diag_form = mo + z * (np.einsum("snnd->snd", vo) + 1e-6) ** 0.5
    f, _, _ = lay.sample_from_conditional(X, z=z, full_cov=True)
    assert np.abs(np.asarray(f) - diag_form).max() > 1e-3
    m1, v1 = lay.conditional_ND(X[0])
    mo1, vo1 = ora.conditional_ND(X[0])
    np.testing.assert_allclose(np.asarray(m1), mo1, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(np.asarray(v1), vo1, rtol=1e-8, atol=1e-9)
    assert abs(lay.KL() - ora.KL()) < 1e-8 * max(1.0, abs(ora.KL()))
    assert np.asarray(lay.sample_from_conditional(X)[0]).shape == (S, N, Dout)      # z drawn when not supplied
    with pytest.raises(ValueError):
        lay.sample_from_conditional(X, z=z[:, :-1])


def test_one_rocm_stack_in_the_normal_load_order(ctx):
    """torch first, then the library (what dgp_dace._native.load does): exactly one libamdhip64 is mapped, also after
    torch.cuda has been initialised, so dgp_comm_init's refusal of a two-stack process stays silent (the one-rank
    communicator tests of test_gpu_parity.py go through it)."""
    import torch
    from dgp_dace import _native
    assert torch.cuda.is_available() and torch.zeros(1, device="cuda").item() == 0.0
    paths = _native.check_one_hip_runtime()
    assert len(paths) == 1, paths
    assert paths == _native.hip_runtimes_in(open("/proc/self/maps"))


@pytest.mark.parametrize("P,w1", [(49280, 9), (12496, 9), (100001, 9), (4224, 2), (20000, 5), (4097, 3), (8200, 4), (6000, 6),
                                  (33000, 7), (5000, 8)])
def test_g_panel_kernel_every_element_against_numpy(ctx, P, w1):
    """csrc/gemm_gpanel.h, the launch that replaces `g = (Cbar Linv) .* k`, `g [Z | 1]` and `g^T [X | 1]` of the backward pass
    through Kuf (what tf.GradientTape derives for layers.py:243 under dgp.py:272-275) at 256 inducing
    points: every element of R1 and GX against NumPy.  Sizes: several panels per workgroup; a rank's share of 8 GPUs (98 panels
    for 256 workgroups); a ragged last panel (rows past P must read as zeros, R1 must not be written behind P); the smallest
    and a middle input width."""
    rng = np.random.default_rng(P + w1)
    Cbar = rng.standard_normal((P, 256))
    E = rng.uniform(0.1, 1.0, (P, 256))
    Linv = np.tril(rng.standard_normal((256, 256))) / 16.0
    Z1 = rng.standard_normal((256, w1))
    X1 = rng.standard_normal((P, w1))
    GX0 = rng.standard_normal((256, w1))
    R1, GX, used = ctx.dev_g_panel(Cbar, Linv, E, Z1, X1, GX0)
    assert used
    g = (Cbar @ Linv) * E
    R1_ref, GX_ref = g @ Z1, GX0 + g.T @ X1
    assert np.abs(R1 - R1_ref).max() < 1e-12 * np.abs(R1_ref).max()
    assert np.abs(GX - GX_ref).max() < 1e-12 * np.abs(GX_ref).max()
    # reproducible: no atomics anywhere in the launch
    R1b, GXb, _ = ctx.dev_g_panel(Cbar, Linv, E, Z1, X1, GX0)
    assert np.array_equal(R1, R1b) and np.array_equal(GX, GXb)


def test_g_panel_declines_small_inputs(ctx):
    rng = np.random.default_rng(0)
    P = 1000
    _, _, used = ctx.dev_g_panel(rng.standard_normal((P, 256)), np.eye(256), np.ones((P, 256)), np.ones((256, 3)), np.ones((P, 3)))
    assert not used


def test_profiling_ring_overflow_is_drained_not_overrun():
    """dgp_prof_*: the ring of HIP-event pairs (8192) is folded into the per-category sums when it fills up.  Round 4 found that
    `bench.py --steps 300` ended in a segmentation fault: with the ring full, a pair still running on a side stream made the fold
    return early, the ring stayed full and the next scope wrote behind it.  A three-layer model with Mp = 128 (launch-by-launch
    chains on side streams, ~50 scopes per iteration) for enough iterations to fill the ring twice."""
    from dgp_dace.gpflow_compat import RBF, Gaussian
    from dgp_dace.models.dgp import DGP
    rng = np.random.default_rng(0)
    N, D, M = 600, 2, 100
    X = rng.standard_normal((N, D)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((N, 1))
    m = DGP(X, Y, X[:M].copy(), [RBF(1.0, np.ones(D)) for _ in range(3)], [2, 2], Gaussian(), num_samples=2)
    ctx = m._sync_model()
    ctx.adam_reset()
    flags = m._trainable_flags()
    ctx.prof_enable(True)

    def run(n):
        for _ in range(n):
            c = m._grad_step(m.data)
            c.adam_step(0.01, 0.9, 0.999, 1e-7, flags)
        ctx.sync()
        return ctx.prof_read()                  # (reading folds the ring and resets the sums)

    per_it = sum(v["launches"] for v in run(50).values()) / 50.0
    assert per_it > 5
    prof = run(int((2 * 8192 + 1000) / per_it) + 1)
    ctx.prof_enable(False)
    m._device_newer = True
    assert sum(v["launches"] for v in prof.values()) > 2 * 8192
    assert all(np.isfinite(v["ms"]) and v["ms"] >= 0 for v in prof.values()) and prof["small_matrix_chain"]["ms"] > 0
    assert np.isfinite(ctx.last_elbo())


def test_profiling_of_one_category_leaves_the_others_without_events():
    """dgp_prof_enable(0x100 | mask): only the selected categories' launches sit between event pairs (bench.py's timed region asks
    for the contractions alone); the others still report launches and algorithmic flops, with 0 ms - and the same launch counts
    and flops as a run that times everything."""
    from dgp_dace.gpflow_compat import RBF, Gaussian
    from dgp_dace.models.dgp import DGP
    rng = np.random.default_rng(1)
    N, D, M = 600, 2, 100
    X = rng.standard_normal((N, D)); Y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((N, 1))
    m = DGP(X, Y, X[:M].copy(), [RBF(1.0, np.ones(D)) for _ in range(3)], [2, 2], Gaussian(), num_samples=2)
    ctx = m._sync_model()
    ctx.adam_reset()
    flags = m._trainable_flags()

    def run(**kw):
        ctx.prof_enable(True, **kw)
        for _ in range(3):
            c = m._grad_step(m.data)
            c.adam_step(0.01, 0.9, 0.999, 1e-7, flags)
        ctx.sync()
        out = ctx.prof_read()
        ctx.prof_enable(False)
        return out

    every = run()
    one = run(categories=["mfma_contractions"])
    m._device_newer = True
    assert one["mfma_contractions"]["ms"] > 0 and one["mfma_contractions"]["launches"] == every["mfma_contractions"]["launches"] > 0
    for k in ("per_point_streaming", "small_matrix_chain", "adam"):
        assert every[k]["ms"] > 0 and one[k]["ms"] == 0.0, k
        assert one[k]["alg_flops"] == every[k]["alg_flops"] and one[k]["alg_bytes"] == every[k]["alg_bytes"], k
        assert one[k]["launches"] > 0, k
