"""CPU tests of the host-side package (no GPU): API surface of the reference, parameter packing,
layer construction, the C-ABI library (loads, exports every declared symbol, refuses to compute
without a device), and the multi-process sharding logic over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import dgp_oracle as O
from helpers import notebook_data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(white=False, num_units=(1, 1), D=1, Dy=1, N=50, M=25, seed=0):
    from dgp_dace.gpflow_compat import RBF, Gaussian
    from dgp_dace.models.dgp import DGP
    if D == 1 and Dy == 1 and N == 50:
        X, Y, Z = notebook_data()
    else:
        rng = np.random.default_rng(seed)
        X, Y, Z = rng.standard_normal((N, D)), rng.standard_normal((N, Dy)), rng.standard_normal((M, D))
    dims = [D] + list(num_units)
    return DGP(X, Y, Z, [RBF(lengthscales=[1.0] * d, variance=1.0) for d in dims], num_units=list(num_units),
               likelihood=Gaussian(), num_samples=10, white=white), (X, Y, Z)


def test_constructor_prints_architecture_and_counts_parameters(capsys):
    """nb_DGP_regression cells 18 and 30: the printed architecture and number_parameters == 2032."""
    m, _ = _model()
    out = capsys.readouterr().out
    assert "The DGP architecture" in out and "layer 1 : dim_in 1 --> dim_out 1" in out and "layer 2 : dim_in 1 --> dim_out 1" in out
    assert m.number_parameters(trainable=False) == 2032
    assert m.number_parameters(trainable=True) == 2032
    assert m.name == "dgp" and m.num_samples == 10 and len(m.layers) == 3
    assert m.likelihood.likelihood.variance.numpy() == 1.0


def test_layer_state_matches_oracle_construction():
    """SVGP_Layer.__init__ (layers.py:181-224): q_mu = 0, q_sqrt = chol(K(Z) + 1e-6 I) tiled (non-white) / I (white);
    init_layers_linear (layer_initializations.py:24-68): identity / PCA / zero-pad mean functions."""
    rng = np.random.default_rng(1)
    X, Y, Z = rng.standard_normal((40, 3)), rng.standard_normal((40, 2)), rng.standard_normal((7, 3))
    from dgp_dace.gpflow_compat import RBF, Gaussian
    from dgp_dace.models.dgp import DGP
    for white in (False, True):
        m = DGP(X, Y, Z, [RBF(1.0, np.ones(d)) for d in (3, 2, 4)], [2, 4], Gaussian(), white=white)
        mo = O.OracleDGP(X, Y, Z, [O.RBF(1.0, np.ones(d)) for d in (3, 2, 4)], [2, 4], white=white)
        assert [l.mean_function.kind for l in m.layers] == ["linear", "linear", "zero"]
        for l, lo in zip(m.layers, mo.layers):
            np.testing.assert_allclose(l.feature.Z.numpy(), lo.Z, rtol=1e-13)
            np.testing.assert_allclose(l.q_sqrt.numpy(), lo.q_sqrt, rtol=1e-12, atol=1e-14)
            np.testing.assert_array_equal(l.q_mu.numpy(), lo.q_mu)
            if l.mean_function.kind == "linear":
                np.testing.assert_allclose(l.mean_function.A.numpy(), lo.mean_function.A, rtol=1e-13)
        # Linear mean functions are fixed: A and b are parameters but not trainable
        assert m.number_parameters(trainable=False) - m.number_parameters(trainable=True) == (3 * 2 + 1) + (2 * 4 + 1)


def test_parameter_surface_assign_numpy_set_trainable():
    """dgp.py:268-269 (`layer.q_sqrt.assign(layer.q_sqrt * 1e-3)`) and dgp.py:316-322 (set_trainable)."""
    from dgp_dace.gpflow_compat import set_trainable
    m, _ = _model()
    q0 = m.layers[0].q_sqrt.numpy()
    m.layers[0].q_sqrt.assign(m.layers[0].q_sqrt * 1e-3)
    np.testing.assert_allclose(m.layers[0].q_sqrt.numpy(), q0 * 1e-3)
    assert np.all(np.triu(m.layers[0].q_sqrt.numpy()[0], 1) == 0)
    set_trainable(m.layers[-1].q_mu, False)
    flags = m._trainable_flags()
    assert flags == [True] * 13 + [False, True, True]
    mask = m._natgrad_setup(ng_all=False)
    assert mask == [False, False, True] and not m.layers[-1].q_sqrt.trainable and m.layers[0].q_sqrt.trainable
    # flat packing order of dgp_model_set: per layer Z, variance, lengthscales, q_mu, q_sqrt; then likelihood variance
    flat = m._flat()
    assert flat.size == 2032 and flat[-1] == 1.0
    np.testing.assert_array_equal(flat[:25], m.layers[0].feature.Z.numpy().ravel())


def test_only_the_accelerated_subset_is_accepted():
    from dgp_dace.gpflow_compat import RBF, Gaussian, kernel_from_any
    from dgp_dace.models.dgp import DGP

    class Matern32:                      # a foreign (gpflow-like) object is read by attribute
        variance, lengthscales = 2.0, np.ones(1) * 0.5

    k = kernel_from_any(Matern32(), 3)
    assert k.kind == "matern32" and k.lengthscales.shape == (3,) and float(k.variance.numpy()) == 2.0

    class Sum:                           # composite kernels (MF-DGP) are not on this path
        kernels = []

    with pytest.raises(NotImplementedError):
        kernel_from_any(Sum(), 1)
    X, Y, Z = notebook_data()
    with pytest.raises(Exception):
        DGP(X, Y, Z, [RBF(1.0, [1.0])] * 2, [1, 1], Gaussian())          # one kernel per layer is required
    from dgp_dace.utils.layers import SVGP_Layer
    with pytest.raises(NotImplementedError):                    # input propagation is unused by DGP and not offered
        SVGP_Layer(RBF(1.0, [1.0]), Z, 1, None, input_prop_dim=1)


def test_library_exports_every_symbol_declared_in_the_header():
    from dgp_dace import _native
    header = open(os.path.join(ROOT, "include", "dgp_abi.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|void|int64_t|const char\*)\s+(dgp_[a-z_0-9]+)\s*\(", header, re.M)))
    assert declared == sorted(_native.SYMBOLS)
    lib = _native.load()          # (torch first, then the library: a bare CDLL here would bring the system ROCm stack in
    for s in declared:            #  ahead of torch's - exactly what test_two_rocm_stacks_in_one_process_are_detected refuses)
        assert hasattr(lib, s), s


def test_two_rocm_stacks_in_one_process_are_detected(monkeypatch):
    """NOTES 12.3 / VERDICT r3 weak #4: torch's wheel ships a libamdhip64 under the system ROCm's soname; a process holding
    both hands one stack's streams to the other's RCCL.  The detector on a fake map list, and on this process."""
    from dgp_dace import _native
    one = ["7f00-7f10 r-xp 0 00:00 1  /usr/lib/python3/site-packages/torch/lib/libamdhip64.so",
           "7f20-7f30 r--p 0 00:00 1  /usr/lib/python3/site-packages/torch/lib/libamdhip64.so",
           "7f40-7f50 r-xp 0 00:00 2  /usr/lib/python3/site-packages/torch/lib/librccl.so",
           "7f60-7f70 rw-p 0 00:00 0  [heap]", ""]
    assert _native.hip_runtimes_in(one) == ["/usr/lib/python3/site-packages/torch/lib/libamdhip64.so"]
    assert _native.check_one_hip_runtime(one) == _native.hip_runtimes_in(one)
    two = one + ["7f80-7f90 r-xp 0 00:00 3  /opt/rocm-7.2.0/lib/libamdhip64.so.7.2.70200"]
    assert len(_native.hip_runtimes_in(two)) == 2
    with pytest.raises(_native.NativeUnavailable) as e:
        _native.check_one_hip_runtime(two)
    assert "torch/lib/libamdhip64.so" in str(e.value) and "/opt/rocm-7.2.0/lib/libamdhip64.so.7.2.70200" in str(e.value)
    monkeypatch.setenv("DGP_ALLOW_TWO_RUNTIMES", "1")
    with pytest.warns(RuntimeWarning):
        assert len(_native.check_one_hip_runtime(two)) == 2
    monkeypatch.delenv("DGP_ALLOW_TWO_RUNTIMES")
    # this process, through the library's own scan (dl_iterate_phdr): the binding imported torch first -> one stack,
    # and it agrees with /proc/self/maps
    lib = _native.load()
    buf = ctypes.create_string_buffer(4096)
    n = lib.dgp_hip_runtimes(buf, len(buf))
    assert n == 1, buf.value
    assert _native.hip_runtimes_in(buf.value.decode().split("\n")) == _native.hip_runtimes_in(open("/proc/self/maps"))
    assert _native.check_one_hip_runtime() == _native.hip_runtimes_in(open("/proc/self/maps"))


def test_no_cpu_fallback_without_a_device():
    """On a machine without a GPU the hot path must fail loudly (it never routes through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from dgp_dace._native import NativeUnavailable
    m, (X, Y, Z) = _model()
    for call in (lambda: m.ELBO(), lambda: m.predict(X, 5), lambda: m.propagate(X, S=2),
                 lambda: m.optimize_adam(iterations=1)):
        with pytest.raises(NativeUnavailable):
            call()
    src = open(os.path.join(ROOT, "dgp-toolbox_amd", "dgp_dace", "models", "dgp.py")).read()
    assert "oracle" not in src and "torch" not in src.replace("torch.distributed", "")


def test_shard_bounds_partition_the_points():
    from dgp_dace.parallel import shard_bounds
    for N in (1, 7, 100_000, 1_000_003):
        for W in (1, 2, 3, 8):
            b = [shard_bounds(N, r, W) for r in range(W)]
            assert b[0][0] == 0 and b[-1][1] == N
            assert all(b[i][1] == b[i + 1][0] for i in range(W - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= (1 if N < 32 * W else 17) and min(sizes) >= (1 if N >= W else 0)
            if N >= 32 * W:
                assert all(lo % 16 == 0 for lo, _ in b)       # aligned interior boundaries: no ragged reduction tail


WORKER = r'''
import os, sys
sys.path[:0] = [os.path.join(ROOT, "dgp-toolbox_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % PORT, rank=RANK, world_size=2)
from dgp_dace import parallel
import dgp_oracle as O, dgp_oracle_torch as T
from helpers import load, oracle_from_golden, n_layers
d = parallel.current()
assert d is not None and d.world == 2 and d.rank == RANK and not d.on_gpu
g = load("case_B_nonwhite")
m = oracle_from_golden(g)
X, Y = g["X"], g["Y"]
S, N = int(g["S"]), X.shape[0]
lo, hi = d.shard(N)
# every rank evaluates ITS data points with normals keyed by the GLOBAL point index, then one all-reduce(sum)
zs = O.draw_zs(m, 99, S, hi - lo, n_offset=lo)
P = T.params_from_model(m)
L = T.data_term(m, P, torch.as_tensor(X[lo:hi]), torch.as_tensor(Y[lo:hi]), [torch.as_tensor(z) for z in zs], S)
L.backward()
flat = torch.cat([L.detach().reshape(1)] + [p[k].grad.reshape(-1) for p in P["layers"] for k in ("Z", "variance", "lengthscales", "q_mu", "q_sqrt")])
buf = d.device_buffer(flat.numel(), 0)
buf.copy_(flat)
d.all_reduce_(buf)
if RANK == 0:
    zs_full = O.draw_zs(m, 99, S, N)
    P2 = T.params_from_model(m)
    L2 = T.data_term(m, P2, torch.as_tensor(X), torch.as_tensor(Y), [torch.as_tensor(z) for z in zs_full], S)
    L2.backward()
    ref = torch.cat([L2.detach().reshape(1)] + [p[k].grad.reshape(-1) for p in P2["layers"] for k in ("Z", "variance", "lengthscales", "q_mu", "q_sqrt")])
    err = float((buf - ref).abs().max() / ref.abs().max())
    assert err < 1e-12, err
    assert abs(d.all_reduce_scalar(1.5) - 3.0) < 1e-15
    print("SHARDED_OK", err)
else:
    d.all_reduce_scalar(1.5)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding_reproduces_the_full_batch_sums(tmp_path):
    """World size 2 over gloo: per-rank partial sums of the ELBO data term and of every gradient over the
    rank's data points (normals keyed by the global point index) + one all-reduce == full batch."""
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for rank in (0, 1):
        code = f"ROOT={ROOT!r}\nPORT={port}\nRANK={rank}\n" + WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "SHARDED_OK" in outs[0]


def test_infill_host_math_without_a_device():
    """Acquisition-side host logic (dgp_dace/Infill_criteria.py): EI partials, moment cotangents, DE restatement."""
    from scipy.stats import norm
    from dgp_dace import Infill_criteria as IC
    rng = np.random.default_rng(0)
    mean, var, y_min = rng.standard_normal((5, 1)), rng.uniform(0.2, 2.0, (5, 1)), 0.3
    ei, d_mean, d_var = IC._ei(y_min, mean, var)
    sd = np.sqrt(var)
    # the reference's expression: (y_min - mu) cdf + var * N(mu, sd).prob(y_min)   (Infill_criteria.py:43-47)
    want = (y_min - mean) * norm.cdf(y_min, mean, sd) + var * norm.pdf(y_min, mean, sd)
    np.testing.assert_allclose(ei, want, rtol=1e-13)
    h = 1e-6
    np.testing.assert_allclose(d_mean, (IC._ei(y_min, mean + h, var)[0] - IC._ei(y_min, mean - h, var)[0]) / (2 * h), rtol=1e-7)
    np.testing.assert_allclose(d_var, (IC._ei(y_min, mean, var + h)[0] - IC._ei(y_min, mean, var - h)[0]) / (2 * h), rtol=1e-7)
    # cotangents of the per-sample moments: directional finite difference of sum(a*mean + b*var)
    Fm, Fv = rng.standard_normal((7, 5, 1)), rng.uniform(0.1, 1.0, (7, 5, 1))
    a, b = rng.standard_normal((5, 1)), rng.standard_normal((5, 1))
    obj = lambda Fm, Fv: float(sum((c * m).sum() for c, m in zip((a, b), IC._moments(Fm, Fv))))
    mb, vb = IC._moment_cotangents(Fm, Fm.mean(0), a, b)
    dm, dv = rng.standard_normal(Fm.shape), rng.standard_normal(Fv.shape)
    fd = (obj(Fm + h * dm, Fv + h * dv) - obj(Fm - h * dm, Fv - h * dv)) / (2 * h)
    assert abs(fd - float((mb * dm).sum() + (vb * dv).sum())) < 1e-7 * max(1.0, abs(fd))
    # differential evolution finds the minimum of a shifted quadratic, evaluating whole populations per call
    calls = []
    def f(U):
        calls.append(U.shape)
        return ((U - np.array([1.0, -2.0])) ** 2).sum(1)
    u = IC._differential_evolution(f, np.zeros(2), 1.5, 40, 120, np.random.default_rng(1))
    np.testing.assert_allclose(u, [1.0, -2.0], atol=1e-3)
    assert all(c == (40, 2) for c in calls)
    # unknown model kinds are refused; an exact GP without an input gradient has no Adam branch
    class Other: name = 'svgp'
    with pytest.raises(NotImplementedError):
        IC.EI(0.0, 2).run(Other(), np.zeros((1, 2)))
    class Gpr: name = 'gpr'
    with pytest.raises(NotImplementedError):
        IC.EI(0.0, 2).optimize(Gpr(), (np.zeros(2), np.ones(2)), method='Adam')


def test_bench_flop_accounting_matches_the_survey_tables():
    """bench.py's roofline numerators: SURVEY §8d's own count reproduces its table (config 2: 4.631e12, 2-alt:
    2.620e12); the executed-formulation count credits the first layer once, one triangular solve and
    the three backward products that are passes over the points (Q' is assembled from the layer's other sums)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert abs(bench.survey_flops_step(100_000, 10, [8, 8, 8], 256, 1) / 4.631e12 - 1) < 1e-3
    assert abs(bench.survey_flops_step(100_000, 10, [8, 8], 256, 1) / 2.620e12 - 1) < 1e-3
    assert abs(bench.survey_flops_step(1_000_000, 10, [16, 16, 16, 16], 512, 1) / 4.542e14 - 1) < 1e-3
    ours = bench.alg_flops_step(100_000, 10, [8, 8, 8], 256, 1)
    want = 256 * 257 * (100_000 * 26.0 + 1_000_000 * 26.0 + 1_000_000 * 5.0)      # (3 D_out + 2) M (M + 1) per point and layer
    assert abs(ours / want - 1) < 1e-12 and ours < bench.survey_flops_step(100_000, 10, [8, 8, 8], 256, 1)
    X, Y, Z = bench.synthetic(1000, 3, 16)
    assert X.shape == (1000, 3) and Y.shape == (1000, 1) and Z.shape == (16, 3)
    np.testing.assert_allclose(X.mean(0), 0, atol=1e-12)
    np.testing.assert_allclose(Y.std(0), 1, rtol=1e-12)


def test_expected_violation_host_math():
    """EV_one_constraint / EV (Infill_criteria.py:234-316): the reference's expression, its partials, and the
    combination rule with an unconstrained criterion, on stub models (no device)."""
    from scipy.stats import norm
    from dgp_dace import Infill_criteria as IC
    from dgp_dace.gpflow_compat import as_tensor
    rng = np.random.default_rng(4)
    mean, var, c = rng.standard_normal((6, 1)), rng.uniform(0.2, 1.5, (6, 1)), -0.3
    ev, d_mean, d_var = IC._ev(c, mean, var)
    sd = np.sqrt(var)
    want = (-c + mean) * norm.cdf(-c, -mean, sd) + var * norm.pdf(-c, -mean, sd)        # Infill_criteria.py:246-248
    np.testing.assert_allclose(ev, want, rtol=1e-13)
    h = 1e-6
    np.testing.assert_allclose(d_mean, (IC._ev(c, mean + h, var)[0] - IC._ev(c, mean - h, var)[0]) / (2 * h), rtol=1e-7)
    np.testing.assert_allclose(d_var, (IC._ev(c, mean, var + h)[0] - IC._ev(c, mean, var - h)[0]) / (2 * h), rtol=1e-7)

    class Stub:                                   # an exact-GP-like model with fixed predictions
        name = 'gpr'
        def __init__(self, m, v): self.m, self.v = m, v
        def predict_y(self, x): return as_tensor(self.m[:len(x)]), as_tensor(self.v[:len(x)])
    x = np.zeros((6, 2))
    mc = [Stub(mean, var), Stub(mean + 1.0, var)]
    e = IC.EV([c, c], 2)
    evs = np.asarray(e.run(mc, x))
    np.testing.assert_allclose(evs[:, :1], ev, rtol=1e-13)
    my = Stub(rng.standard_normal((6, 1)), rng.uniform(0.2, 1.0, (6, 1)))
    ei = IC.EI(0.1, 2)
    got = np.asarray(e.run_with_IC(ei, my, mc, x, threshold=0.9))
    bad = evs.max(1, keepdims=True) > 0.9
    np.testing.assert_allclose(got, np.where(bad, evs.sum(1, keepdims=True) + 10000.0, np.asarray(ei.run(my, x))), rtol=1e-13)
    assert bad.any() and (~bad).any()


def test_so_bo_host_logic_without_a_device():
    """Constructor, normalisation, feasibility bookkeeping and model factory of the BO driver mirror
    (dgp_dace/BO/SO_BO.py against R/dgp_dace/BO/SO_BO.py:27-50,100-176,177-250); nothing here touches the GPU."""
    from dgp_dace.BO.SO_BO import SO_BO, denormalize, normalize, normalize_C, normalize_X

    class Problem:                                        # nb_dgp_BO.ipynb cells 4-6
        constraint, dim = True, 1
        def fun(self, x):
            return [(x - 0.5) ** 2, np.where(x > 0.25, 1.0, 0.0)]

    X = np.array([[0.05], [0.2], [0.4], [0.7], [0.9]])
    Y, C = Problem().fun(X)
    dgp_dic = {'num_layers': 2, 'num_units': 1, 'kernels': ['rbf', 'matern32', 'matern52'], 'num_samples': 10}
    bo = SO_BO(Problem(), X=X, Y=Y, C=C, model_Y_dic={'num_layers': 0, 'kernels': 'rbf'}, model_C_dic=dgp_dic)
    Xn, lw, up = normalize_X(X)
    np.testing.assert_allclose(bo.X_train, Xn)
    np.testing.assert_allclose(denormalize(bo.X_train, X), X, rtol=1e-13)
    np.testing.assert_allclose([bo.lw_n, bo.up_n], [lw, up])
    np.testing.assert_allclose(bo.feasible_0, normalize_C(C)[1])
    np.testing.assert_allclose(bo.Y_train, normalize(Y))
    assert bo.model_Y.name == 'gpr' and bo.model_C[0].name == 'dgp' and len(bo.model_C[0].layers) == 3
    assert [l.kern.kind for l in bo.model_C[0].layers] == ['rbf', 'matern32', 'matern52']
    assert bo.model_C[0].layers[0].feature.Z.shape == (5, 1)                        # Z = X (SO_BO.py:248)
    np.testing.assert_allclose(bo.Xfeasible, [0.05, 0.2])                           # constraint <= 0 only below 0.25
    assert np.isclose(bo.Ymin[-1], (0.2 - 0.5) ** 2)
    with pytest.raises(Exception):
        SO_BO(Problem(), X=X, Y=Y, C=C, model_Y_dic=None, model_C_dic=dgp_dic)
    with pytest.raises(Exception):
        bo.make_model({'num_layers': 2, 'num_units': [1], 'kernels': 'rbf', 'num_samples': 5}, Xn, normalize(Y))
    # an infeasible proposal leaves the best value unchanged, a feasible better one updates it
    bo.added_points = (np.array([[0.8]]) - X.mean(0)) / X.std(0)
    bo.add_point()
    assert bo.X.shape == (6, 1) and np.isclose(bo.X[-1, 0], 0.8) and np.isclose(bo.Ymin[-1], bo.Ymin[-2])
    bo.added_points = (np.array([[0.24]]) - bo.X.mean(0)) / bo.X.std(0)
    bo.add_point()
    assert np.isclose(bo.Ymin[-1], (0.24 - 0.5) ** 2) and bo.X_train.shape == (7, 1) and bo.C_train.shape == (7, 1)
    # unconstrained problems and a generated design
    class Free:
        constraint, dim = False, 2
        def fun(self, x):
            return [np.sum(x ** 2, 1, keepdims=True)]
    bo2 = SO_BO(Free(), DoE_size=6, model_Y_dic={'num_layers': 1, 'num_units': 2, 'kernels': 'rbf', 'num_samples': 4}, seed=3)
    assert bo2.X.shape == (6, 2) and bo2.C is None and np.isclose(bo2.Ymin[-1], bo2.Y.min())


def test_mf_kernel_plans_and_no_cpu_fallback():
    """The multi-fidelity mirror maps the reference's kernel expressions (MF_DGP_EM.py:341-367) onto the C-ABI kernel
    kinds, refuses anything else, and needs the device as soon as a layer is evaluated."""
    from dgp_dace.gpflow_compat import RBF, LinearKernel, White, Matern32
    from dgp_dace.models import MF_DGP_EM as MF
    k0 = RBF(active_dims=[0, 1], variance=1.0, lengthscales=[1.0, 1.0]) + White(variance=1e-6)
    kind, pars, white = MF._kernel_plan(k0, 2)
    assert kind == 0 and [p._value.size for p in pars] == [1, 2] and float(white._value) == 1e-6
    kc, kp, kl, ki = RBF(active_dims=[0, 1]), RBF(active_dims=[2]), LinearKernel(active_dims=[2]), RBF(active_dims=[0, 1])
    kind, pars, white = MF._kernel_plan(kc * (kp + kl) + ki, 3)
    assert kind == 3 and white is None
    assert [id(p) for p in pars] == [id(kc.variance), id(kc.lengthscales), id(kp.variance), id(kp.lengthscales),
                                     id(kl.variance), id(ki.variance), id(ki.lengthscales)]
    kind, pars, white = MF._kernel_plan(RBF(active_dims=[0]) * RBF(active_dims=[1]) + RBF(active_dims=[0]) + White(0.1), 2)
    assert kind == 3 and pars[4] is None and float(white._value) == 0.1            # add_linear=False
    for bad in (Matern32() + White(), RBF(active_dims=[1]) * RBF(active_dims=[0]) + RBF(active_dims=[0]),
                RBF() + RBF()):
        with pytest.raises(NotImplementedError):
            MF._kernel_plan(bad, 2)
    # host kernel used once for the prior initialisation of q_sqrt against the oracle's kernel
    import torch
    import mf_dgp_em_oracle as mo
    rng = np.random.default_rng(0)
    Z = rng.uniform(0, 1, (7, 3))
    vals = list(rng.uniform(0.5, 1.5, 7))
    k = {"type": "mf", "Dx": 2, "white_variance": mo._t(0.2)}
    for nm, v in zip(["corr_variance", "corr_lengthscales", "prev_variance", "prev_lengthscales", "lin_variance",
                      "in_variance", "in_lengthscales"], vals):
        k[nm] = mo._t(v)
    np.testing.assert_allclose(MF._kernel_matrix_host(3, vals, 0.2, Z), mo.kern_K(k, torch.as_tensor(Z)).detach().numpy(),
                               rtol=1e-12, atol=1e-12)
    if not torch.cuda.is_available():
        from dgp_dace._native import NativeUnavailable
        X = [rng.uniform(0, 1, (6, 2)), rng.uniform(0, 1, (4, 1))]
        Y = [rng.standard_normal((6, 1)), rng.standard_normal((4, 1))]
        with pytest.raises(NativeUnavailable):
            MF.MultiFidelityDeepGP_EM(X, Y, [rng.uniform(0, 1, (4, 2))])
    src = open(os.path.join(ROOT, "dgp-toolbox_amd", "dgp_dace", "models", "MF_DGP_EM.py")).read()
    assert "oracle" not in src and "import torch" not in src


def test_mf_oracle_autograd_against_finite_differences():
    """The torch restatement of the MF-DGP-EM bound: autograd against central differences on a few leaves, and the
    literal N_{f+1}/N_f scale of the projection term (MF_DGP_EM.py:292-294)."""
    import torch
    import mf_dgp_em_oracle as mo
    rng = np.random.default_rng(0)
    X = [rng.uniform(0, 1, (9, 2)), rng.uniform(0, 1, (5, 3))]
    Y = [rng.standard_normal((9, 1)), rng.standard_normal((5, 1))]
    X_red = [rng.uniform(0, 1, (5, 2))]
    P = mo.make_params(X, [x.copy() for x in X], [X[-1].copy()])
    with torch.no_grad():
        for i, l in enumerate(P["layers"]):
            l["q_mu"].copy_(torch.as_tensor(Y[i])); l["q_sqrt"].mul_(0.3)
        P["layers"][0]["kern"]["white_variance"].fill_(0.05)
    S = 3
    nm = mo.draw_normals(rng, X, P, S)
    val, parts, g = mo.elbo_and_grads(P, X, Y, X_red, nm, S)
    assert abs(val - (parts["L"] + parts["L_red"] - parts["KL"] - parts["KL_red"])) < 1e-9 * abs(val)
    lv = mo.leaves(P)
    for name in ("layers.1.kern.lin_variance", "layers.1.Z", "layers_red.0.Z", "layers.0.kern.white_variance", "layers.0.q_sqrt"):
        t = lv[name]
        idx = tuple(0 for _ in t.shape)
        h = 1e-6
        with torch.no_grad(): t[idx] += h
        vp = mo.elbo_and_grads(P, X, Y, X_red, nm, S)[0]
        with torch.no_grad(): t[idx] -= 2 * h
        vm = mo.elbo_and_grads(P, X, Y, X_red, nm, S)[0]
        with torch.no_grad(): t[idx] += h
        fd = (vp - vm) / (2 * h)
        assert abs(fd - g[name][idx]) <= 1e-5 * max(1.0, abs(fd)), (name, fd, g[name][idx])
    # doubling the high-fidelity set's size while halving nothing else doubles the scale factor only through N_1 / N_0
    X2 = [X[0], np.concatenate([X[1], X[1]])]
    P2 = mo.make_params(X2, [X[0].copy(), X[1].copy()], [X[1].copy()])
    with torch.no_grad():
        for a, b in zip(mo.leaves(P2).values(), mo.leaves(P).values()):
            a.copy_(b)
    nm2 = {"zright": nm["zright"], "zs": [nm["zs"][0], [np.concatenate([z, z], 1) for z in nm["zs"][1]]],
           "ws": [nm["ws"][0], [np.concatenate([w, w], 1) for w in nm["ws"][1]]],
           "ws_proj": [[np.concatenate([w, w], 1) for w in nm["ws_proj"][0]]]}
    _, parts2, _ = mo.elbo_and_grads(P2, X2, [Y[0], np.concatenate([Y[1], Y[1]])], [np.concatenate([X_red[0], X_red[0]])], nm2, S)
    assert abs(parts2["L_red"] - 4.0 * parts["L_red"]) <= 1e-9 * abs(parts["L_red"])      # 2x the points, 2x the scale


def test_utils_surface_reparameterize_and_broadcasting_likelihood():
    """Callable surface of utils.py:22-117 (element-wise helpers on arrays; the models evaluate these inside kernels)."""
    from dgp_dace.gpflow_compat import Gaussian
    from dgp_dace.utils.utils import BroadcastingLikelihood, reparameterize
    rng = np.random.default_rng(0)
    S, N, D = 3, 5, 2
    mean, z = rng.standard_normal((S, N, D)), rng.standard_normal((S, N, D))
    var = rng.uniform(0.1, 1.0, (S, N, D))
    np.testing.assert_allclose(reparameterize(mean, var, z), mean + z * np.sqrt(var + 1e-6), rtol=1e-15)
    assert reparameterize(mean, None, z) is mean
    A = rng.standard_normal((S, D, N, N))
    full = np.transpose(A @ np.transpose(A, (0, 1, 3, 2)) + 0.1 * np.eye(N), (0, 2, 3, 1))            # S,N,N,D
    f = reparameterize(mean, full, z, full_cov=True)
    for s in range(S):
        for d in range(D):
            Lc = np.linalg.cholesky(full[s, :, :, d] + 1e-6 * np.eye(N))
            np.testing.assert_allclose(f[s, :, d], mean[s, :, d] + Lc @ z[s, :, d], rtol=1e-12, atol=1e-12)
    lik = BroadcastingLikelihood(Gaussian(variance=0.3))
    Y = rng.standard_normal((N, D))
    m = O.OracleDGP(*notebook_data(), [O.RBF(), O.RBF(), O.RBF()], [1, 1], lik_variance=0.3)
    np.testing.assert_allclose(lik.variational_expectations(mean, var, Y), m.variational_expectations(mean, var, Y), rtol=1e-14)
    mu, v = lik.predict_mean_and_var(mean, var)
    np.testing.assert_allclose(v, var + 0.3, rtol=1e-15); np.testing.assert_allclose(mu, mean)
    np.testing.assert_allclose(lik.logp(mean, Y), lik.variational_expectations(mean, 0 * var, Y), rtol=1e-14)
    np.testing.assert_allclose(lik.predict_density(mean, var, Y), -0.5 * np.log(2 * np.pi * (var + 0.3)) - 0.5 * (Y[None] - mean) ** 2 / (var + 0.3))
    assert lik.conditional_variance(mean).shape == (S, N, D) and np.allclose(lik.conditional_variance(mean), 0.3)
    assert lik.needs_broadcasting is False


def test_mfma_hazard_gate_is_part_of_the_build():
    """The kernels with inline-asm MFMAs (gemm_gram.h, gemm_tall.h, gemm_tallu.h) are scanned for a VALU write of an MFMA
    operand right in front of the MFMA - a hazard hipcc cannot see through the asm statement and that once produced
    single wrong accumulator elements.  (1) the scanner flags a synthetic hazard and accepts the separated form;
    (2) `make` runs it on the kept gfx950 assembly and leaves its stamp only when it found nothing."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd", "csrc"))
    import check_mfma_hazard as H
    hazard = ["v_mov_b64 v[10:11], 0", "v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[4:5], v[10:11]"]
    n, found = H.scan(hazard)
    assert n == 1 and len(found) == 1
    n, found = H.scan([hazard[0], "s_nop 1", hazard[1]])
    assert n == 1 and not found
    n, found = H.scan(["v_mul_f64 v[2:3], v[2:3], v[8:9]", "ds_read_b128 v[20:23], v30", hazard[1]])      # A operand, one slot
    assert len(found) == 1
    csrc = os.path.join(ROOT, "dgp-toolbox_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-j8"], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(csrc, "build", ".hazard_ok"))
    asms = [os.path.join(csrc, "build", f"{k}-hip-amdgcn-amd-amdhsa-gfx950.s") for k in ("gemm_gram", "gemm_tall", "gemm_tallu")]
    assert H.main(asms) == 0
