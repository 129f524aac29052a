#!/usr/bin/env python3
"""Benchmark of the DGP hot path: ELBO iterations/sec on BASELINE.json config 2
(`num_units=[8,8]` = 3 SVGP layers, N=100k, D=8, M=256, S=10, fp64), one iteration = one
`DGP.optimize_adam` loop body (dgp.py:271-276): ELBO forward + backward + Adam update.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

Prints ONE JSON line (rank 0).  Strong scaling: the N data points are sharded over the ranks and the
per-point sums are all-reduced once per iteration (RCCL).  `roofline` is measured live with HIP
events on the engine's stream (dgp_prof_*); `cpu_baseline` times the oracle's torch-CPU restatement
of the reference's dense formulation on a bounded sample, on rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd"))

import numpy as np  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6     # AMD MI355X fp64 matrix spec; issue-rate microbench in NOTES.md
HBM_PEAK_GBS = 8000.0


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """CPU share of this process: affinity mask and cgroup quota (os.cpu_count() reports the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def synthetic(N, D, M, seed=0):
    """BASELINE.md §3 inputs."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    w = rng.standard_normal((D, 1))
    Y = np.sin(2 * np.pi * X @ w / np.sqrt(D)) + (X[:, :1] > 0.5) + 0.05 * rng.standard_normal((N, 1))
    X = (X - X.mean(0)) / X.std(0)
    Y = (Y - Y.mean(0)) / Y.std(0)
    Z = X[rng.permutation(N)[:M]].copy()
    return X, Y, Z


def alg_flops_step(N, S, dims, M, Dy):
    """Algorithmic flops of one iteration as built (triangular products, first layer once per data point): per layer and point
    (1 + D_out) * M(M+1) forward (c, t_d) and (2 D_out + 1) * M(M+1) backward (dC from the stored t_d, dK, G_d).  The fourth
    backward product of rounds 1-2, Q' = sum_p cbar_p c_p^T, is no longer a pass over the points (it follows from G_d, du and W in
    the finish chain) and is no longer counted."""
    total = 0.0
    douts = dims[1:] + [Dy]
    for l, dout in enumerate(douts):
        P = N if l == 0 else N * S
        total += P * (3.0 * dout + 2.0) * M * (M + 1.0)
    return total


def survey_per_unit_flops_step(N, S, dims, M, Dy, white=False):
    """SURVEY.md §8d's per-unit figure (flops per point of a layer, F_fwd(layer)/P, x3 for a step) times the units
    the launches of one step actually process: N points in the first layer (evaluated once, not per sample),
    S*N in the others."""
    w = 1.0 if white else 2.0
    douts = dims[1:] + [Dy]
    total = 0.0
    for l, (din, dout) in enumerate(zip(dims, douts)):
        per_point = M * (2 * din + 3) + w * M * M + 2 * M * dout + dout * M * (M + 3.0) + 6 * dout
        total += 3.0 * per_point * (N if l == 0 else float(N) * S)
    return total


def survey_flops_step(N, S, dims, M, Dy, white=False):
    """SURVEY.md §8d's own count (kept for the CPU test that reproduces SURVEY's table; no longer part of the bench line): F_step = 3 * sum_layers P*[M(2 D_in + 3) + w M^2 + 2 M D_out +
    D_out M (M+3) + 6 D_out] with P = S*N in EVERY layer and w = 2 solves (non-white).  It credits work this
    implementation does not execute (the first layer once per sample, the second triangular solve)."""
    w = 1.0 if white else 2.0
    douts = dims[1:] + [Dy]
    P = float(N) * S
    return 3.0 * sum(P * (M * (2 * din + 3) + w * M * M + 2 * M * dout + dout * M * (M + 3.0) + 6 * dout)
                     for din, dout in zip(dims, douts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--N", type=int, default=100_000)
    ap.add_argument("--D", type=int, default=8)
    ap.add_argument("--M", type=int, default=256)
    ap.add_argument("--S", type=int, default=10)
    ap.add_argument("--num-units", type=str, default="8,8")
    ap.add_argument("--minibatch", type=int, default=0,
                    help="points per iteration over all ranks (0 = full batch, the headline configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nat-steps", type=int, default=10,
                    help="after the timed Adam region: this many part-2 iterations of optimize_nat_adam (2 evaluations + Adam + "
                         "natural gradient, dgp.py:337-345), timed the same way and reported as extra keys (0 = skip)")
    ap.add_argument("--breakdown-steps", type=int, default=0,
                    help="iterations of the per-category breakdown pass behind the timed region, beside which rocm-smi is sampled "
                         "(0 = 40, or min(steps, 10) for iterations longer than 0.2 s)")
    ap.add_argument("--cpu-sample", type=int, default=8192)
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries write to file descriptor 1 behind Python's back (RCCL prints a
    # version banner when a communicator is created): everything but the final line is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("DGP_BENCH_BACKEND", "nccl")       # "gloo": rehearsal of the N>1 path on fewer GPUs
        ndev = max(1, torch.cuda.device_count())
        local_rank = local_rank % ndev
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from dgp_dace.gpflow_compat import RBF, Gaussian
    from dgp_dace.models.dgp import DGP

    num_units = [int(x) for x in args.num_units.split(",") if x]
    dims = [args.D] + num_units
    X, Y, Z = synthetic(args.N, args.D, args.M)
    stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")          # the constructor prints the architecture (as the reference does)
    model = DGP(X, Y, Z, [RBF(variance=1.0, lengthscales=[1.0] * d) for d in dims], num_units, Gaussian(),
                num_samples=args.S, seed=0, device=local_rank, minibatch_size=args.minibatch or None)
    sys.stdout = stdout
    for layer in model.layers[:-1]:
        layer.q_sqrt.assign(layer.q_sqrt * 1e-3)      # dgp.py:268-269, what optimize_adam does first
    log("model built")
    ctx = model._sync_model()
    ctx.adam_reset()
    flags = model._trainable_flags()

    # single-process rehearsals of what a rank of a multi-GPU run executes (tools/shard_sizes.sh):
    #   DGP_BENCH_PATH=partial  the three-stage form of the default collective (grad_partial -> [all-reduce] -> grad_finish)
    #   DGP_BENCH_ONE_RANK_COMM=1  a one-rank library-owned RCCL communicator attached (DGP_COMM=native's path: per-layer
    #                              all-reduce on the comm stream, persistent kernels leave 8 CUs free)
    bench_path = os.environ.get("DGP_BENCH_PATH", "step")
    if world == 1 and bench_path == "partial":
        ctx.acc_info()          # as a sharded run does: the transport form of the partial sums (pack / unpack) is part of the rehearsal
    if world == 1 and os.environ.get("DGP_BENCH_ONE_RANK_COMM") == "1":
        from dgp_dace._native import Context
        ctx.comm_init(0, 1, Context.comm_unique_id())

    def step():
        if world == 1 and bench_path == "partial":
            model._sync_data(model.data)
            model._select_batch(ctx, True)
            ctx.grad_partial(model.num_samples, model._next_seed(), None)
            ctx.grad_finish()
            c = ctx
        else:
            c = model._grad_step(model.data)
        c.adam_step(0.01, 0.9, 0.999, 1e-7, flags)

    def fence():
        ctx.sync()
        if dist:
            torch.cuda.synchronize()
            dist.barrier()
            ctx.sync()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    log("warm-up done")
    # events only where the roofline needs a duration (measured: the iteration takes the same 47.5 ms with every scope between events)
    ctx.prof_enable(True, categories=["mfma_contractions"])
    if dist and model._dist is not None:
        model._dist.timing(True)
    t0 = time.perf_counter()
    ctx.prof_mark()
    for _ in range(args.steps):
        step()
        ctx.prof_mark()                      # HIP event on the engine's stream: per-step times without a host sync
    fence()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt:.3f} s")
    step_ms = ctx.prof_marks_read()
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    model._device_newer = True
    med_ms = float(np.median(step_ms)) if len(step_ms) else 1e3 * dt / args.steps
    ar_ms, ar_bytes = None, None
    if dist:
        dt = model._dist.all_reduce_max(dt, local_rank)
        med_ms = model._dist.all_reduce_max(med_ms, local_rank)
        ar = model._dist.timing_read()
        model._dist.timing(False)
        if ar:                                  # the collective of the default path (three-stage form), event-timed per iteration
            ar_ms = model._dist.all_reduce_max(float(np.median(ar)), local_rank)
            ar_bytes = int(model._acc_tensor.numel()) * 8
    elbo_last = ctx.last_elbo()

    # ---- per-category breakdown: a pass of its own behind the timed region (every category's launches between event pairs)
    # (dt is the max over ranks by now: every rank runs the same number of iterations here)
    bsteps = args.breakdown_steps if args.breakdown_steps > 0 else (40 if dt / args.steps < 0.2 else min(args.steps, 10))
    sampler = PowerSampler(local_rank) if rank == 0 else None       # package power / shader clock beside the SAME iterations, never beside
    ctx.prof_enable(True)                                           # the timed region (rocm-smi children on the host, nothing on the GPU)
    for _ in range(bsteps):
        step()
    fence()
    prof_all = ctx.prof_read()
    ctx.prof_enable(False)
    model._device_newer = True
    power = sampler.stop() if sampler is not None else None

    # ---- the natural-gradient iteration north_star names (part 2 of optimize_nat_adam, dgp.py:337-345): one Adam step on the
    # hyper-parameters + one natural-gradient step on every layer's q(u), each behind its own ELBO evaluation with fresh normals
    nat = None
    if args.nat_steps > 0:
        mask = model._natgrad_setup(True)
        nflags = model._trainable_flags()

        def nat_step():
            c = model._grad_step(model.data)
            c.adam_step(0.01, 0.9, 0.999, 1e-7, nflags)
            c = model._grad_step(model.data)
            c.natgrad_step(0.01, mask)

        for _ in range(2):
            nat_step()
        fence()
        tn = time.perf_counter()
        for _ in range(args.nat_steps):
            nat_step()
        fence()
        tn = time.perf_counter() - tn
        model._device_newer = True
        if dist:
            tn = model._dist.all_reduce_max(tn, local_rank)
        nat = {"ms": 1e3 * tn / args.nat_steps, "elbo": ctx.last_elbo()}
        log(f"natural-gradient iterations done: {nat['ms']:.2f} ms each")

    if rank == 0:
        it_s = args.steps / dt
        mf = prof["mfma_contractions"]
        achieved = mf["alg_flops"] / (mf["ms"] * 1e-3) / 1e12 if mf["ms"] > 0 else 0.0
        name, cus, mem = ctx.device_info()
        # HBM bytes per launch of the dominant kernel family: NOT measured in this run (PMC counters need rocprofv3 passes
        # of their own); taken from the committed summary of those passes, whose commit / command travel with the number
        traffic, traffic_source = None, None
        tpath = os.environ.get("DGP_TRAFFIC_JSON") or os.path.join(ROOT, "profiles", "r4_pmc_traffic.json")
        if world == 1 and os.path.exists(tpath) and (args.N, args.M, args.S) == (100_000, 256, 10):
            try:
                tj = json.load(open(tpath))
                traffic = tj["hbm_bytes_per_launch"]
                traffic_source = f"{os.path.relpath(tpath, ROOT)} (rocprofv3 PMC passes at commit {tj.get('commit', '?')}: {tj.get('source', '')})"
            except Exception:
                traffic, traffic_source = None, None
        out = {
            "metric": "ELBO iterations/sec (2-layer DGP, N=100k, M=256) at 1/2/4/8 GPUs; fp64 ELBO match",
            "value": it_s, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "ms_per_step_median": med_ms, "it_s_median": 1e3 / med_ms,
            "ms_per_step_min_max": [float(np.min(step_ms)), float(np.max(step_ms))] if len(step_ms) else None,
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"DGP num_units={num_units} ({len(num_units) + 1} SVGP layers), N={args.N}, "
                                   f"D={args.D}, M={args.M}, S={args.S}, " + (f"minibatch {args.minibatch}" if args.minibatch else "full batch") + ", optimize_adam iteration",
                       "N": args.N, "D": args.D, "M": args.M, "S": args.S, "num_units": num_units,
                       "parallelism": (f"data points sharded over {world} GPU(s), " +
                                       ("one all-reduce per layer and iteration under the backward pass (library-owned RCCL communicator)"
                                        if os.environ.get("DGP_COMM", "torch") == "native" else
                                        "one all-reduce of the partial-sum buffer per iteration (process group's collective) before the replicated small-matrix chains and Adam"))},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "dgp::gemm_wide_kernel + dgp::gemm_tall_kernel + dgp::gemm_tallu_kernel + dgp::gemm_gram_kernel + dgp::g_panel_kernel + dgp::gemm_f64_kernel (all point contractions, rank 0)",
                         "kernel_ms_per_step": mf["ms"] / args.steps, "launches_per_step": mf["launches"] / args.steps,
                         "alg_flops_per_step_rank0": mf["alg_flops"] / args.steps,
                         "whole_step_frac": alg_flops_step(args.minibatch or args.N, args.S, dims, args.M, 1) / world / (dt / args.steps)
                                            / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                         "frac_by_survey_8d_per_unit_figure": survey_per_unit_flops_step(args.minibatch or args.N, args.S, dims, args.M, 1)
                                                              / world / (mf["ms"] / args.steps * 1e-3) / 1e12
                                                              / FP64_MFMA_PEAK_TFLOPS},
            "breakdown_ms_per_step": {k: v["ms"] / bsteps for k, v in prof_all.items()},
            "breakdown_steps": bsteps,
            "breakdown_note": "from a pass of its own behind the timed region (in the timed region only the contractions' launches sit "
                              "between event pairs); HIP-event spans per category on the stream each launch was issued on; the small-matrix chains run on side "
                              "streams BESIDE full-chip persistent kernels, so their spans include waiting for CUs and the categories "
                              "overlap: the sum exceeds ms_per_step (their own cost is ~0.9 ms per step, profiles/r3_shard_sizes.txt)",
            "elbo_last": elbo_last, "device": name,
            "power": power,
        }
        if nat is not None:
            flops_it = 2.0 * alg_flops_step(args.minibatch or args.N, args.S, dims, args.M, 1) / world
            out["nat_adam_ms_per_iteration"] = nat["ms"]
            out["nat_adam_it_s"] = 1e3 / nat["ms"]
            out["nat_adam_steps"] = args.nat_steps
            out["nat_adam_whole_step_frac"] = flops_it / (nat["ms"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS
            out["nat_adam_note"] = ("part 2 of DGP.optimize_nat_adam (dgp.py:337-345): 2 ELBO evaluations + gradients, Adam on the "
                                    "hyper-parameters, natural-gradient step (gamma 0.01) on every layer's q(u); timed after the "
                                    "headline region, same barriers, max over ranks; `value` stays the optimize_adam iteration")
            out["nat_adam_elbo_last"] = nat["elbo"]
        if world > 1:
            native = os.environ.get("DGP_COMM", "torch") == "native" and getattr(model, "_native_comm", False)
            out["collective"] = ("library-owned RCCL communicator: one ncclAllReduce per layer on the comm stream (dgp_grad_step)" if native else
                                 f"torch.distributed all_reduce ({dist.get_backend()}) of the packed partial-sum buffer between dgp_grad_partial and dgp_grad_finish")
            out["allreduce_ms_per_step"] = ar_ms
            out["allreduce_bytes"] = ar_bytes
            out["allreduce_note"] = ("median over the timed steps of a HIP-event pair around the collective on the engine's stream (waiting for "
                                     "the slowest rank included), max over ranks; null under DGP_COMM=native (the collectives are inside the library)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, num_units, X, Y, Z)
            out["elbo_rel_err_vs_oracle"] = out["cpu_baseline"].pop("elbo_rel_err")
            if not args.minibatch and len(set(dims)) == 1:
                out["elbo_vs_closed_form"] = closed_form_check(args, num_units, X, Y, Z, local_rank)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()


class PowerSampler:
    """Package power, shader clock and temperatures as `rocm-smi` reports them, sampled from a thread while iterations run (DESIGN.md §5:
    the fp64 kernels sit at the package power limit, which is what bounds roofline.frac).  Best effort: None if rocm-smi is missing."""

    def __init__(self, gpu):
        import re
        import shutil
        import subprocess
        import threading
        self.gpu, self.samples, self.run = int(gpu), [], shutil.which("rocm-smi") is not None
        self.cap = None
        # not under a profiler: its preloaded tool would ride into every rocm-smi child
        if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
            self.run = False

        def loop():
            pat = {"sclk_mhz": re.compile(r"GPU\[%d\]\s*:\s*sclk clock level: \S+ \((\d+)Mhz\)" % self.gpu),
                   "package_w": re.compile(r"GPU\[%d\]\s*:.*Package Power \(W\):\s*([\d.]+)" % self.gpu),
                   "junction_c": re.compile(r"GPU\[%d\]\s*:\s*Temperature \(Sensor junction\) \(C\):\s*([\d.]+)" % self.gpu),
                   "hbm_c": re.compile(r"GPU\[%d\]\s*:\s*Temperature \(Sensor memory\) \(C\):\s*([\d.]+)" % self.gpu)}
            while self.run:
                try:
                    txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
                except Exception:
                    break
                row = {}
                for k, p in pat.items():
                    m = p.search(txt)
                    if m:
                        row[k] = float(m.group(1))
                if "package_w" in row and "sclk_mhz" in row:
                    self.samples.append(row)
            try:
                txt = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
                m = re.search(r"GPU\[%d\]\s*:.*Power \(W\):\s*([\d.]+)" % self.gpu, txt)
                self.cap = float(m.group(1)) if m else None
            except Exception:
                pass
        self.thread = threading.Thread(target=loop, daemon=True)
        if self.run:
            self.thread.start()

    def stop(self):
        if not self.run:
            return None
        self.run = False
        self.thread.join(timeout=30)
        busy = [r for r in self.samples if r["package_w"] > 600.0]
        if not busy:
            return None
        med = lambda k: float(np.median([r[k] for r in busy if k in r])) if any(k in r for r in busy) else None
        return {"samples": len(busy), "sclk_mhz_median": med("sclk_mhz"), "package_w_median": med("package_w"), "package_w_max": max(r["package_w"] for r in busy),
                "package_w_cap": self.cap, "junction_c_median": med("junction_c"), "hbm_c_median": med("hbm_c"),
                "note": "rocm-smi sampled from a host thread during the breakdown pass (the same optimize_adam iterations, behind the timed region); "
                        "roofline.peak assumes 2400 MHz"}


def closed_form_check(args, num_units, X, Y, Z, device):
    """`fp64 ELBO match` of BASELINE.json's metric at the FULL workload, against a formula rather than a program: a second model of the
    same shape with a random q(u) in every layer and given normals; dgp_elbo on the device against the doubly-stochastic bound assembled
    from the papers' formulas in NumPy (tests/helpers.py::dsdgp_elbo - Salimbeni & Deisenroth 2017 eq. 13-16 over the SVGP marginals of
    Hensman et al. 2013; identity mean functions, so equal layer widths only).  ~30 s of host time at config 2."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import dsdgp_elbo, stationary_kernel
    from dgp_dace.gpflow_compat import RBF, Gaussian
    from dgp_dace.models.dgp import DGP
    rng = np.random.default_rng(16)
    D, M, S, N = args.D, args.M, args.S, args.N
    kp = [(0.5 + 0.2 * i, np.linspace(0.9, 1.3, D) + 0.05 * i) for i in range(len(num_units) + 1)]
    noise = 0.25
    stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")
    m = DGP(X, Y, Z, [RBF(v, l) for v, l in kp], num_units, Gaussian(variance=noise), num_samples=S, seed=0, device=device)
    sys.stdout = stdout
    lay = []
    for l, (v, ls), dout in zip(m.layers, kp, num_units + [Y.shape[1]]):
        Lu = np.linalg.cholesky(stationary_kernel(Z, Z, v, ls) + 1e-6 * np.eye(M))
        q_mu = Lu @ (0.3 * rng.standard_normal((M, dout)))
        q_sqrt = np.stack([np.tril(Lu @ np.tril(0.4 * np.eye(M) + 0.5 / M * rng.standard_normal((M, M)))) for _ in range(dout)])
        l.q_mu.assign(q_mu)
        l.q_sqrt.assign(q_sqrt)
        lay.append(dict(Z=np.asarray(l.feature.Z.numpy()), variance=v, lengthscales=ls.copy(), q_mu=q_mu, q_sqrt=q_sqrt))
    ctx = m._sync_model()
    m._sync_data(m.data)
    zs = [rng.standard_normal((S, N, d)) for d in num_units] + [np.zeros((S, N, Y.shape[1]))]
    data, kl = ctx.elbo(S, 1, zs)
    t0 = time.perf_counter()
    want = dsdgp_elbo(X, Y, zs, lay, noise, 1e-6)
    dt = time.perf_counter() - t0
    return {"rel_err": abs((data - kl) - want) / abs(want), "elbo_device": data - kl, "elbo_closed_form": want, "host_seconds": dt,
            "what": f"dgp_elbo of a model of the bench's shape (num_units={num_units}, N={N}, D={D}, M={M}, S={S}; random q(u) in every layer, "
                    "given normals) against tests/helpers.py::dsdgp_elbo (the doubly-stochastic bound from the papers' formulas, NumPy)"}


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def cpu_baseline(args, num_units, X, Y, Z):
    """the oracle's torch-CPU fp64 restatement of the reference's dense formulation
    (layers.py:243-276, autograd backward, Keras-Adam update: one `optimize_adam` loop body, dgp.py:271-276) on all
    host cores.  Protocol: per sample size 1 warm-up + 3 timed iterations, median; the sample is evaluated as ONE
    block of points (no small chunks that would handicap BLAS); two sample sizes (n/2 and n) to show that the time is
    linear in the number of points before it is extrapolated to N.  Also checks the GPU ELBO against the oracle on the
    larger sample with identical Philox normals."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import dgp_oracle as O
    from dgp_oracle_train import OracleTrainer
    from dgp_dace.gpflow_compat import RBF, Gaussian
    from dgp_dace.models.dgp import DGP
    cores = usable_cores()
    cpu = cpu_model_name()
    log(f"cpu_baseline: {cores} cores, {cpu}")
    torch.set_num_threads(cores)
    n_big = min(args.cpu_sample, args.N)
    dims = [args.D] + num_units
    med = {}
    for n in sorted({max(256, n_big // 2), n_big}):
        Xs, Ys = X[:n].copy(), Y[:n].copy()
        mo = O.OracleDGP(Xs, Ys, Z, [O.RBF(1.0, np.ones(d)) for d in dims], num_units, num_samples=args.S)
        tr = OracleTrainer(mo, base_seed=0)
        tr.scale_inner_q_sqrt()
        adam = tr.new_adam(0.01, 0.9, 0.999, 1e-7)
        times = []
        for it in range(4):                       # the first one is the warm-up
            t0 = time.perf_counter()
            tr.adam_iteration(adam)
            times.append(time.perf_counter() - t0)
        med[n] = float(np.median(times[1:]))
        log(f"cpu_baseline: n={n}: iterations {['%.2f' % t for t in times]} s, median of the last three {med[n]:.2f} s")
    ns = sorted(med)
    lin = (med[ns[-1]] / med[ns[0]]) / (ns[-1] / ns[0]) if len(ns) > 1 else 1.0
    t = med[n_big]
    # ELBO of the GPU path against the oracle on the larger sample, same Philox normals
    Xs, Ys = X[:n_big].copy(), Y[:n_big].copy()
    mo = O.OracleDGP(Xs, Ys, Z, [O.RBF(1.0, np.ones(d)) for d in dims], num_units, num_samples=args.S)
    for l in mo.layers[:-1]:
        l.q_sqrt = l.q_sqrt * 1e-3
    import dgp_oracle_torch as T
    elbo_o, _ = T.elbo_and_grads(mo, O.draw_zs(mo, 123, args.S, n_big), want_grads=False)
    stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")
    mg = DGP(Xs, Ys, Z, [RBF(1.0, [1.0] * d) for d in dims], num_units, Gaussian(), num_samples=args.S)
    sys.stdout = stdout
    for l in mg.layers[:-1]:
        l.q_sqrt.assign(l.q_sqrt * 1e-3)
    c = mg._sync_model()
    mg._sync_data(mg.data)
    Ld, KL = c.elbo(args.S, 123, None)
    rel = abs((Ld - KL) - elbo_o) / abs(elbo_o)
    return {"value": 1.0 / (t * args.N / n_big), "unit": "it/s", "cores": cores, "kind": "port", "cpu": cpu,
            "sample": f"{n_big} of {args.N} data points (S={args.S}) as one block: 1 warm-up + 3 timed optimize_adam iterations "
                      f"(ELBO + autograd + Adam), median {t:.2f} s; {ns[0]} points: median {med[ns[0]]:.2f} s, i.e. time per point "
                      f"x{lin:.2f} from {ns[0]} to {ns[-1]} points (1.00 = linear), extrapolated linearly in N; torch-CPU fp64, "
                      f"dense SK@A formulation of layers.py:243-276",
            "seconds_per_iteration": {str(k): v for k, v in med.items()}, "linearity": lin,
            "elbo_rel_err": rel}


if __name__ == "__main__":
    main()
