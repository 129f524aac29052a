"""Functional + timing check of the other BASELINE.json configurations on one GPU."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgp-toolbox_amd")); sys.path.insert(0, ROOT)
import numpy as np
from bench import synthetic
from dgp_dace.gpflow_compat import RBF, Gaussian
from dgp_dace.models.dgp import DGP


def build(N, D, M, S, num_units):
    X, Y, Z = synthetic(N, D, M)
    with contextlib.redirect_stdout(io.StringIO()):
        m = DGP(X, Y, Z, [RBF(1.0, [1.0] * d) for d in [D] + num_units], num_units, Gaussian(), num_samples=S)
    return m


def timed(label, f, n):
    f(); m.sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    m.sync(); dt = (time.perf_counter() - t0) / n
    print(f"{label}: {1e3 * dt:.2f} ms/iteration = {1 / dt:.2f} it/s", flush=True)


# config 1 (plumbing): 1 hidden layer, N=1k, D=1, M=32
m = build(1000, 1, 32, 10, [1])
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
def it():
    c = m._grad_step(m.data); c.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
m.sync(); timed("config 1 (N=1k, D=1, M=32, [1]) adam iteration", it, 50)
with contextlib.redirect_stdout(io.StringIO()) as buf:
    m.optimize_nat_adam(iterations1=20, iterations2=30, messages=10)
print("config 1 optimize_nat_adam ELBO trace:", [round(float(l.split(':')[1]), 2) for l in buf.getvalue().splitlines()], flush=True)
mean, var = m.predict(np.linspace(-2, 2, 7)[:, None], 20)
print("config 1 predict ok:", mean.shape, var.shape, bool(np.all(np.isfinite(mean)) and np.all(var > 0)), flush=True)

# config 2: natural-gradient iteration (Part 2 of optimize_nat_adam: 2 ELBO evaluations + Adam + natgrad)
m = build(100_000, 8, 256, 10, [8, 8])
mask = m._natgrad_setup(True)
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
def it2():
    c = m._grad_step(m.data); c.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
    c = m._grad_step(m.data); c.natgrad_step(0.01, mask)
m.sync(); timed("config 2 (N=100k, D=8, M=256, [8,8]) nat_adam Part-2 iteration", it2, 5)
print("config 2 ELBO after those iterations:", ctx.last_elbo(), flush=True)

# config 2-alt: one hidden layer [8]
m = build(100_000, 8, 256, 10, [8])
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
m.sync(); timed("config 2-alt (N=100k, D=8, M=256, [8]) adam iteration", it, 10)

# config 4 shape, minibatch of 10 000 points per GPU
m = build(10_000, 16, 512, 10, [16, 16, 16])
mask = m._natgrad_setup(True)
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
m.sync(); timed("config 4 minibatch (10 000 points, D=16, M=512, [16,16,16]) adam iteration", it, 10)
m.sync(); timed("config 4 minibatch (10 000 points, D=16, M=512, [16,16,16]) nat_adam Part-2 iteration", it2, 5)

# config 4 shape, one GPU's share of N=1M over 8 GPUs: 4 SVGP layers, D=16, M=512
m = build(125_000, 16, 512, 10, [16, 16, 16])
mask = m._natgrad_setup(True)
for l in m.layers[:-1]: l.q_sqrt.assign(l.q_sqrt * 1e-3)
ctx = m._sync_model(); ctx.adam_reset(); fl = m._trainable_flags()
def it4():
    c = m._grad_step(m.data); c.adam_step(0.01, 0.9, 0.999, 1e-7, fl)
    c = m._grad_step(m.data); c.natgrad_step(0.01, mask)
m.sync(); timed("config 4 shard (N=125k of 1M, D=16, M=512, [16,16,16]) nat_adam Part-2 iteration", it4, 2)
print("config 4 ELBO:", ctx.last_elbo(), flush=True)
