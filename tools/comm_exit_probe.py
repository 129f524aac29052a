import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "dgp-toolbox_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from helpers import load, product_from_golden, n_layers
from dgp_dace._native import Context
flags = set(sys.argv[1:])
g = load("case_B_nonwhite")
m = product_from_golden(g)
nl = n_layers(g)
zs = [g[f"zs{i}"] for i in range(nl)]
ctx = m._sync_model()
m._sync_data(m.data)
S = int(g["S"])
if "avail" in flags:
    assert Context.comm_available()
if "step0" in flags:
    ctx.grad_step(S, 0, zs, want_elbo=True)
if "init" in flags:
    ctx.comm_init(0, 1, Context.comm_unique_id())
if "step" in flags:
    ctx.grad_step(S, 0, zs, want_elbo=True)
if "ar" in flags:
    import torch
    t = torch.arange(1000, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.comm_allreduce(t.data_ptr(), t.numel())
    ctx.sync()
if "destroy" in flags:
    ctx.comm_destroy()
if "close" in flags:
    ctx.close()
print("done", sorted(flags), flush=True)
