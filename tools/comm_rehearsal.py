"""Two ranks on ONE GPU through the library-owned RCCL communicator (DGP_COMM=native over a gloo process group for the
id exchange): prints whether RCCL accepted two ranks on one device, and the ELBO / gradient against the single-process
values.  RCCL normally refuses duplicate devices; the point is that the refusal is survived (all ranks fall back to the
process group's all-reduce together) and that, where it is accepted, the native path gives the same numbers."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path[:0] = [os.path.join(ROOT, "dgp-toolbox_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.environ["LOCAL_RANK"] = str(RANK)
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % PORT, rank=RANK, world_size=WORLD)
from helpers import load, product_from_golden
g = load("case_B_nonwhite")
m = product_from_golden(g, seed=21)
m._engine()
ctx = m._grad_step(m.data)
elbo, grad = ctx.last_elbo(), ctx.grad_get()
if RANK == 0:
    np.savez(OUT, elbo=elbo, grad=grad, native=int(getattr(m, "_native_comm", False)))
dist.barrier()
dist.destroy_process_group()
'''
def run(world, env_extra):
    out = tempfile.mktemp(suffix=".npz")
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(world):
        code = f"ROOT={ROOT!r}; RANK={r}; WORLD={world}; PORT={port}; OUT={out!r}\n" + WORKER
        env = dict(os.environ, **env_extra)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    rc = [p.wait(timeout=300) for p in procs]
    import numpy as np
    return rc, (np.load(out) if os.path.exists(out) else None)

if __name__ == "__main__":
    import numpy as np
    rc1, one = run(1, {"DGP_COMM": "torch"})
    rc2, two = run(2, {"DGP_COMM": "native"})
    print("single process rc", rc1, " two ranks rc", rc2)
    if one is not None and two is not None:
        print("native communicator in use on two ranks:", bool(two["native"]))
        print("ELBO single %.15g  two ranks %.15g  rel diff %.2e" % (one["elbo"], two["elbo"], abs(one["elbo"] - two["elbo"]) / abs(one["elbo"])))
        print("gradient max rel diff %.2e" % (np.abs(one["grad"] - two["grad"]).max() / np.abs(one["grad"]).max()))
