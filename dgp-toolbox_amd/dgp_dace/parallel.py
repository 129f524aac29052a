"""Data-point sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The reference has no multi-device code at all; this is new.  The ELBO data term
and every gradient are sums over data points (dgp.py:87,96), so rank r keeps the points
[r*N/W, (r+1)*N/W) with all S samples of each point, computes its partial sums into one contiguous
fp64 buffer (dgp_grad_partial), and a single all-reduce(sum) of that buffer over xGMI precedes the
replicated small-matrix chain and parameter update.  The Monte-Carlo normals are keyed by the
global point index, so the result does not depend on the number of ranks (up to summation order).
torch is used for the process group, the device buffer and the collective only.
"""
from __future__ import annotations

import os
import sys


SHARD_ALIGN = 16      # k-tile of the reductions over points: an aligned shard needs no tail launch (gemm_f64.hip)


def shard_bounds(N, rank, world):
    """Contiguous, balanced split of N points: rank r owns [lo, hi).  Interior boundaries sit on multiples of 16 when
    every shard stays non-empty (the sizes then differ by at most 16 + 1 points), so that the products that reduce over
    a rank's points (K = number of points, 16 per k-tile) have no ragged tail."""
    def cut(r):
        if r <= 0:
            return 0
        if r >= world:
            return N
        b = (r * N) // world
        return b if N < 2 * SHARD_ALIGN * world else ((b + SHARD_ALIGN // 2) // SHARD_ALIGN) * SHARD_ALIGN
    return cut(rank), cut(rank + 1)


class Dist:
    def __init__(self, torch, dist):
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.local_rank = int(os.environ.get("LOCAL_RANK", self.rank))
        # device tensors whenever a GPU is visible: RCCL ("nccl") in production; gloo also reduces device tensors
        # (staged through the host), which lets two ranks rehearse the whole path on ONE GPU in tests
        self.on_gpu = dist.get_backend() == "nccl" or (torch.cuda.is_available() and os.environ.get("DGP_DIST_DEVICE", "1") == "1")
        self._stream = None
        self._timing = None

    def local_device(self):
        n = self.torch.cuda.device_count() if self.on_gpu else 1
        return self.local_rank % max(1, n)

    def shard(self, N):
        return shard_bounds(N, self.rank, self.world)

    def stream_handle(self, dev):
        """A torch-owned HIP stream the native context launches on, so collectives are ordered with the kernels."""
        if not self.on_gpu:
            return None
        self.torch.cuda.set_device(dev)
        self._stream = self.torch.cuda.Stream(device=dev)
        return self._stream.cuda_stream

    def device_buffer(self, n, dev):
        """Zero-filled fp64 buffer.  On a GPU it is allocated and filled on the stream the native context launches on, so
        the fill is ordered before every kernel that accumulates into it (the default stream would not be)."""
        if not self.on_gpu:
            return self.torch.zeros(int(n), dtype=self.torch.float64, device="cpu")
        if self._stream is None:
            t = self.torch.zeros(int(n), dtype=self.torch.float64, device=f"cuda:{dev}")
            self.torch.cuda.current_stream(dev).synchronize()
            return t
        with self.torch.cuda.stream(self._stream):
            t = self.torch.zeros(int(n), dtype=self.torch.float64, device=f"cuda:{dev}")
        t.record_stream(self._stream)
        return t

    def all_reduce_(self, t):
        """In-place sum over ranks of the partial-sum buffer (RCCL ring over xGMI on GPUs)."""
        if self._stream is not None:
            with self.torch.cuda.stream(self._stream):
                if self._timing is not None:       # event pair on the stream the collective is ordered on (bench.py)
                    e0, e1 = (self.torch.cuda.Event(enable_timing=True) for _ in range(2))
                    e0.record(self._stream)
                    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
                    e1.record(self._stream)
                    self._timing.append((e0, e1))
                else:
                    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def timing(self, on):
        """Start (True) or stop (False) collecting one HIP-event pair per all-reduce of the partial-sum buffer."""
        self._timing = [] if on else None

    def timing_read(self):
        """Milliseconds of every all-reduce since timing(True): from the moment the rank's stream reaches the collective to
        the moment it may go on, i.e. waiting for the slowest rank included.  Synchronises the stream."""
        if not self._timing:
            return []
        self._stream.synchronize()
        return [e0.elapsed_time(e1) for e0, e1 in self._timing]

    def all_reduce_scalar(self, v, dev=0):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=(f"cuda:{dev}" if self.on_gpu else "cpu"))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def all_reduce_max(self, v, dev=0):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=(f"cuda:{dev}" if self.on_gpu else "cpu"))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_reduce_min(self, v, dev=0):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=(f"cuda:{dev}" if self.on_gpu else "cpu"))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def barrier(self):
        self.dist.barrier()

    def native_comm(self):
        """True when the library should own the collective (dgp_comm_init: RCCL on the library's comm stream, per-layer
        all-reduces overlapped with the backward pass).  Opt-in with DGP_COMM=native (any backend with device tensors):
        the default is the process group's own all-reduce (`torch`), because the library-owned communicator has only ever
        run with one rank on the boxes this was built on - a multi-rank run must be recorded before it becomes the default."""
        mode = os.environ.get("DGP_COMM", "torch")      # torch (default) | native
        return mode == "native" and self.on_gpu

    def broadcast_bytes(self, payload, dev):
        """Rank 0's `payload` (bytes) on every rank, through the process group (a uint8 tensor on the group's device)."""
        n = len(payload)
        device = f"cuda:{dev}" if self.dist.get_backend() == "nccl" else "cpu"
        t = self.torch.tensor(list(payload), dtype=self.torch.uint8, device=device) if self.rank == 0 else \
            self.torch.zeros(n, dtype=self.torch.uint8, device=device)
        self.dist.broadcast(t, src=0)
        return bytes(t.cpu().tolist())


def current():
    """The active process group wrapper, or None when running single-process."""
    if "torch" not in sys.modules:
        return None
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return None
    return Dist(torch, dist)
