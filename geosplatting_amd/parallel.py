"""Data-parallel layer for the path: one process per GPU, views sharded across ranks, ONE flat fp32 gradient
bucket summed with RCCL (``torch.distributed`` backend "nccl" on ROCm) over xGMI.

The reference has no distributed code at all: its 8 views per step are a Python loop on
one device (rfstudio/model/geosplat.py:869-879) and the loss is the mean over views
(rfstudio/trainer/geosplat_trainer.py:180).  Here every rank holds a replica of the Gaussians / PBR attributes /
cubemap, renders its own views, and the only exchange is the gradient sum:
   message = 76 B x N (means 12, scales 12, quats 16, opacity 4, normals 12, kd 12, ks 8) + cubemap + exposure
   (149 MB + 18.9 MB at N = 1.97 M) in ONE all-reduce -- xGMI is point-to-point, so a single large collective
   that RCCL can spread over all 7 links beats many small ones.
The bucket is laid out once; gradients are copied in with a single fused foreach copy, reduced on a side
stream, and handed back as views (no per-parameter collectives).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def collectives_active() -> bool:
    """True when the step has to run its collectives: a process group with more than one rank -- or ANY initialised group under
    GEOSPLAT_COLLECTIVES_AT_WORLD1=1, the switch with which a one-GPU box executes the complete multi-GPU code path (sharded
    prefilter with its own communicator, two-phase gradient all-reduce on the communication stream, graph replay beside them)
    through RCCL itself: `torch.distributed` backend "nccl" with ONE rank (tests/test_gpu_parallel.py::test_rccl_*, bench.py
    --rccl-world1).  Two ranks cannot share a GPU under RCCL ("Duplicate GPU detected"), one rank can have it to itself."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("GEOSPLAT_COLLECTIVES_AT_WORLD1") == "1"


def shard_views(num_views: int, rank: int, world_size: int) -> List[int]:
    """View i -> rank i mod world_size; returns this rank's view indices in order."""
    return list(range(rank, num_views, world_size))


class GradBucket:
    """One flat fp32 buffer holding every parameter gradient of the path."""

    def __init__(self, shapes: Dict[str, Sequence[int]], device: torch.device, comm_stream=None):
        self.names = list(shapes.keys())
        self.shapes = {k: tuple(v) for k, v in shapes.items()}
        self.offsets: Dict[str, int] = {}
        off = 0
        for k in self.names:
            self.offsets[k] = off
            n = 1
            for s in self.shapes[k]:
                n *= int(s)
            off += (n + 63) // 64 * 64              # 256-byte aligned segments
        self.numel = off
        self.storage = torch.zeros(off, dtype=torch.float32, device=device)   # bucket + per-step scratch behind it (zero_with_scratch)
        self.flat = self.storage[:off]
        if device.type == "cuda":
            from ._lib import shared_stream
            self.comm_stream = comm_stream if comm_stream is not None else shared_stream(device, "comm")
        else:
            self.comm_stream = None

    def zero_with_scratch(self, sizes: Sequence[int]) -> List[Tensor]:
        """Zero the bucket AND `len(sizes)` scratch tensors (float counts) that live behind it in the same allocation with ONE fill
        -- the step's accumulators (activation gradients, texel gradients) used to be four `torch.zeros` launches at the head of
        every step.  The allocation grows on first use (the bucket's views stay valid: `flat` is re-derived before anyone reads it);
        the scratch tensors are per-step temporaries, valid until the next call."""
        offs, off = [], self.numel
        for n in sizes:
            offs.append(off)
            off += (int(n) + 63) // 64 * 64
        if self.storage.numel() < off:
            self.storage = torch.zeros(off, dtype=torch.float32, device=self.storage.device)
            self.flat = self.storage[:self.numel]
        else:
            self.storage[:off].zero_()
        return [self.storage[o:o + int(n)] for o, n in zip(offs, sizes)]

    def view(self, name: str) -> Tensor:
        o = self.offsets[name]
        n = 1
        for s in self.shapes[name]:
            n *= int(s)
        return self.flat[o:o + n].view(self.shapes[name])

    def pack(self, grads: Dict[str, Optional[Tensor]]) -> None:
        """Copy (not accumulate) per-parameter gradients into the bucket; missing grads become zeros."""
        dst, src = [], []
        for k in self.names:
            g = grads.get(k)
            if g is None:
                self.view(k).zero_()
            else:
                dst.append(self.view(k)); src.append(g.reshape(self.shapes[k]).to(torch.float32))
        if dst:
            torch._foreach_copy_(dst, src)

    def all_reduce(self, average: bool = False, async_op: bool = False):
        """Sum over ranks (RCCL on GPUs, gloo on CPU).  With async_op the collective runs on a side stream and
        the returned callable must be invoked before the gradients are read."""
        if not collectives_active():
            return (lambda: None) if async_op else None
        if self.comm_stream is not None and async_op:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                if average:
                    self.flat.div_(dist.get_world_size())

            def wait():
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            return wait
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if average:
            self.flat.div_(dist.get_world_size())
        return (lambda: None) if async_op else None

    def all_reduce_split(self, first_of_tail: str):
        """Two-phase sum for a step whose last gradients arrive late (the cubemap gradient comes out of the prefilter
        backward, the per-Gaussian gradients are complete before it): returns (start_head, finish).  `start_head()`
        launches the all-reduce of everything BEFORE segment `first_of_tail` on the communication stream -- it then
        overlaps the prefilter backward -- and `finish()` reduces the tail segments and joins the streams."""
        cut = self.offsets[first_of_tail]
        head, tail = self.flat[:cut], self.flat[cut:]
        if not collectives_active():
            return (lambda: None), (lambda: None)
        cs = self.comm_stream

        def start_head():
            if cs is not None:
                cs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cs):
                    dist.all_reduce(head, op=dist.ReduceOp.SUM)
            else:
                dist.all_reduce(head, op=dist.ReduceOp.SUM)

        def finish():
            if cs is not None:
                cs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cs):
                    dist.all_reduce(tail, op=dist.ReduceOp.SUM)
                torch.cuda.current_stream().wait_stream(cs)
            else:
                dist.all_reduce(tail, op=dist.ReduceOp.SUM)
        return start_head, finish

    def all_reduce_rows(self, names: Sequence[str], n0: int, n1: int, after=None) -> None:
        """Sum rows [n0, n1) of the listed segments (one Gaussian RANGE of the per-Gaussian gradients) on the communication stream,
        which first waits for `after` (an event: the tail launches that made those rows final).  The slices go out as ONE coalesced
        group call where the backend has one (RCCL: a single launch for the seven slices), else one after the other.  Chunk k's sum
        then runs while the tail still computes chunk k + 1 (engine._chunked_tail); `join_comm()` orders the caller behind them."""
        if not collectives_active():
            return
        views = [self.view(k)[n0:n1] for k in names]
        cs = self.comm_stream
        if cs is None:
            for v in views:
                dist.all_reduce(v, op=dist.ReduceOp.SUM)
            return
        if after is not None:
            cs.wait_event(after)
        else:
            cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            _all_reduce_coalesced(views)

    def join_comm(self) -> None:
        if self.comm_stream is not None and collectives_active():
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def all_reduce_names(self, names: Sequence[str], group=None) -> None:
        """Sum only the listed segments (on the communication stream, joined before returning)."""
        if not collectives_active():
            return
        cs = self.comm_stream
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                for k in names:
                    dist.all_reduce(self.view(k), op=dist.ReduceOp.SUM, group=group)
            torch.cuda.current_stream().wait_stream(cs)
        else:
            for k in names:
                dist.all_reduce(self.view(k), op=dist.ReduceOp.SUM, group=group)

    def unpack(self) -> Dict[str, Tensor]:
        return {k: self.view(k) for k in self.names}


def _all_reduce_coalesced(tensors: Sequence[Tensor]) -> None:
    """SUM every tensor over the ranks, as one group call when the backend coalesces (RCCL: ncclGroupStart / End around the calls)."""
    mgr = getattr(dist, "_coalescing_manager", None)
    if mgr is not None and tensors[0].is_cuda and dist.get_backend() == "nccl":
        try:
            with mgr(device=tensors[0].device):
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return
        except TypeError:
            pass                                               # (a different signature, raised before any call went out: single calls)
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def init_distributed_from_env(device_type: str = "cuda"):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run.  Returns (rank, world, device)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "nccl" if device_type == "cuda" else "gloo"
    if device_type == "cuda":
        # GEOSPLAT_DEBUG_SHARE_GPU=1: every rank on cuda:0 with the gloo backend -- lets a 1-GPU box exercise the
        # multi-rank code path of bench.py end to end (never used for measurements)
        if os.environ.get("GEOSPLAT_DEBUG_SHARE_GPU") == "1":
            local, backend = 0, "gloo"
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if (world > 1 or os.environ.get("GEOSPLAT_COLLECTIVES_AT_WORLD1") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device
