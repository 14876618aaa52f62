"""Batch (data) parallelism over the GPUs of one node: one process per GPU, torch.distributed.

The path shards by image with no halo and no reduction: every rank runs an independent
Engine on its contiguous slab of the global batch; the only exchange is ONE all-gather of the output
slabs [B/G, rows, 85] into [B, rows, 85] at the end of a step (RCCL over xGMI on the GPU box, gloo in the
CPU tests).  Nothing here computes.
"""
from __future__ import annotations

import os
from typing import Tuple


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slab [begin, end) of the global batch owned by `rank`; requires even division, because
    batch is baked into a pnnx model file (reference src/pnnx/ir.cpp:597-651) and every rank loads the
    same per-rank-batch model."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def init_process_group(backend: str, device_index: int = None):
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")          # a lone process (SI_BENCH_FORCE_DIST=1) is rank 0 of 1
        os.environ.setdefault("WORLD_SIZE", "1")
        kw = {}
        if backend == "nccl" and device_index is not None:
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend=backend, **kw)
    return dist


def all_gather_slabs(local, gathered=None, group=None):
    """all-gather `local` [b, ...] into `gathered` [world*b, ...] (rank r's slab at rows r*b..); in place into
    a preallocated `gathered` when given.  One collective, bucket = the whole slab."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if world == 1:
        gathered.copy_(local)
        return gathered
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    return gathered


class OverlappedGather:
    """The step's one exchange, taken off the critical path: the all-gather of step i's output slab runs on the
    collective's own stream while step i+1 computes (RCCL kernels only need a handful of CUs; the xGMI links are
    otherwise idle during a forward).

    The engine reuses its output buffer every Forward, so `submit(local)` first copies the slab into one of two staging
    tensors (a device-to-device copy, ~0.1 ms for the 274 MB YOLOv5s slab, waited for before returning so the next
    Forward cannot overwrite it) and then issues `all_gather_into_tensor(..., async_op=True)` from the staging tensor
    into the matching one of two gathered tensors.  `drain()` waits for everything in flight; `latest()` is the last
    completed [world*b, ...] tensor.  Every step's output is gathered -- nothing is skipped, only overlapped."""

    def __init__(self, local_like, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        shape = (self.world * local_like.shape[0],) + tuple(local_like.shape[1:])
        self.stage = [torch.empty_like(local_like) for _ in range(2)]
        self.gathered = [torch.empty(shape, dtype=local_like.dtype, device=local_like.device) for _ in range(2)]
        self.handles = [None, None]
        self.i = 0
        self._cuda = local_like.is_cuda
        self._torch = torch

    def submit(self, local):
        k = self.i & 1
        if self.handles[k] is not None:   # the gather issued two steps ago used these buffers
            self.handles[k].wait()
            self.handles[k] = None
        self.stage[k].copy_(local)
        if self._cuda:
            self._torch.cuda.current_stream().synchronize()   # the slab is safe: the next Forward may overwrite `local`
        if self.world == 1:
            self.gathered[k].copy_(self.stage[k])
        else:
            self.handles[k] = self.dist.all_gather_into_tensor(self.gathered[k], self.stage[k], group=self.group, async_op=True)
        self.i += 1

    # ---- zero-copy variant: the producer writes straight into the staging tensor (Engine::Output binding) --------------
    def target(self):
        """The tensor the NEXT step's output must be written into.  Blocks the host until the gather that last read it
        (two steps ago) has completed."""
        k = self.i & 1
        if self.handles[k] is not None:
            self.handles[k].wait()
            self.handles[k] = None
            if self._cuda:
                self._torch.cuda.current_stream().synchronize()
        return self.stage[k]

    def submit_inplace(self):
        """target() has been filled (the producer has finished): issue its all-gather."""
        k = self.i & 1
        if self.world == 1:
            self.gathered[k].copy_(self.stage[k])
        else:
            self.handles[k] = self.dist.all_gather_into_tensor(self.gathered[k], self.stage[k], group=self.group, async_op=True)
        self.i += 1

    def latest_local(self):
        return self.stage[(self.i - 1) & 1]

    def drain(self):
        for k in (0, 1):
            if self.handles[k] is not None:
                self.handles[k].wait()
                self.handles[k] = None
        if self._cuda:
            self._torch.cuda.synchronize()

    def latest(self):
        """the gathered tensor of the most recent submit (call drain() first)"""
        return self.gathered[(self.i - 1) & 1]


class DeviceArrayView:
    """Zero-copy handle that lets torch wrap an engine-owned device buffer (`__cuda_array_interface__`)."""

    def __init__(self, ptr: int, shape, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


def as_torch(ptr: int, shape, device_index: int):
    import torch
    return torch.as_tensor(DeviceArrayView(ptr, shape), device="cuda:%d" % device_index)
