"""ctypes mirror of include/si_shard.h: the node-local rank group (POSIX shared memory rendezvous) and the direct
output all-gather over IPC-shared HBM.  This is the torch-free multi-GPU path: ``bench.py --gpus N`` uses it by default
and falls back to RCCL (``simpleinfer_amd.distributed``) only when the direct path cannot be set up.

Nothing here computes; the GPU is touched only by si_gather_* (hipMalloc / hipIpc* / device-to-device copies).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _native


class ShardError(RuntimeError):
    def __init__(self, what: str, code: int):
        super().__init__("%s failed with code %d" % (what, code))
        self.code = code


def _check(rc: int, what: str):
    if rc != 0:
        raise ShardError(what, rc)


def default_group_name() -> str:
    """A name every rank of one launch derives identically and no other launch shares: the launcher's pid (the parent of
    all ranks under torch.distributed.run and simpleinfer_amd.launch) + the rendezvous port."""
    return "/si_%d_%d_%s" % (os.getuid(), os.getppid(), os.environ.get("MASTER_PORT", "0"))


class NodeGroup:
    """Ranks of one node (include/si_shard.h SiNodeGroup).  Host only."""

    def __init__(self, name: str, rank: int, world: int, timeout_s: float = 60.0):
        self._h = _native.host()
        self._g = C.c_void_p()
        _check(self._h.si_group_create(name.encode(), rank, world, float(timeout_s), C.byref(self._g)), "si_group_create")
        self.rank, self.world = rank, world

    def barrier(self):
        _check(self._h.si_group_barrier(self._g), "si_group_barrier")

    def allgather_bytes(self, mine: bytes) -> list:
        n = len(mine)
        out = C.create_string_buffer(n * self.world)
        _check(self._h.si_group_allgather(self._g, mine, n, out), "si_group_allgather")
        return [out.raw[r * n:(r + 1) * n] for r in range(self.world)]

    def allgather_f64(self, value: float) -> np.ndarray:
        parts = self.allgather_bytes(np.float64(value).tobytes())
        return np.frombuffer(b"".join(parts), np.float64).copy()

    def max_f64(self, value: float) -> float:
        return float(self.allgather_f64(value).max())

    def close(self):
        if self._g:
            self._h.si_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DirectGather:
    """include/si_shard.h SiDirectGather: `slots` gathered buffers [world][slab_bytes] per rank; every step each rank
    pushes its slab into the same slot of every peer with one device-to-device copy per peer."""

    def __init__(self, group: NodeGroup, device: int, slab_bytes: int, slots: int = 4):
        self._h = _native.host()
        self.group = group
        self._d = C.c_void_p()
        _check(self._h.si_gather_create(group._g, device, slab_bytes, slots, C.byref(self._d)), "si_gather_create")
        self.slots, self.slab_bytes = slots, slab_bytes

    def slab_ptr(self, slot: int) -> int:
        return int(self._h.si_gather_slab(self._d, slot) or 0)

    def buffer_ptr(self, slot: int) -> int:
        return int(self._h.si_gather_buffer(self._d, slot) or 0)

    def push(self, slot: int, producer_stream: Optional[int] = None):
        _check(self._h.si_gather_push(self._d, slot, producer_stream), "si_gather_push")

    def complete(self, slot: int):
        _check(self._h.si_gather_complete(self._d, slot), "si_gather_complete")

    def stats(self, reset: bool = False) -> dict:
        """si_gather_stats: where the steps' time went since the last reset (host waits in si_gather_complete, device time of
        the peer copies) -- per-step / per-copy averages and the achieved per-peer copy rate"""
        st = _native.SiGatherStats()
        _check(self._h.si_gather_stats(self._d, C.byref(st), 1 if reset else 0), "si_gather_stats")
        n, c = max(int(st.completes), 1), int(st.copies)
        out = {"gather_wait_copies_ms": round(st.wait_copies_ms_total / n, 4), "gather_wait_barrier_ms": round(st.wait_barrier_ms_total / n, 4),
               "gather_wait_ms": round((st.wait_copies_ms_total + st.wait_barrier_ms_total) / n, 4), "completes": int(st.completes),
               "peer_copies": c, "peer_copy_ms": None, "peer_copy_ms_max": None, "gather_gbps_per_peer": None}
        if c:
            out["peer_copy_ms"] = round(st.copy_ms_total / c, 4)
            out["peer_copy_ms_max"] = round(st.copy_ms_max, 4)
            out["gather_gbps_per_peer"] = round(self.slab_bytes / (st.copy_ms_total / c * 1e-3) / 1e9, 2)
        return out

    def close(self):
        if self._d:
            d, self._d = self._d, C.c_void_p()
            _check(self._h.si_gather_destroy(d), "si_gather_destroy")


class ShardedForward:
    """The step of the sharded path (the Python twin of SimpleInfer::ShardedEngine, include/shard.h): Forward() into this
    step's slot, start the fan-out, complete the previous step's gather."""

    def __init__(self, engine, output_name: str, group: NodeGroup, device: int, slots: int = 4):
        self.e, self.oname = engine, output_name
        shape = engine.operand_shape(output_name)
        self.local_shape = tuple(shape)
        self.gather = DirectGather(group, device, int(np.prod(shape)) * 4, slots)
        self.step = 0
        self.pending = -1
        self.completed = -1

    def forward(self):
        slot = self.step % self.gather.slots
        self.e.bind_output(self.oname, self.gather.slab_ptr(slot))
        self.e.forward()                                  # synchronous: the slab is complete
        self.gather.push(slot, self.e.stream())
        self.step += 1
        prev, self.pending = self.pending, slot
        if self.gather.slots == 1:
            return self.flush()
        if prev >= 0:
            self.gather.complete(prev)
            self.completed = prev

    def flush(self):
        if self.pending >= 0:
            self.gather.complete(self.pending)
            self.completed, self.pending = self.pending, -1

    def gathered_ptr(self) -> int:
        assert self.completed >= 0, "no completed gather yet"
        return self.gather.buffer_ptr(self.completed)

    def gathered_shape(self):
        return (self.local_shape[0] * self.gather.group.world,) + self.local_shape[1:]

    def close(self):
        self.e.bind_output(self.oname, None)
        self.gather.close()
