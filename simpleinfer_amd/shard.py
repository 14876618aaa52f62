"""ctypes mirror of include/si_shard.h: the node-local rank group (POSIX shared memory rendezvous) and the output
all-gather -- the direct fan-out over IPC-shared HBM, or RCCL's ncclAllGather through the C-ABI (si_rccl_*, librccl.so loaded
with dlopen).  This is the torch-free multi-GPU path: ``bench.py --gpus N`` uses the direct form by default and falls back to
RCCL collectively when the direct path cannot be set up (SI_GATHER_AUTO).

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


GATHER_MODES = {"direct": 0, "rccl": 1, "auto": 2}


def rccl_available() -> bool:
    return bool(_native.host().si_rccl_available())


class RcclComm:
    """include/si_shard.h si_rccl_*: a communicator over the node group, the unique id exchanged through the group."""

    def __init__(self, group: NodeGroup, device: int):
        self._h = _native.host()
        self._c = C.c_void_p()
        _check(self._h.si_rccl_init(group._g, device, C.byref(self._c)), "si_rccl_init")

    def allgather(self, send_ptr: int, recv_ptr: int, bytes_per_rank: int, stream: Optional[int] = None):
        _check(self._h.si_rccl_allgather(self._c, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), bytes_per_rank, stream), "si_rccl_allgather")

    def close(self):
        if self._c:
            c, self._c = self._c, C.c_void_p()
            _check(self._h.si_rccl_destroy(c), "si_rccl_destroy")


class DirectGather:
    """include/si_shard.h SiDirectGather: `slots` gathered buffers [world][slab_bytes] per rank; every step each rank
    pushes its slab into the same slot of every peer with one device-to-device copy per peer (mode "direct"), or all ranks
    run an in-place ncclAllGather into the slot (mode "rccl"); "auto" = direct, RCCL on every rank if that cannot be set up."""

    def __init__(self, group: NodeGroup, device: int, slab_bytes: int, slots: int = 4, mode: str = "direct"):
        self._h = _native.host()
        self.group = group
        self._d = C.c_void_p()
        _check(self._h.si_gather_create_mode(group._g, device, slab_bytes, slots, GATHER_MODES[mode], C.byref(self._d)), "si_gather_create_mode")
        self.slots, self.slab_bytes = slots, slab_bytes
        self.mode = "rccl" if self._h.si_gather_mode(self._d) == 1 else "direct"

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
               "peer_copies": c, "peer_copy_ms": None, "peer_copy_ms_max": None, "gather_gbps_per_peer": None, "peer_landed_ms": None}
        if c:
            # peer_copy_ms: the copy itself (start event on the copy stream -> landed), so slab / that IS the link rate;
            # peer_landed_ms: slab ready -> landed, i.e. the same plus queueing behind earlier copies on that peer's stream
            out["peer_copy_ms"] = round(st.copy_ms_total / c, 4)
            out["peer_landed_ms"] = round(st.landed_ms_total / c, 4)
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

    def __init__(self, engine, output_name: str, group: NodeGroup, device: int, slots: int = 4, mode: str = "direct"):
        self.e, self.oname = engine, output_name
        shape = engine.operand_shape(output_name)
        self.local_shape = tuple(shape)
        self.gather = DirectGather(group, device, int(np.prod(shape)) * 4, slots, mode)
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
