"""Python mirror of SimpleInfer::Engine (reference include/engine.h:12-38) over include/si_engine.h.

numpy arrays are NHWC float32, exactly what the reference's pybind11 module exchanges
(python/pybind11_main.cpp:28-49: SetTensorDim4 / GetTensorDim4).
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _native


class Status(enum.IntEnum):
    kSuccess = 0
    kFail = 1
    kEmpty = 2
    kErrorShape = 3
    kErrorContext = 4
    kUnsupport = 5


class StatusError(RuntimeError):
    def __init__(self, what: str, status: int):
        self.status = Status(status) if 0 <= status <= 5 else status
        super().__init__("%s -> %s" % (what, getattr(self.status, "name", self.status)))


def device_count() -> int:
    n = C.c_int(0)
    _native.hip().si_hip_device_count(C.byref(n))
    return n.value


class Engine:
    """LoadModel / Input / Forward / Extract with the reference's semantics; methods raise StatusError
    instead of returning a non-success Status."""

    def __init__(self, **options: int):
        self._L = _native.host()
        h = C.c_void_p()
        rc = self._L.si_engine_create(C.byref(h))
        if rc != 0:
            raise StatusError("si_engine_create", rc)
        self._h = h
        self._inputs: Dict[str, np.ndarray] = {}  # keep borrowed host buffers alive
        for k, v in options.items():
            self.set_option(k, int(v))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.si_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _check(self, what: str, rc: int):
        if rc != 0:
            raise StatusError(what, rc)

    def set_option(self, key: str, value: int):
        self._check("SetOption(%s)" % key, self._L.si_engine_set_option(self._h, key.encode(), value))

    def load_model(self, param_path: str, bin_path: str):
        self._check("LoadModel", self._L.si_engine_load_model(self._h, param_path.encode(), bin_path.encode()))

    def release(self):
        self._check("Release", self._L.si_engine_release(self._h))   # (first: Release is where a pinned buffer is let go)
        self._inputs.clear()

    def input_names(self) -> List[str]:
        return [self._L.si_engine_input_name(self._h, i).decode() for i in range(self._L.si_engine_num_inputs(self._h))]

    def output_names(self) -> List[str]:
        return [self._L.si_engine_output_name(self._h, i).decode() for i in range(self._L.si_engine_num_outputs(self._h))]

    def operand_shape(self, name: str) -> Tuple[int, ...]:
        rank = C.c_int()
        dims = (C.c_int * 8)()
        self._check("OperandShape(%s)" % name, self._L.si_engine_operand_shape(self._h, name.encode(), C.byref(rank), dims))
        return tuple(dims[i] for i in range(rank.value))

    def input(self, name: str, array: np.ndarray):
        """Borrow a host NHWC float32 array (read at forward time, like Engine::Input)."""
        shape = self.operand_shape(name)
        a = np.ascontiguousarray(array, dtype=np.float32)
        if a.size != int(np.prod(shape)):
            raise StatusError("Input(%s): %s does not match %s" % (name, a.shape, shape), Status.kErrorShape)
        # hand the new buffer over BEFORE the previous array is dropped: Input() is where the engine lets go of the old one
        # (with "pin_inputs" it is still registered with the driver until then)
        self._check("Input(%s)" % name, self._L.si_engine_input(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), 0))
        self._inputs[name] = a

    def input_device(self, name: str, device_ptr: int):
        """Device-resident input: the engine reads the buffer in place (no H2D in forward)."""
        self._check("Input(%s)" % name, self._L.si_engine_input(self._h, name.encode(), C.c_void_p(device_ptr), 1))

    def bind_output(self, name: str, device_ptr: Optional[int]):
        """Engine::Output: write output `name` into caller-owned device memory from the next forward on (None restores
        the engine's own buffer)."""
        self._check("Output(%s)" % name, self._L.si_engine_bind_output(self._h, name.encode(), C.c_void_p(device_ptr) if device_ptr else None))

    def forward(self):
        self._check("Forward", self._L.si_engine_forward(self._h))

    def forward_async(self):
        """Engine::ForwardAsync: enqueue the launches and return; sync() waits (forward() = both)."""
        self._check("ForwardAsync", self._L.si_engine_forward_async(self._h))

    def sync(self):
        self._check("Sync", self._L.si_engine_sync(self._h))

    def extract_ptr(self, name: str) -> Tuple[int, bool]:
        p = C.c_void_p()
        on_dev = C.c_int()
        self._check("Extract(%s)" % name, self._L.si_engine_extract(self._h, name.encode(), C.byref(p), C.byref(on_dev)))
        return p.value, bool(on_dev.value)

    def extract(self, name: str, copy: bool = True) -> np.ndarray:
        """Host view (or copy) of an output; like Engine::Extract the view dies at the next forward."""
        shape = self.operand_shape(name)
        ptr, on_dev = self.extract_ptr(name)
        n = int(np.prod(shape))
        if on_dev:
            out = np.empty(shape, np.float32)
            H = _native.hip()
            rc = H.si_hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 4, self.stream())
            rc = rc or H.si_hip_stream_sync(self.stream())
            if rc != 0:
                raise StatusError("Extract d2h: " + H.si_hip_error_string(rc).decode(), Status.kFail)
            return out
        view = np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), C.POINTER(C.c_float)), shape=(n,)).reshape(shape)
        return view.copy() if copy else view

    def stream(self) -> Optional[int]:
        return self._L.si_engine_stream(self._h)

    def last_forward_ms(self) -> float:
        return float(self._L.si_engine_last_forward_ms(self._h))

    def profile(self) -> List[dict]:
        n = self._L.si_engine_profile(self._h)
        if n < 0:
            raise StatusError("Profile", -n)
        out = []
        for i in range(n):
            name, typ, kern = C.c_char_p(), C.c_char_p(), C.c_char_p()
            ms, fl, by = C.c_float(), C.c_double(), C.c_double()
            self._L.si_engine_profile_entry(self._h, i, C.byref(name), C.byref(typ), C.byref(kern), C.byref(ms),
                                            C.byref(fl), C.byref(by))
            out.append(dict(name=name.value.decode(), type=typ.value.decode(), kernel=kern.value.decode(),
                            ms=ms.value, flops=fl.value, bytes=by.value))
        return out

    def schedule(self) -> Dict[str, List[str]]:
        n = self._L.si_engine_schedule(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self._L.si_engine_schedule(self._h, buf, n + 1)
        out: Dict[str, List[str]] = {"run": [], "fused": [], "alias": [], "split_demoted": []}
        for ln in buf.value.decode().splitlines():
            k, v = ln.split(" ", 1)
            if k in ("arena_bytes", "per_operand_bytes", "lanes", "split_reruns"):
                out[k] = int(v)      # HBM held for intermediates: shared by lifetime / one allocation per operand
            else:
                out[k].append(v)
        return out


def pnnx_dump(param_path: str, bin_path: str, expand: bool, out_path: str):
    rc = _native.host().si_pnnx_dump(param_path.encode(), bin_path.encode(), 1 if expand else 0, out_path.encode())
    if rc != 0:
        raise RuntimeError("si_pnnx_dump rc=%d" % rc)


def pnnx_save(param_path: str, bin_path: str, out_param_path: str, out_bin_path: str, expand: bool = False, batch: int = 0):
    """Rewrite a model through the C++ loader + writer (Graph::save): optional pnnx.Expression lowering and re-batching."""
    rc = _native.host().si_pnnx_save(param_path.encode(), bin_path.encode(), 1 if expand else 0, int(batch),
                                     out_param_path.encode(), out_bin_path.encode())
    if rc != 0:
        raise RuntimeError("si_pnnx_save rc=%d" % rc)


def registry_types() -> List[str]:
    L = _native.host()
    n = L.si_registry_types(None, 0)
    buf = C.create_string_buffer(n + 1)
    L.si_registry_types(buf, n + 1)
    return buf.value.decode().split()
