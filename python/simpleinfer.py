"""simpleinfer -- the reference's Python module, name for name (zpye/SimpleInfer python/pybind11_main.cpp:13-68), on top
of the MI355X engine's C-ABI (include/si_engine.h).  A script written against the reference's pybind11 module runs unchanged
with this directory on PYTHONPATH:

    import simpleinfer as infer
    infer.InitializeContext()
    engine = infer.Engine()
    rc = engine.LoadModel(param, bin)                       # -> infer.Status
    t = infer.Tensor(infer.DataType.Float32, [4, 320, 320, 3])
    rc = t.SetTensorDim4(np.ones([4, 320, 320, 3], np.float32) * 42.0)     # NHWC, borrowed (not copied)
    rc = engine.Input(engine.InputNames()[0], t); rc = engine.Forward()
    out = infer.Tensor(); rc = engine.Extract(engine.OutputNames()[0], out)
    y = out.GetTensorDim4()                                  # numpy view, rank adapted to 4 (include/eigen_helper.h:32-63)

Same conventions as the reference's module: every call returns a Status instead of raising; SetTensorDim4 borrows the
array (Tensor::SetEigenTensor, include/tensor.h:39-52: it only stores the pointer); GetTensorDim4 is a view of the tensor's
memory with the leading dimensions folded -- or padded with 1 -- to exactly four (ToEigenDSize); Extract hands out a
non-owning view of engine memory that the next Forward() overwrites (src/engine_impl.cpp:546-555).
Only what the reference binds is here; the engine's extensions (options, device-resident I/O, profiling) are in the
`simpleinfer_amd` package.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
import sys
from typing import List, Optional, Sequence

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from simpleinfer_amd import _native  # noqa: E402  (loads the two in-tree native libraries; raises if they are not built)


# python/pybind11_main.cpp:18-20 exports exactly these two enumerators of include/types.h:8-22.  `None` is a Python keyword, so --
# as with the pybind11 module -- that member is reached as DataType["None"] or getattr(DataType, "None").
DataType = enum.IntEnum("DataType", {"None": 0, "Float32": 1})


class Status(enum.IntEnum):
    """include/types.h:24-31 under the names of python/pybind11_main.cpp:22-28"""
    Success = 0
    Fail = 1
    Empty = 2
    ErrorShape = 3
    ErrorContext = 4
    Unsupport = 5


def _status(rc: int) -> Status:
    return Status(rc) if 0 <= rc <= 5 else Status.Fail


def InitializeContext() -> None:
    """src/logger.cpp:5-12 (idempotent).  Here: load the native libraries so a missing build fails at this call."""
    _native.host()


def _shape_as(shape: Sequence[int], rank: int) -> List[int]:
    """include/eigen_helper.h:32-63 ToEigenDSize: fold the leading dims into dim 0, or pad in front with 1."""
    shape = [int(s) for s in shape]
    if rank <= len(shape):
        lead = 1
        for s in shape[:len(shape) - rank + 1]:
            lead *= s
        return [lead] + shape[len(shape) - rank + 1:]
    return [1] * (rank - len(shape)) + shape


class Tensor:
    """include/tensor.h:13-69 as the module exposes it (python/pybind11_main.cpp:30-46)."""

    def __init__(self, data_type: DataType = None, shape: Optional[Sequence[int]] = None):
        self._dtype = DataType["None"] if data_type is None else DataType(data_type)
        self._shape = [int(s) for s in (shape or [])]
        self._array: Optional[np.ndarray] = None    # flat float32 view of the bytes (borrowed from numpy or from the engine)
        self._keep = None                           # what keeps those bytes alive

    def GetDataType(self) -> DataType:
        return self._dtype

    def Shape(self) -> List[int]:
        return list(self._shape)

    def SetTensorDim4(self, array) -> Status:
        """Tensor::SetEigenTensor<float, 4> behind pybind11's TensorMap caster: a float32, C-contiguous, rank-4 array is
        borrowed -- the pointer is stored, nothing is copied and the tensor's Shape() is left as constructed."""
        if self._dtype != DataType.Float32:
            return Status.Fail
        if not isinstance(array, np.ndarray) or array.dtype != np.float32 or array.ndim != 4 or not array.flags["C_CONTIGUOUS"]:
            raise TypeError("SetTensorDim4(): incompatible function arguments: expected a C-contiguous float32 array of rank 4")
        self._array = array.reshape(-1)
        self._keep = array
        return Status.Success

    def GetTensorDim4(self) -> np.ndarray:
        """Tensor::GetEigenTensor<float, 4>: a view (never a copy) of the tensor's memory, shape = ToEigenDSize<4>(Shape())."""
        if self._array is None:
            raise RuntimeError("GetTensorDim4(): the tensor holds no data")
        return self._array.reshape(_shape_as(self._shape, 4))


class Engine:
    """include/engine.h:12-38 as the module exposes it (python/pybind11_main.cpp:48-67)."""

    def __init__(self):
        self._L = _native.host()
        h = C.c_void_p()
        if self._L.si_engine_create(C.byref(h)) != 0:
            raise RuntimeError("si_engine_create failed")
        self._h = h
        self._inputs = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.si_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def LoadModel(self, parampath: str, binpath: str) -> Status:
        self._inputs.clear()
        return _status(self._L.si_engine_load_model(self._h, parampath.encode(), binpath.encode()))

    def Release(self) -> Status:
        self._inputs.clear()
        return _status(self._L.si_engine_release(self._h))

    def InputNames(self) -> List[str]:
        return [self._L.si_engine_input_name(self._h, i).decode() for i in range(max(self._L.si_engine_num_inputs(self._h), 0))]

    def OutputNames(self) -> List[str]:
        return [self._L.si_engine_output_name(self._h, i).decode() for i in range(max(self._L.si_engine_num_outputs(self._h), 0))]

    def _operand_shape(self, name: str) -> Optional[List[int]]:
        rank, dims = C.c_int(0), (C.c_int * 8)()
        if self._L.si_engine_operand_shape(self._h, name.encode(), C.byref(rank), dims) != 0:
            return None
        return [dims[i] for i in range(rank.value)]

    def Input(self, name: str, tensor: Tensor) -> Status:
        """Borrows the tensor's buffer; it is read when Forward() runs (src/engine_impl.cpp:522-531)."""
        if tensor._array is None:
            return Status.Fail
        want = self._operand_shape(name)
        if want is None:
            return Status.Fail
        if int(np.prod(want)) != tensor._array.size:
            return Status.ErrorShape
        rc = self._L.si_engine_input(self._h, name.encode(), tensor._array.ctypes.data_as(C.c_void_p), 0)
        if rc == 0:
            self._inputs[name] = tensor._keep    # alive until the next Input / Release, as the contract says
        return _status(rc)

    def Forward(self) -> Status:
        return _status(self._L.si_engine_forward(self._h))

    def Extract(self, name: str, tensor: Tensor) -> Status:
        """`tensor` becomes a non-owning view of the engine's output memory (src/engine_impl.cpp:546-555)."""
        ptr, on_dev = C.c_void_p(), C.c_int(0)
        rc = self._L.si_engine_extract(self._h, name.encode(), C.byref(ptr), C.byref(on_dev))
        if rc != 0:
            return _status(rc)
        shape = self._operand_shape(name)
        if shape is None or on_dev.value or not ptr.value:
            return Status.Fail
        n = int(np.prod(shape))
        tensor._dtype = DataType.Float32
        tensor._shape = shape
        tensor._array = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,))
        tensor._keep = self
        return Status.Success
