"""simpleinfer_amd -- MI355X-native drop-in for the forward pass of zpye/SimpleInfer.

The product is two native libraries (HIP kernels + C++ host); this package is the
Python mirror of the reference's public interface (``Engine`` / ``Tensor`` of include/engine.h and
python/pybind11_main.cpp) on top of their C-ABI, plus the pnnx model synthesizer used by the
benchmarks.  Importing it never falls back to a CPU implementation.
"""
from .engine import Engine, Status, StatusError, device_count  # noqa: F401
from . import modelgen  # noqa: F401

__all__ = ["Engine", "Status", "StatusError", "device_count", "modelgen"]
