"""CPU: the C-ABI libraries load and export every symbol include/*.h declares; the product has no CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    return sorted(set(re.findall(r"\b(si_[a-z0-9_]+)\s*\(", txt)))


def test_hip_abi_exports_every_declared_symbol(native_libs):
    hip, _ = native_libs
    names = declared("si_hip.h")
    assert len(names) > 35
    for n in names:
        assert hasattr(hip, n), "libsi_hip.so does not export " + n
    # and the python binding table covers the whole header (no entry point left untested / unbound)
    assert sorted(hip._si_signatures) == names


def test_engine_abi_exports_every_declared_symbol(native_libs):
    _, host = native_libs
    names = declared("si_engine.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(host, n), "libsimpleinfer_amd.so does not export " + n
    assert sorted(host._si_signatures) == names


def test_shard_abi_exports_every_declared_symbol(native_libs):
    _, host = native_libs
    names = declared("si_shard.h")
    assert len(names) >= 14
    for n in names:
        assert hasattr(host, n), "libsimpleinfer_amd.so does not export " + n
    assert sorted(host._si_shard_signatures) == names


def test_host_library_exports_cpp_drop_in_api(native_libs):
    """Engine / Tensor / registry C++ symbols of the reference API (include/engine.h, tensor.h) are exported."""
    import subprocess
    from simpleinfer_amd import _native
    out = subprocess.run(["nm", "-DC", "--defined-only", _native.LIB_HOST_PATH], capture_output=True, text=True).stdout
    for sym in ("SimpleInfer::Engine::LoadModel(", "SimpleInfer::Engine::Forward()", "SimpleInfer::Engine::Extract(",
                "SimpleInfer::Engine::Input(", "SimpleInfer::Engine::InputNames", "SimpleInfer::Engine::Release()",
                "SimpleInfer::InitializeContext()", "SimpleInfer::Tensor::Allocate()", "SimpleInfer::GetLayerRegistry(",
                "SimpleInfer::Conv2d_LayerCreator()", "SimpleInfer::YoloDetect_LayerDestroyer(", "pnnx::Graph::load(",
                "pnnx::expand_expression(", "SimpleInfer::ShardedEngine::Init(", "SimpleInfer::ShardedEngine::Forward()",
                "SimpleInfer::ShardedEngine::Gathered("):
        assert sym in out, sym


def test_registry_has_the_reference_type_strings(native_libs):
    from simpleinfer_amd import engine
    have = set(engine.registry_types())
    # reference src/layer_registry.cpp:33-49
    ref = {"nn.AdaptiveAvgPool2d", "nn.BatchNorm2d", "BinaryOp", "torch.cat", "nn.Conv2d", "torch.flatten", "nn.Hardsigmoid",
           "nn.Hardswish", "nn.Linear", "nn.MaxPool2d", "nn.ReLU", "nn.Sigmoid", "nn.SiLU", "nn.Upsample",
           "models.yolo.Detect"}
    assert ref <= have


def test_no_cpu_fallback(native_libs, tmp_path):
    """Without a HIP device LoadModel must fail loudly (kErrorContext) -- never compute on the host."""
    from simpleinfer_amd import Engine, StatusError, Status, device_count, modelgen as mg
    if device_count() > 0:
        pytest.skip("a HIP device is present")
    b = mg.build_toy_classifier(1, 16)
    pp, bp = str(tmp_path / "m.param"), str(tmp_path / "m.bin")
    b.save(pp, bp)
    e = Engine()
    with pytest.raises(StatusError) as ei:
        e.load_model(pp, bp)
    assert ei.value.status == Status.kErrorContext
    with pytest.raises(StatusError):
        e.forward()
    from simpleinfer_amd import hipops
    import numpy as np
    with pytest.raises(hipops.HipError):
        hipops.activation("relu", np.zeros((1, 2, 2, 4), np.float32))


def test_product_does_not_touch_the_oracle():
    """simpleinfer_amd/ must not import, link or open anything under oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "simpleinfer_amd")
    bad = []
    for dp, _, fns in os.walk(pkg):
        if os.path.basename(dp) in ("build", "__pycache__"):
            continue
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"liboracle|si_oracle|from oracle|import oracle|oracle/orc", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
    from simpleinfer_amd import _native
    import subprocess
    for lib in (_native.LIB_HIP_PATH, _native.LIB_HOST_PATH):
        if os.path.exists(lib):
            deps = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
            assert "oracle" not in deps
