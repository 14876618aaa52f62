"""GPU: the C++ drop-in surface itself -- Layer subclasses poked through public fields and SimpleInfer::Engine --
exercised by a C++ program written like the reference's Catch2 layer tests (tests/cpp/test_layers.cpp)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_layer_and_engine_api(gpu, orc, tmp_path):
    from simpleinfer_amd import modelgen as mg
    exe = str(tmp_path / "test_layers")
    pkg = os.path.join(ROOT, "simpleinfer_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wno-ignored-qualifiers", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(pkg, "csrc", "host"), os.path.join(ROOT, "tests", "cpp", "test_layers.cpp"),
                           "-L" + pkg, "-lsimpleinfer_amd", "-lsi_hip", "-Wl,-rpath," + pkg, "-o", exe])
    pp, bp = str(tmp_path / "m.pnnx.param"), str(tmp_path / "m.pnnx.bin")
    mg.build_toy_yolo(2, 64).save(pp, bp)
    x = mg.synth_input((2, 64, 64, 3))
    (name, ref), = orc.run_graph(pp, bp, {"0": x}).items()
    xin, exp = str(tmp_path / "in.f32"), str(tmp_path / "exp.f32")
    x.tofile(xin)
    np.ascontiguousarray(ref, np.float32).tofile(exp)
    r = subprocess.run([exe, pp, bp, xin, exp, name], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-2000:]
    assert " 0 failed" in r.stdout


def test_cpp_bench_harness_builds_and_runs(gpu, tmp_path):
    """examples/bench_yolo.cpp: the reference's benchmark flow (bench/bench_yolo.cpp) through the C++ Engine API."""
    from simpleinfer_amd import modelgen as mg
    exe = str(tmp_path / "bench_yolo")
    pkg = os.path.join(ROOT, "simpleinfer_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "bench_yolo.cpp"),
                           "-L" + pkg, "-lsimpleinfer_amd", "-lsi_hip", "-Wl,-rpath," + pkg, "-o", exe])
    pp, bp = str(tmp_path / "m.pnnx.param"), str(tmp_path / "m.pnnx.bin")
    mg.build_yolov5s(2, 160).save(pp, bp)
    r = subprocess.run([exe, pp, bp, "3"], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr[-1000:])
    assert r.returncode == 0
    assert "host tensors" in r.stdout and "device-resident" in r.stdout and "device memory" in r.stdout
