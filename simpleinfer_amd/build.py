"""In-tree build of the two native libraries (no cmake needed; hipcc + g++ directly).

  libsi_hip.so           hand-written gfx950 HIP kernels behind the C-ABI of include/si_hip.h
  libsimpleinfer_amd.so  C++17 host: Engine / Tensor / Layer / LayerRegistry / pnnx loader and the
                         C-ABI of include/si_engine.h; links libsi_hip.so, includes no HIP header

Both land next to this file so they travel with the source snapshot.  ``python -m
simpleinfer_amd.build`` rebuilds what is stale; ``--force`` rebuilds everything.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
INC = os.path.join(ROOT, "include")
HIP_DIR = os.path.join(PKG, "csrc", "hip")
HOST_DIR = os.path.join(PKG, "csrc", "host")
LIB_HIP = os.path.join(PKG, "libsi_hip.so")
LIB_HOST = os.path.join(PKG, "libsimpleinfer_amd.so")
ARCH = "gfx950"


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _stale(target, deps):
    return (not os.path.exists(target)) or os.path.getmtime(target) < _newest(deps)


def hip_sources():
    return sorted(glob.glob(os.path.join(HIP_DIR, "*.hip")))


def host_sources():
    return sorted(glob.glob(os.path.join(HOST_DIR, "*.cpp")) + glob.glob(os.path.join(HOST_DIR, "pnnx", "*.cpp")) +
                  glob.glob(os.path.join(HOST_DIR, "layer", "*.cpp")))


def _headers():
    return (glob.glob(os.path.join(INC, "*.h")) + glob.glob(os.path.join(HIP_DIR, "*.h")) +
            glob.glob(os.path.join(HOST_DIR, "*.h")) + glob.glob(os.path.join(HOST_DIR, "*", "*.h")))


def _run(cmd):
    print("+ " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_hip(force=False, verbose_resources=False, defines=(), out=None, objdir=None):
    """One object per .hip file (compiled in parallel, rebuilt only when stale), then one link.  `defines` / `out` /
    `objdir` build a variant (diagnostic stamps, experiment switches) next to the product library without touching it."""
    srcs = hip_sources()
    out = out or LIB_HIP
    objdir = objdir or os.path.join(PKG, "build", "hip")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_time = _newest(_headers())
    base = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INC, "-I" + HIP_DIR] + ["-D" + d for d in defines]
    if verbose_resources:
        base.append("-Rpass-analysis=kernel-resource-usage")
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            cmd = base + ["-c", s, "-o", o]
            print("+ " + " ".join(cmd), flush=True)
            procs.append(subprocess.Popen(cmd))
            if len(procs) >= 6:
                for p in procs:
                    if p.wait() != 0:
                        raise subprocess.CalledProcessError(p.returncode, p.args)
                procs = []
    for p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, p.args)
    if force or _stale(out, objs):
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", out])
    return out


def build_host(force=False):
    srcs = host_sources()
    if not force and not _stale(LIB_HOST, srcs + _headers() + [LIB_HIP]):
        return LIB_HOST
    cxx = os.environ.get("CXX", "g++")
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = _newest(_headers())
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.relpath(s, HOST_DIR).replace(os.sep, "_") + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            cmd = [cxx, "-std=c++17", "-O2", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-ignored-qualifiers", "-I" + INC,
                   "-I" + HOST_DIR, "-c", s, "-o", o]
            print("+ " + " ".join(cmd), flush=True)
            procs.append(subprocess.Popen(cmd))
            if len(procs) >= 8:
                for p in procs:
                    if p.wait() != 0:
                        raise subprocess.CalledProcessError(p.returncode, p.args)
                procs = []
    for p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, p.args)
    _run([cxx, "-shared", "-o", LIB_HOST] + objs + ["-L" + PKG, "-lsi_hip", "-Wl,-rpath,$ORIGIN"])
    return LIB_HOST


def pybind_target():
    import sysconfig
    return os.path.join(ROOT, "python", "simpleinfer_pybind" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pybind(force=False):
    """python/simpleinfer_pybind*.so: the reference's pybind11 module (python/pybind11_main.cpp:13-68) compiled over this repo's C++ Engine /
    Tensor.  Needs the pybind11 headers (the wheel) and Python.h; returns None -- the ctypes mirror python/simpleinfer.py serves the same
    surface -- when either is missing."""
    try:
        import pybind11
        import sysconfig
    except ImportError:
        return None
    inc_py = sysconfig.get_paths().get("include") or ""
    if not os.path.exists(os.path.join(inc_py, "Python.h")):
        return None
    src = os.path.join(ROOT, "python", "pybind11_main.cpp")
    out = pybind_target()
    if not force and not _stale(out, [src, LIB_HOST] + glob.glob(os.path.join(INC, "*.h"))):
        return out
    cxx = os.environ.get("CXX", "g++")
    _run([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + INC, "-I" + pybind11.get_include(), "-I" + inc_py, src, "-o", out,
          "-L" + PKG, "-lsimpleinfer_amd", "-Wl,-rpath,$ORIGIN/../simpleinfer_amd"])
    return out


def build_all(force=False):
    build_hip(force)
    build_host(force)
    build_pybind(force)
    return LIB_HIP, LIB_HOST


def build_experiment():
    """build_variants/libsi_hip_exp.so: the kernel library compiled with -DSI_EXPERIMENT -- the ONLY build in which the SI_CONV_* / SI_WINO_* /
    SI_DETECT_* / SI_SPLIT3_* environment switches of the sweep / ablation scripts under tools/ exist (the product library reads no environment
    variable and has no process-global setter: kernel-form choices travel in SiConv2dDesc::plan).  Select it with SI_HIP_LIB=<path>."""
    vdir = os.path.join(ROOT, "build_variants")
    return build_hip(defines=("SI_EXPERIMENT",), out=os.path.join(vdir, "libsi_hip_exp.so"), objdir=os.path.join(vdir, "obj_exp"))


if __name__ == "__main__":
    if "--experiment" in sys.argv:
        print("built:", build_experiment())
    else:
        build_all(force="--force" in sys.argv)
        print("built:", LIB_HIP, LIB_HOST)
