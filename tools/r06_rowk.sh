#!/bin/bash
# tools/r06_rowk.sh (GPU box): conv_split3's row K-tiles (3 x 32 channels per barrier) for YOLOv5s conv_1 under f32_split: tests, the layer alone
# (experiment build: SI_SPLIT3_ROWK=0/1), the network against the previous commit's library
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -q -x -k "split3 or stem_split3_random" 2>&1 | tail -3
python3 - <<'PY'
import os, subprocess, sys
code = r'''
import ctypes as C, numpy as np, os
from simpleinfer_amd import _native, hipops
from simpleinfer_amd._native import SiConv2dDesc
H = _native.hip()
n, ih, ic, oc = 32, 320, 32, 64
oh = 160
d = SiConv2dDesc(n, ih, ih, ic, ic, oh, oh, oc, oc, 3, 3, 2, 2, 1, 1, 1, 1, 1, 1, hipops.ACT["silu"], 0, oc, 0, 0.0)
rng = np.random.default_rng(0)
w = ((rng.random((oc, ic, 3, 3), dtype=np.float32) - 0.5) * 0.2)
p = np.zeros(H.si_hip_conv2d_split3_weight_elems(C.byref(d)), np.float16)
assert H.si_hip_conv2d_split3_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p)) == 0
dx = hipops.DeviceBuffer.from_numpy(rng.random((n, ih, ih, ic), dtype=np.float32)); dw = hipops.DeviceBuffer.from_numpy(p)
db = hipops.DeviceBuffer.from_numpy(rng.random(oc, dtype=np.float32)); dy = hipops.DeviceBuffer(n * oh * oh * oc * 4)
fn = lambda: H.si_hip_conv2d_split3_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, None)
assert fn() == 0
ev0, ev1 = C.c_void_p(), C.c_void_p(); H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1))
for r in range(2):
    reps = 1500
    H.si_hip_device_sync(); H.si_hip_event_record(ev0, None)
    for _ in range(reps): fn()
    H.si_hip_event_record(ev1, None); H.si_hip_event_sync(ev1)
    ms = C.c_float(); H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
print("conv_1 under f32_split, SI_SPLIT3_ROWK=%s: %.1f us" % (os.environ.get("SI_SPLIT3_ROWK", "1"), 1e3 * ms.value / reps))
'''
for rep in range(2):
    for v in ("0", "1"):
        env = dict(os.environ, SI_HIP_LIB=os.getcwd() + "/build_variants/libsi_hip_exp.so", SI_SPLIT3_ROWK=v)
        subprocess.run([sys.executable, "-c", code], check=True, env=env)
PY
bash tools/ab_prev.sh "--engine-opt f32_split=1" 3
