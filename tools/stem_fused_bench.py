"""YOLOv5s conv_0 + conv_1 with fp16 storage (batch 32, 640x640): the two launches against si_hip_conv2d_stem_s2c32_f16.
usage: python tools/stem_fused_bench.py [batch] [size]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from simpleinfer_amd import _native, hipops
from simpleinfer_amd._native import SiConv2dDesc
H = _native.hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 640
s1, s2 = sz // 2, sz // 4
A = hipops.ACT
d0 = SiConv2dDesc(n, sz, sz, 3, 3, s1, s1, 32, 32, 6, 6, 2, 2, 1, 1, 2, 2, 1, 1, A["silu"], 0, 32, 0, 0.0)
d1 = SiConv2dDesc(n, s1, s1, 32, 32, s2, s2, 64, 64, 3, 3, 2, 2, 1, 1, 1, 1, 1, 1, A["silu"], 0, 64, 0, 0.0)
rng = np.random.default_rng(0)
w0 = (rng.random((32, 3, 6, 6), dtype=np.float32) - 0.5) * 0.3
w1 = (rng.random((64, 32, 3, 3), dtype=np.float32) - 0.5) * 0.1
p0 = np.zeros(H.si_hip_conv2d_stem_f16_weight_elems(C.byref(d0)), np.float16)
H.si_hip_conv2d_stem_f16_pack_weight_host(C.byref(d0), w0.ctypes.data_as(C.c_void_p), p0.ctypes.data_as(C.c_void_p))
p1 = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d1)), np.float16)
H.si_hip_conv2d_f16_pack_weight_host(C.byref(d1), w1.ctypes.data_as(C.c_void_p), p1.ctypes.data_as(C.c_void_p))
dx = hipops.DeviceBuffer.from_numpy(rng.random((n, sz, sz, 3), dtype=np.float32))
dp0, dp1 = hipops.DeviceBuffer.from_numpy(p0), hipops.DeviceBuffer.from_numpy(p1)
db0 = hipops.DeviceBuffer.from_numpy(rng.random(32, dtype=np.float32))
db1 = hipops.DeviceBuffer.from_numpy(rng.random(64, dtype=np.float32))
dm = hipops.DeviceBuffer(n * s1 * s1 * 32 * 2)
dy, dz = hipops.DeviceBuffer(n * s2 * s2 * 64 * 2), hipops.DeviceBuffer(n * s2 * s2 * 64 * 2)
ev0, ev1 = C.c_void_p(), C.c_void_p()
H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1))
def t(fn, reps=50):
    for _ in range(5): fn()
    H.si_hip_device_sync(); H.si_hip_event_record(ev0, None)
    for _ in range(reps): fn()
    H.si_hip_event_record(ev1, None); H.si_hip_event_sync(ev1)
    ms = C.c_float(); H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms)); return ms.value / reps
stem = lambda: H.si_hip_conv2d_stem_f16(C.byref(d0), dx.ptr, dp0.ptr, db0.ptr, dm.ptr, None)
conv = lambda: H.si_hip_conv2d_f16(C.byref(d1), dm.ptr, dp1.ptr, db1.ptr, None, dy.ptr, 0, None)
def both(): stem(); conv()
fused = lambda: H.si_hip_conv2d_stem_s2c32_f16(C.byref(d0), C.byref(d1), dx.ptr, dp0.ptr, db0.ptr, dp1.ptr, db1.ptr, dz.ptr, None)
assert fused() == 0
# the triple: + the first C3's cv1 | cv2 (64 -> 32 + 32) inside the launch (si_hip_conv2d_stem_s2c32_pw_f16)
d2 = SiConv2dDesc(n, s2, s2, 64, 64, s2, s2, 64, 64, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, A["silu"], 0, 64, 0, 0.0)
w2 = (rng.random((64, 64, 1, 1), dtype=np.float32) - 0.5) * 0.2
p2 = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d2)), np.float16)
H.si_hip_conv2d_f16_pack_weight_host(C.byref(d2), w2.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p))
dp2, db2 = hipops.DeviceBuffer.from_numpy(p2), hipops.DeviceBuffer.from_numpy(rng.random(64, dtype=np.float32))
dt = hipops.DeviceBuffer(n * s2 * s2 * 64 * 2)
triple = lambda: H.si_hip_conv2d_stem_s2c32_pw_f16(C.byref(d0), C.byref(d1), C.byref(d2), dx.ptr, dp0.ptr, db0.ptr, dp1.ptr, db1.ptr, dp2.ptr, db2.ptr, dt.ptr, 0, None, 0, None)
assert triple() == 0
for r in range(2):
    print("SI_FUSED_BST=%s: pair %.4f ms   triple %.4f ms   (400 launches each)" % (os.environ.get("SI_FUSED_BST", "1"), t(fused, 400), t(triple, 400)))
if os.environ.get("SI_FUSED_ONLY"):
    sys.exit(0)
for r in range(2):
    a, b, c, f = t(stem), t(conv), t(both), t(fused)
    print("stem %.4f ms  conv_1 %.4f ms  both %.4f ms   fused %.4f ms (%.0f GB/s over %d MB in + %d MB out)" % (
        a, b, c, f, (n * sz * sz * 12 + n * s2 * s2 * 128) / f / 1e6, n * sz * sz * 12 // 1000000, n * s2 * s2 * 128 // 1000000))
both(); fused(); H.si_hip_device_sync()
y, z = dy.to_numpy((n, s2, s2, 64), np.float16), dz.to_numpy((n, s2, s2, 64), np.float16)
print("identical:", bool(np.array_equal(y, z)))
