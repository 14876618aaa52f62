import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, "/root/repo")
from simpleinfer_amd import _native, hipops
from simpleinfer_amd._native import SiConv2dDesc, SiYoloLevel
H = _native.hip()
n, na, ne = 32, 3, 85
rows_total = (80*80 + 40*40 + 20*20) * na
dout = hipops.DeviceBuffer(n * rows_total * ne * 4)
ev0, ev1 = C.c_void_p(), C.c_void_p()
H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1))
off = 0
for h, c in ((80, 128), (40, 256), (20, 512)):
    rng = np.random.default_rng(0)
    d = SiConv2dDesc(n, h, h, c, c, h, h, na*ne, na*ne, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, na*ne, 0, 0.0)
    w = (rng.random((na*ne, c, 1, 1), dtype=np.float32) - 0.5) * 0.1
    packed = np.zeros(H.si_hip_conv2d_weight_elems(C.byref(d)), np.float32)
    H.si_hip_conv2d_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p))
    dx = hipops.DeviceBuffer.from_numpy(rng.random((n, h, h, c), dtype=np.float32))
    dw = hipops.DeviceBuffer.from_numpy(packed)
    db = hipops.DeviceBuffer.from_numpy(rng.random(na*ne, dtype=np.float32))
    dg = hipops.DeviceBuffer.from_numpy(rng.random((h*h*na, 2), dtype=np.float32))
    da = hipops.DeviceBuffer.from_numpy(rng.random((h*h*na, 2), dtype=np.float32))
    dy = hipops.DeviceBuffer(n*h*h*na*ne*4)
    lv = SiYoloLevel(na, ne, rows_total, off, 8.0)
    def t(fn):
        for _ in range(3): fn()
        H.si_hip_device_sync(); H.si_hip_event_record(ev0, None)
        for _ in range(20): fn()
        H.si_hip_event_record(ev1, None); H.si_hip_event_sync(ev1)
        ms = C.c_float(); H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms)); return ms.value / 20
    plain = t(lambda: H.si_hip_conv2d_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, None))
    yolo = t(lambda: H.si_hip_conv2d_yolo_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr, C.byref(lv), dg.ptr, da.ptr, dout.ptr, None))
    fl = 2.0*n*h*h*c*na*ne
    print("level %dx%dx%d: plain conv %.4f ms (%.0f TF/s), conv+decode epilogue %.4f ms (%.0f TF/s)" % (h, h, c, plain, fl/plain/1e9, yolo, fl/yolo/1e9))
    off += h*h*na
