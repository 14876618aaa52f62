"""Detect levels of YOLOv5s 640x640 batch 32 with fp16 features: the generic fp16 tiles vs detect_f16_tile_kernel (round 4).
usage: python tools/detect_bench_f16.py [batch]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from simpleinfer_amd import _native, hipops
from simpleinfer_amd._native import SiConv2dDesc, SiYoloLevel
H = _native.hip()
n, na, ne = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 3, 85
rows_total = (80*80 + 40*40 + 20*20) * na
dout = hipops.DeviceBuffer(n * rows_total * ne * 4)
ev0, ev1 = C.c_void_p(), C.c_void_p()
H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1))
off = 0
tot = {0: 0.0, 1: 0.0}
for h, c in ((80, 128), (40, 256), (20, 512)):
    rng = np.random.default_rng(0)
    d = SiConv2dDesc(n, h, h, c, c, h, h, na*ne, na*ne, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, na*ne, 0, 0.0)
    w = (rng.random((na*ne, c, 1, 1), dtype=np.float32) - 0.5) * 0.1
    packed = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d)), np.float16)
    H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p))
    dx = hipops.DeviceBuffer.from_numpy((rng.random((n, h, h, c), dtype=np.float32) - 0.5).astype(np.float16))
    dw = hipops.DeviceBuffer.from_numpy(packed)
    db = hipops.DeviceBuffer.from_numpy(rng.random(na*ne, dtype=np.float32))
    dg = hipops.DeviceBuffer.from_numpy(rng.random((h*h*na, 2), dtype=np.float32))
    da = hipops.DeviceBuffer.from_numpy(rng.random((h*h*na, 2), dtype=np.float32))
    lv = SiYoloLevel(na, ne, rows_total, off, 8.0)
    def t(fn, reps=50):
        for _ in range(5): fn()
        H.si_hip_device_sync(); H.si_hip_event_record(ev0, None)
        for _ in range(reps): fn()
        H.si_hip_event_record(ev1, None); H.si_hip_event_sync(ev1)
        ms = C.c_float(); H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms)); return ms.value / reps
    run = lambda: H.si_hip_conv2d_yolo_f16(C.byref(d), dx.ptr, dw.ptr, db.ptr, C.byref(lv), dg.ptr, da.ptr, dout.ptr, None)
    by = n*h*h*(c*2 + na*ne*4)
    res = []
    for on in (0, 1, 0, 1):
        pl = _native.SiConvPlan(f16_detect_tile=on)
        d.plan = C.pointer(pl)
        ms = t(run); res.append(ms); tot[on] += ms / 2
    print("level %dx%dx%d: generic %.4f / %.4f ms (%.0f GB/s)   detect tile %.4f / %.4f ms (%.0f GB/s)" % (
        h, h, c, res[0], res[2], by/min(res[0], res[2])/1e6, res[1], res[3], by/min(res[1], res[3])/1e6))
    off += h*h*na
print("three levels: generic %.4f ms, detect tile %.4f ms" % (tot[0], tot[1]))
