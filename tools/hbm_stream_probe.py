"""What a plain write / copy stream reaches on this box (calibration for the write-bound launches: Detect, the fp16 stem).
usage: python tools/hbm_stream_probe.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from simpleinfer_amd import _native, hipops
H = _native.hip()
ev0, ev1 = C.c_void_p(), C.c_void_p()
H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1))
def t(fn, reps=30):
    for _ in range(5): fn()
    H.si_hip_device_sync(); H.si_hip_event_record(ev0, None)
    for _ in range(reps): fn()
    H.si_hip_event_record(ev1, None); H.si_hip_event_sync(ev1)
    ms = C.c_float(); H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms)); return ms.value / reps
for mb in (26, 105, 209, 274, 548, 1096):
    n = mb * 1000 * 1000
    a, b = hipops.DeviceBuffer(n), hipops.DeviceBuffer(n)
    w = t(lambda: H.si_hip_memset_async(a.ptr, 0, n, None))
    c = t(lambda: H.si_hip_memcpy_d2d(b.ptr, a.ptr, n, None))
    print("%5d MB: memset %.4f ms = %.0f GB/s written;  d2d copy %.4f ms = %.0f GB/s read+written" % (mb, w, n / w / 1e6, c, 2 * n / c / 1e6))
