import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from simpleinfer_amd import _native, hipops
H = _native.hip()
for mb in (0.1, 0.6, 2, 8, 32, 157):
    n = int(mb * 1e6 / 4)
    a = np.ones(n, np.float32)
    d = hipops.DeviceBuffer(a.nbytes)
    ph = C.c_void_p(); H.si_hip_host_alloc(C.byref(ph), a.nbytes)
    pinned = np.ctypeslib.as_array(C.cast(ph, C.POINTER(C.c_float)), shape=(n,))
    def t(fn, reps=5):
        fn(); H.si_hip_device_sync()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        H.si_hip_device_sync()
        return (time.perf_counter() - t0) / reps * 1e3
    direct = t(lambda: (H.si_hip_memcpy_h2d(d.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, None), H.si_hip_stream_sync(None)))
    def staged():
        np.copyto(pinned, a)
        H.si_hip_memcpy_h2d(d.ptr, ph, a.nbytes, None); H.si_hip_stream_sync(None)
    st = t(staged)
    pin_only = t(lambda: (H.si_hip_memcpy_h2d(d.ptr, ph, a.nbytes, None), H.si_hip_stream_sync(None)))
    print("%7.1f MB: pageable direct %.3f ms | host copy to pinned + DMA %.3f ms | pinned DMA only %.3f ms" % (mb, direct, st, pin_only))
