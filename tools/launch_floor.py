"""tools/launch_floor.py -- per-kernel floor on one stream: N back-to-back launches of a trivial kernel (activation over
4 KB), eager and timed with HIP events.  Tells how much of a small layer's time is the launch itself."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpleinfer_amd import _native, hipops
H = _native.hip()
x = hipops.DeviceBuffer.from_numpy(np.zeros(1024, np.float32)); y = hipops.DeviceBuffer(4096)
ev0, ev1 = C.c_void_p(), C.c_void_p(); H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1))
for n in (1, 10, 100, 1000):
    for _ in range(3): H.si_hip_activation_f32(1, 0.0, x.ptr, 256, 4, 4, y.ptr, 4, None)
    H.si_hip_device_sync(); H.si_hip_event_record(ev0, None)
    for _ in range(n): H.si_hip_activation_f32(1, 0.0, x.ptr, 256, 4, 4, y.ptr, 4, None)
    H.si_hip_event_record(ev1, None); H.si_hip_event_sync(ev1)
    ms = C.c_float(); H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
    print("%5d launches: %.2f us per launch" % (n, ms.value * 1e3 / n))
