#!/usr/bin/env python3
"""tools/slab_diag.py -- phase stamps of the one-shot slab kernels (conv_slab_f16.hip), from the diagnostic build
(tools/hip_variant.sh conv_slab_f16.hip slab_diag SI_DIAG_STAMPS; SI_HIP_LIB=build_variants/libsi_hip_slab_diag.so).
Per shape: back-to-back time, then over the workgroups of one stamped launch (after a sustained warm-up) the in-kernel clock
(cycle stamps / realtime stamps) and the median cycles of each phase: start -> patch in LDS -> K loop done -> staged -> stored."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleinfer_amd import _native, hipops  # noqa: E402
from simpleinfer_amd._native import SiConv2dDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", action="append", default=[])
    ap.add_argument("--warm-ms", type=float, default=300.0)
    args = ap.parse_args()
    H = _native.hip()
    rd, cl = H.si_hip_diag_stamps_read_slab, H.si_hip_diag_stamps_clear_slab
    rd.restype = C.c_int
    rd.argtypes = [C.c_void_p, C.c_size_t]
    cl.restype = C.c_int
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0))
    H.si_hip_event_create(C.byref(ev1))
    for sp in args.shape or ["32,40,40,128,128", "32,20,20,256,256"]:
        n, h, w, ci, co = [int(v) for v in sp.split(",")][:5]
        d = SiConv2dDesc(n, h, w, ci, ci, h, w, co, co, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, hipops.ACT["silu"], 0, co, 0, 0.0)
        rng = np.random.default_rng(0)
        wn = H.si_hip_conv2d_f16_weight_elems(C.byref(d))
        w32 = ((rng.random((co, ci, 3, 3), dtype=np.float32) - 0.5) * 0.1)
        packed = np.zeros(wn, np.float16)
        assert H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), w32.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)) == 0
        dx = hipops.DeviceBuffer.from_numpy(rng.random((n, h, w, ci), dtype=np.float32).astype(np.float16))
        dw = hipops.DeviceBuffer.from_numpy(packed)
        db = hipops.DeviceBuffer.from_numpy(rng.random(co, dtype=np.float32))
        dy = hipops.DeviceBuffer(n * h * w * co * 2)

        def fn():
            return H.si_hip_conv2d_f16(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, 0, None)
        assert fn() == 0
        H.si_hip_device_sync()
        reps, ms = 50, C.c_float()
        while True:
            H.si_hip_event_record(ev0, None)
            for _ in range(reps):
                fn()
            H.si_hip_event_record(ev1, None)
            H.si_hip_event_sync(ev1)
            H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
            if ms.value >= args.warm_ms:
                break
            reps *= 2
        cl()
        fn()
        H.si_hip_device_sync()
        raw = np.zeros(65536 * 8, np.uint64)
        assert rd(raw.ctypes.data_as(C.c_void_p), raw.size) == 0
        st = raw.reshape(-1, 8)
        st = st[st[:, 1] != 0]
        cyc = (st[:, 5] - st[:, 1]).astype(np.float64)
        rt = (st[:, 6] - st[:, 0]).astype(np.float64) * 10.0   # 100 MHz ticks -> ns
        clock = np.median(cyc / rt)
        ph = [np.median((st[:, i + 1] - st[:, i]).astype(np.float64)) for i in range(1, 5)]
        span = (st[:, 6].max() - st[:, 0].min()) * 10.0
        print("%s: %.2f us back-to-back; stamped launch: %d workgroups, span %.1f us, workgroup life %.0f cycles = %.2f us, clock %.2f GHz"
              % (sp, ms.value / reps * 1e3, len(st), span / 1e3, np.median(cyc), np.median(rt) / 1e3, clock))
        print("   cycles: prologue (patch -> LDS) %.0f | K loop %.0f | epilogue math + staging %.0f | read-back + stores %.0f" % tuple(ph))
        for b in (dx, dw, db, dy):
            b.free()


if __name__ == "__main__":
    main()
