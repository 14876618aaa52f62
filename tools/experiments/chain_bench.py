#!/usr/bin/env python3
"""tools/chain_bench.py [--batches 32,8,4] -- the chained pointwise launch (si_hip_conv2d_chain_f32) against the two launches it replaces
(si_hip_conv2d_split_f32 + si_hip_conv2d_f32), on YOLOv5s' C3 blocks with <= 64 hidden channels.  Sustained timing, us per forward."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpleinfer_amd import _native, hipops  # noqa: E402
from simpleinfer_amd._native import SiConv2dChain, SiConv2dDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="32,8,4")
    ap.add_argument("--min-ms", type=float, default=100.0)
    args = ap.parse_args()
    H = _native.hip()
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1))
    SILU = hipops.ACT["silu"]

    def timed(fn):
        for _ in range(3):
            fn()
        reps, ms = 20, C.c_float()
        while True:
            H.si_hip_event_record(ev0, None)
            for _ in range(reps):
                fn()
            H.si_hip_event_record(ev1, None); H.si_hip_event_sync(ev1)
            H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
            if ms.value >= args.min_ms:
                return ms.value / reps * 1e3
            reps = int(reps * max(2.0, 1.2 * args.min_ms / max(ms.value, 1e-3)))

    for batch in [int(b) for b in args.batches.split(",")]:
        for hw, ic, c in ((160, 64, 32), (80, 128, 64)):
            rng = np.random.default_rng(0)
            n = batch
            x = hipops.DeviceBuffer.from_numpy(rng.random((n, hw, hw, ic), dtype=np.float32))
            w = hipops.DeviceBuffer.from_numpy((rng.random(2 * c * ic, dtype=np.float32) - 0.5) * 0.2)
            b = hipops.DeviceBuffer.from_numpy(rng.random(2 * c, dtype=np.float32))
            w3 = hipops.DeviceBuffer.from_numpy((rng.random(c * c, dtype=np.float32) - 0.5) * 0.2)
            b3 = hipops.DeviceBuffer.from_numpy(rng.random(c, dtype=np.float32))
            ya, yb, z = (hipops.DeviceBuffer(n * hw * hw * c * 4) for _ in range(3))
            d = SiConv2dDesc(n, hw, hw, ic, ic, hw, hw, 2 * c, c, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, SILU, 0, 0, 0, 0.0)
            d3 = SiConv2dDesc(n, hw, hw, c, c, hw, hw, c, c, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, SILU, 0, 0, 0, 0.0)
            ch = SiConv2dChain(w3.ptr, b3.ptr, z.ptr, c, c, SILU)

            def separate():
                assert H.si_hip_conv2d_split_f32(C.byref(d), x.ptr, w.ptr, b.ptr, ya.ptr, c, yb.ptr, c, None) == 0
                assert H.si_hip_conv2d_f32(C.byref(d3), ya.ptr, w3.ptr, b3.ptr, None, z.ptr, None) == 0

            def chained():
                assert H.si_hip_conv2d_chain_f32(C.byref(d), x.ptr, w.ptr, b.ptr, ya.ptr, c, yb.ptr, c, C.byref(ch), None) == 0

            ts, tc = timed(separate), timed(chained)
            ts2, tc2 = timed(separate), timed(chained)
            print("batch %2d  %3dx%3dx%3d -> %d|%d -> %d : separate %.1f / %.1f us  chained %.1f / %.1f us" % (batch, hw, hw, ic, c, c, c, ts, ts2, tc, tc2))
            for buf in (x, w, b, w3, b3, ya, yb, z):
                buf.free()


if __name__ == "__main__":
    main()
