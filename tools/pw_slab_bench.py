#!/usr/bin/env python3
"""tools/pw_slab_bench.py -- the C3 bottleneck's two convs with fp16 storage (1x1 c -> c SiLU, 3x3 c -> c SiLU + shortcut): two launches
(si_hip_conv2d_f16 twice) against one (si_hip_conv2d_pw_slab_f16), each timed as a replayed hipGraph of 40 pairs, sustained, interleaved."""
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
    ap.add_argument("--min-ms", type=float, default=300.0)
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    H = _native.hip()
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0))
    H.si_hip_event_create(C.byref(ev1))
    st = C.c_void_p()
    assert H.si_hip_stream_create(C.byref(st)) == 0
    for sp in args.shape or ["32,40,40,128", "32,20,20,256"]:
        n, h, w, c = [int(v) for v in sp.split(",")]
        rng = np.random.default_rng(0)
        d0 = SiConv2dDesc(n, h, w, c, c, h, w, c, c, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, hipops.ACT["silu"], 0, c, 0, 0.0)
        d1 = SiConv2dDesc(n, h, w, c, c, h, w, c, c, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, hipops.ACT["silu"], 1, c, 0, 0.0)
        assert H.si_hip_conv2d_pw_slab_f16_supported(C.byref(d0), C.byref(d1)) >= 1

        def pack(d, shape):
            wts = ((rng.random(shape, dtype=np.float32) - 0.5) * 0.1)
            p = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d)), np.float16)
            assert H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), wts.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p)) == 0
            return hipops.DeviceBuffer.from_numpy(p)
        p0, p1 = pack(d0, (c, c, 1, 1)), pack(d1, (c, c, 3, 3))
        dx = hipops.DeviceBuffer.from_numpy(rng.random((n, h, w, c), dtype=np.float32).astype(np.float16))
        b0, b1 = hipops.DeviceBuffer.from_numpy(rng.random(c, dtype=np.float32)), hipops.DeviceBuffer.from_numpy(rng.random(c, dtype=np.float32))
        dm, dy = hipops.DeviceBuffer(n * h * w * c * 2), hipops.DeviceBuffer(n * h * w * c * 2)

        def two(s):
            assert H.si_hip_conv2d_f16(C.byref(d0), dx.ptr, p0.ptr, b0.ptr, None, dm.ptr, 0, s) == 0
            assert H.si_hip_conv2d_f16(C.byref(d1), dm.ptr, p1.ptr, b1.ptr, dx.ptr, dy.ptr, 0, s) == 0

        def one(s):
            assert H.si_hip_conv2d_pw_slab_f16(C.byref(d0), C.byref(d1), dx.ptr, p0.ptr, b0.ptr, p1.ptr, b1.ptr, dx.ptr, dy.ptr, s) == 0

        def timed(fn):
            gx = C.c_void_p()
            assert H.si_hip_graph_begin_capture(st) == 0
            for _ in range(40):
                fn(st)
            assert H.si_hip_graph_end_capture(st, C.byref(gx)) == 0
            reps, ms = 2, C.c_float()
            while True:
                H.si_hip_event_record(ev0, st)
                for _ in range(reps):
                    H.si_hip_graph_launch(gx, st)
                H.si_hip_event_record(ev1, st)
                H.si_hip_event_sync(ev1)
                H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
                if ms.value >= args.min_ms:
                    break
                reps = int(reps * max(2.0, 1.2 * args.min_ms / max(ms.value, 1e-3)))
            H.si_hip_graph_destroy(gx)
            return ms.value / (reps * 40) * 1e3
        for r in range(args.rounds):
            t2, t1 = timed(two), timed(one)
            print("%-16s round %d: two launches %.2f us, one launch %.2f us (%.2fx)" % (sp, r, t2, t1, t2 / t1))


if __name__ == "__main__":
    main()
