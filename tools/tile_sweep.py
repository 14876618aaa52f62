#!/usr/bin/env python3
"""tools/tile_sweep.py -- which implicit-GEMM tile is fastest for which launch size?  Times si_hip_conv2d_f32 on every distinct
YOLOv5s / ResNet18 conv shape that the implicit-GEMM kernel serves (not the stem, not the Winograd layers), per batch size and
per tile variant (SiConvPlan::f32_tile in the call's descriptor), sustained (--min-ms per point).  Prints one line per (batch, shape) with the
time of every variant and the winner, then the totals per batch for: the default tile, the best tile per shape, and each
candidate policy.  Development tool (GPU box only); its output is what conv_variant()'s thresholds are read from.

    python tools/tile_sweep.py [--batches 1,4,8,16,32] [--variants 4,11,12,13,14,15] [--min-ms 20] [--model yolov5s]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from conv_bench import conv_shapes  # noqa: E402
from simpleinfer_amd import _native, hipops, modelgen as mg  # noqa: E402
from simpleinfer_amd._native import SiConv2dDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,4,8,16,32")
    ap.add_argument("--variants", default="4,2,11,12,13,14,15")
    ap.add_argument("--min-ms", type=float, default=20.0)
    ap.add_argument("--model", default="yolov5s")
    ap.add_argument("--size", type=int, default=640)
    args = ap.parse_args()
    H = _native.hip()
    variants = [int(v) for v in args.variants.split(",")]
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0))
    H.si_hip_event_create(C.byref(ev1))
    for batch in [int(b) for b in args.batches.split(",")]:
        b = mg.build_yolov5s(batch, args.size) if args.model == "yolov5s" else mg.build_resnet18(batch, 224)
        totals = {v: 0.0 for v in variants}
        best_total = 0.0
        print("== batch %d" % batch)
        print("%-34s %3s %6s %6s  " % ("shape", "cnt", "tiles", "ntile") + " ".join("%8s" % ("v%d" % v) for v in variants) + "   best")
        for key, count in conv_shapes(b).items():
            n, ih, iw, ci, oh, ow, co, k, s, p, g = key
            d = SiConv2dDesc(n, ih, iw, ci, ci, oh, ow, co, co, k[0], k[1], s[0], s[1], 1, 1, p[0], p[1], g, 1, hipops.ACT["silu"], 0, co, 0, 0.0)
            d.plan = None
            name = H.si_hip_conv2d_kernel_name(C.byref(d), C.c_void_p(4096)).decode()
            if "fast" not in name or H.si_hip_conv2d_wino23_preferred(C.byref(d)):
                continue
            wn = H.si_hip_conv2d_weight_elems(C.byref(d))
            rng = np.random.default_rng(0)
            dx = hipops.DeviceBuffer.from_numpy(rng.random((n, ih, iw, ci), dtype=np.float32))
            dw = hipops.DeviceBuffer.from_numpy((rng.random(wn, dtype=np.float32) - 0.5) * 0.1)
            db = hipops.DeviceBuffer.from_numpy(rng.random(co, dtype=np.float32))
            dy = hipops.DeviceBuffer(n * oh * ow * co * 4)
            times = {}
            for v in variants:
                pl = _native.SiConvPlan(f32_tile=v)
                d.plan = C.pointer(pl)
                for _ in range(2):
                    assert H.si_hip_conv2d_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, None) == 0
                H.si_hip_device_sync()
                reps, ms = 10, C.c_float()
                while True:
                    H.si_hip_event_record(ev0, None)
                    for _ in range(reps):
                        H.si_hip_conv2d_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, None)
                    H.si_hip_event_record(ev1, None)
                    H.si_hip_event_sync(ev1)
                    H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
                    if ms.value >= args.min_ms:
                        break
                    reps = int(reps * max(2.0, 1.2 * args.min_ms / max(ms.value, 1e-3)))
                times[v] = ms.value / reps * 1e3
            for buf in (dx, dw, db, dy):
                buf.free()
            M = n * oh * ow
            t64 = ((M + 63) // 64) * ((co // g + 63) // 64) * g
            bv = min(times, key=times.get)
            for v in variants:
                totals[v] += times[v] * count
            best_total += times[bv] * count
            print("%-34s %3d %6d %6d  " % ("%dx%dx%d->%dx%dx%d k%ds%d" % (ih, iw, ci, oh, ow, co, k[0], s[0]), count, t64, (co // g + 63) // 64) +
                  " ".join("%8.1f" % times[v] for v in variants) + "   v%d" % bv, flush=True)
        print("total us/forward: " + " ".join("v%d %.1f" % (v, totals[v]) for v in variants) + "  best-per-shape %.1f" % best_total, flush=True)


if __name__ == "__main__":
    main()
