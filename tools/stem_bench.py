#!/usr/bin/env python3
"""tools/stem_bench.py -- the RGB stem conv (YOLOv5s: 32 x 640 x 640 x 3 -> 32 channels, 6x6 s2 p2) on its three kernels, sustained timing with HIP
events: the fp32 rolling-window kernel (si_hip_conv2d_f32), the fp16 stem kernel (fp16 out) and its split form (si_hip_conv2d_stem_split3_f32, fp32
out).  With SI_HIP_LIB=build_variants/libsi_hip_exp.so, SI_STEM_EXP=1|2|4 ablates the split form (no stores / no MFMA loop / no lo halves staged).
Development tool; not part of the product or the tests."""
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
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--min-ms", type=float, default=300.0)
    ap.add_argument("--which", default="f32,f16,split")
    args = ap.parse_args()
    H = _native.hip()
    n, ih, ci, co, k, s, p = args.batch, args.size, 3, 32, 6, 2, 2
    oh = (ih + 2 * p - k) // s + 1
    d = SiConv2dDesc(n, ih, ih, ci, ci, oh, oh, co, co, k, k, s, s, 1, 1, p, p, 1, 1, hipops.ACT["silu"], 0, co, 0, 0.0)
    rng = np.random.default_rng(0)
    w = ((rng.random((co, ci, k, k), dtype=np.float32) - 0.5) * 0.3)
    dx = hipops.DeviceBuffer.from_numpy(rng.random((n, ih, ih, ci), dtype=np.float32))
    db = hipops.DeviceBuffer.from_numpy(rng.random(co, dtype=np.float32))
    dy = hipops.DeviceBuffer(n * oh * oh * co * 4)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    runs = {}
    p32 = np.zeros(H.si_hip_conv2d_weight_elems(C.byref(d)), np.float32)
    assert H.si_hip_conv2d_pack_weight_host(C.byref(d), vp(w), vp(p32)) == 0
    w32 = hipops.DeviceBuffer.from_numpy(p32)
    runs["f32"] = lambda: H.si_hip_conv2d_f32(C.byref(d), dx.ptr, w32.ptr, db.ptr, None, dy.ptr, None)
    p16 = np.zeros(H.si_hip_conv2d_stem_f16_weight_elems(C.byref(d)), np.float16)
    assert H.si_hip_conv2d_stem_f16_pack_weight_host(C.byref(d), vp(w), vp(p16)) == 0
    w16 = hipops.DeviceBuffer.from_numpy(p16)
    runs["f16"] = lambda: H.si_hip_conv2d_stem_f16(C.byref(d), dx.ptr, w16.ptr, db.ptr, dy.ptr, None)
    ps = np.zeros(H.si_hip_conv2d_stem_split3_weight_elems(C.byref(d)), np.float16)
    assert H.si_hip_conv2d_stem_split3_pack_weight_host(C.byref(d), vp(w), vp(ps)) == 0
    ws = hipops.DeviceBuffer.from_numpy(ps)
    runs["split"] = lambda: H.si_hip_conv2d_stem_split3_f32(C.byref(d), dx.ptr, ws.ptr, db.ptr, dy.ptr, None)
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0))
    H.si_hip_event_create(C.byref(ev1))
    for name in args.which.split(","):
        fn = runs[name]
        assert fn() == 0
        H.si_hip_device_sync()
        reps, ms = 10, C.c_float()
        while True:
            H.si_hip_event_record(ev0, None)
            for _ in range(reps):
                fn()
            H.si_hip_event_record(ev1, None)
            H.si_hip_event_sync(ev1)
            H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
            if ms.value >= args.min_ms:
                break
            reps = int(reps * max(2.0, 1.2 * args.min_ms / max(ms.value, 1e-3)))
        us = 1e3 * ms.value / reps
        out_b = n * oh * oh * co * (2 if name == "f16" else 4)
        print("%-6s exp=%s  %7.1f us   out %.0f MB -> %.0f GB/s (in + out)" % (name, os.environ.get("SI_STEM_EXP", "0"), us, out_b / 1e6,
              (out_b + n * ih * ih * ci * 4) / us / 1e3))


if __name__ == "__main__":
    main()
