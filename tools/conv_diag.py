#!/usr/bin/env python3
"""tools/conv_diag.py -- where a launch of the fp32 implicit-GEMM conv spends its time (development tool).

Needs the DIAGNOSTIC build of the kernel library (per-workgroup s_memtime / s_memrealtime stamps at the phase boundaries,
compiled in only with -DSI_DIAG_STAMPS):

    python tools/conv_diag.py --build          # hipcc ... -DSI_DIAG_STAMPS -> build_variants/libsi_hip_diag.so
    SI_HIP_LIB=build_variants/libsi_hip_diag.so python tools/conv_diag.py --shape 32,80,80,128,256,3,2,1 ...

Per shape: kernel span, in-kernel clock, launch ramp (when the workgroups start), per-phase cycles (prologue = index math +
first loads issued; fill = first tile in LDS; loop; epilogue), MFMA-pipe utilisation inside the loop and over the span.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DIAG = os.path.join(ROOT, "build_variants", "libsi_hip_diag.so")


def build():
    from simpleinfer_amd import build as b
    os.makedirs(os.path.dirname(DIAG), exist_ok=True)
    extra = [d for d in os.environ.get("SI_DIAG_DEFINES", "").split() if d]
    tag = os.environ.get("SI_DIAG_TAG", "diag")
    out = os.path.join(os.path.dirname(DIAG), "libsi_hip_%s.so" % tag)
    b.build_hip(defines=([] if os.environ.get("SI_DIAG_NOSTAMPS") else ["SI_DIAG_STAMPS"]) + extra, out=out, objdir=os.path.join(b.PKG, "build", "hip_" + tag))
    print("built", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--shape", action="append", default=[], help="n,h,w,ci,co,k,s,p")
    ap.add_argument("--warm", type=int, default=1500)
    ap.add_argument("--act", default="silu")
    args = ap.parse_args()
    if args.build:
        build()
        if not args.shape:
            return
    from simpleinfer_amd import _native, hipops
    from simpleinfer_amd._native import SiConv2dDesc
    H = _native.hip()
    H.si_hip_diag_stamps_read.restype = C.c_int
    H.si_hip_diag_stamps_read.argtypes = [C.c_void_p, C.c_size_t]
    H.si_hip_diag_stamps_clear.restype = C.c_int
    shapes = args.shape or ["32,80,80,128,256,3,2,1", "32,40,40,256,512,3,2,1", "32,160,160,64,128,3,2,1", "32,320,320,32,64,3,2,1",
                            "32,40,40,256,256,1,1,0", "32,80,80,128,128,1,1,0", "32,80,80,64,64,1,1,0", "32,20,20,512,512,1,1,0",
                            "32,20,20,1024,512,1,1,0", "32,40,40,128,128,1,1,0"]
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0))
    H.si_hip_event_create(C.byref(ev1))
    for sp in shapes:
        n, h, w, ci, co, k, st, pd = [int(v) for v in sp.split(",")]
        oh, ow = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
        d = SiConv2dDesc(n, h, w, ci, ci, oh, ow, co, co, k, k, st, st, 1, 1, pd, pd, 1, 1, hipops.ACT[args.act], 0, co, 0, 0.0)
        wn = H.si_hip_conv2d_weight_elems(C.byref(d))
        rng = np.random.default_rng(0)
        dx = hipops.DeviceBuffer.from_numpy(rng.random((n, h, w, ci), dtype=np.float32))
        dw = hipops.DeviceBuffer.from_numpy((rng.random(wn, dtype=np.float32) - 0.5) * 0.1)
        db = hipops.DeviceBuffer.from_numpy(rng.random(co, dtype=np.float32))
        dy = hipops.DeviceBuffer(n * oh * ow * co * 4)
        name = H.si_hip_conv2d_kernel_name(C.byref(d), dx.ptr).decode()

        def launch():
            rc = H.si_hip_conv2d_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, None)
            assert rc == 0, rc
        for _ in range(args.warm):
            launch()
        H.si_hip_device_sync()
        reps = 50
        H.si_hip_event_record(ev0, None)
        for _ in range(reps):
            launch()
        H.si_hip_event_record(ev1, None)
        H.si_hip_event_sync(ev1)
        ms = C.c_float()
        H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
        ms = ms.value / reps
        H.si_hip_diag_stamps_clear()
        launch()
        H.si_hip_device_sync()
        M, K = n * oh * ow, k * k * ci
        tiles = ((M + 63) // 64 + 7) // 8 * 8 * ((co + 63) // 64)
        tiles = min(tiles, 65536)
        raw = np.zeros(tiles * 8, np.uint64)
        rc = H.si_hip_diag_stamps_read(raw.ctypes.data_as(C.c_void_p), raw.size)
        assert rc == 0, rc
        s = raw.reshape(tiles, 8)
        s = s[s[:, 6] != 0].astype(np.float64)
        flops = 2.0 * M * K * co
        rt0, rt1 = s[:, 0], s[:, 6]
        span_us = (rt1.max() - rt0.min()) / 100.0
        clock = ((s[:, 5] - s[:, 1]) / ((rt1 - rt0) * 10.0)).mean()
        start = (rt0 - rt0.min()) / 100.0
        pro, fill, loop, epi = s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4]
        nk = K // 32
        mfma_cyc = nk * 16 * 64
        hw = raw.reshape(tiles, 8)[:, 7]
        cu_key = ((hw >> np.uint64(32)) << np.uint64(16)) | ((hw >> np.uint64(8)) & np.uint64(0xff))
        _, per_cu = np.unique(cu_key[raw.reshape(tiles, 8)[:, 6] != 0], return_counts=True)
        med = lambda a: float(np.median(a))
        total_pipe = len(s) * mfma_cyc                        # pipe cycles per SIMD column (one wave of each wg per SIMD)
        span_cyc = span_us * 1e-6 * clock * 1e9
        print("%s  %dx%dx%d->%dx%dx%d k%ds%d  [%s]" % (sp, h, w, ci, oh, ow, co, k, st, name))
        print("   %.4f ms back-to-back = %.1f TF/s | stamped launch: span %.1f us, clock %.2f GHz, %d workgroups on %d CUs (per CU %d..%d)"
              % (ms, flops / ms / 1e9, span_us, clock, len(s), len(per_cu), per_cu.min(), per_cu.max()))
        print("   start of workgroups after the first: p50 %.1f us  p90 %.1f us  max %.1f us" % (med(start), float(np.percentile(start, 90)), start.max()))
        print("   cycles per workgroup (median): prologue %.0f  fill %.0f  loop %.0f (MFMA alone %d = %.0f %% of it)  epilogue %.0f  | total %.0f"
              % (med(pro), med(fill), med(loop), mfma_cyc, 100.0 * mfma_cyc / med(loop), med(epi), med(s[:, 5] - s[:, 1])))
        print("   MFMA pipe busy over the span: %.1f %%   (at 2.4 GHz and 100 %% the launch would take %.1f us)"
              % (100.0 * total_pipe / (256.0 * span_cyc), flops / 157.3e12 * 1e6))
        for b in (dx, dw, db, dy):
            b.free()


if __name__ == "__main__":
    main()
