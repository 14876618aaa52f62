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
    ap.add_argument("--timeline", type=int, default=0, help="print the phase stamps of every workgroup of the N-th CU")
    ap.add_argument("--algo", default="direct", choices=["direct", "wino"], help="wino: the fused Winograd F(2,3) kernel (3x3 s1 shapes)")
    args = ap.parse_args()
    if args.build:
        build()
        if not args.shape:
            return
    from simpleinfer_amd import _native, hipops
    from simpleinfer_amd._native import SiConv2dDesc
    H = _native.hip()
    wino = args.algo == "wino"
    stamps_read = H.si_hip_diag_stamps_read_wino if wino else H.si_hip_diag_stamps_read
    stamps_clear = H.si_hip_diag_stamps_clear_wino if wino else H.si_hip_diag_stamps_clear
    stamps_read.restype = C.c_int
    stamps_read.argtypes = [C.c_void_p, C.c_size_t]
    stamps_clear.restype = C.c_int
    if wino and not args.shape:
        args.shape = ["32,80,80,64,64,3,1,1", "32,40,40,128,128,3,1,1", "32,20,20,256,256,3,1,1", "32,160,160,32,32,3,1,1"]
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
        wn = H.si_hip_conv2d_wino23_weight_elems(C.byref(d)) if wino else H.si_hip_conv2d_weight_elems(C.byref(d))
        conv_fn = H.si_hip_conv2d_wino23_f32 if wino else H.si_hip_conv2d_f32
        rng = np.random.default_rng(0)
        dx = hipops.DeviceBuffer.from_numpy(rng.random((n, h, w, ci), dtype=np.float32))
        dw = hipops.DeviceBuffer.from_numpy((rng.random(wn, dtype=np.float32) - 0.5) * 0.1)
        db = hipops.DeviceBuffer.from_numpy(rng.random(co, dtype=np.float32))
        dy = hipops.DeviceBuffer(n * oh * ow * co * 4)
        name = "conv_wino23_kernel" if wino else H.si_hip_conv2d_kernel_name(C.byref(d), dx.ptr).decode()

        def launch():
            rc = conv_fn(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, None)
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
        stamps_clear()
        launch()
        H.si_hip_device_sync()
        M, K = n * oh * ow, k * k * ci
        tiles = 65536   # (every stamp slot: the tile shape follows the launch size since round 3; empty slots are filtered below)
        raw = np.zeros(tiles * 8, np.uint64)
        rc = stamps_read(raw.ctypes.data_as(C.c_void_p), raw.size)
        assert rc == 0, rc
        s = raw.reshape(tiles, 8)
        s = s[s[:, 6] != 0].astype(np.float64)
        flops = 2.0 * M * K * co
        rt0, rt1 = s[:, 0], s[:, 6]
        span_us = (rt1.max() - rt0.min()) / 100.0
        clock = ((s[:, 5] - s[:, 1]) / ((rt1 - rt0) * 10.0)).mean()
        start = (rt0 - rt0.min()) / 100.0
        pro, fill, loop, epi = s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4]
        # matrix-pipe cycles per wave of a workgroup: its share of the launch's FLOPs at 64 FLOP / clk / SIMD, four waves (any tile, any
        # fp32 MFMA shape; a launch with more than 65536 workgroups is sampled by its first 65536)
        n_wg = max(len(s), 1)
        mfma_cyc = int(flops * min(1.0, 65536.0 / max(n_wg, 1)) / n_wg / 4 / 64) if n_wg < 65536 else 0
        if wino:   # per wave: 4 planes x (ic / 2) MFMAs of 64 cycles for 32 tiles x 32 channels
            mfma_cyc = 4 * (ci // 2) * 64
        hw = raw.reshape(tiles, 8)[:, 7]
        cu_key = ((hw >> np.uint64(32)) << np.uint64(16)) | ((hw >> np.uint64(8)) & np.uint64(0xff))
        _, per_cu = np.unique(cu_key[raw.reshape(tiles, 8)[:, 6] != 0], return_counts=True)
        med = lambda a: float(np.median(a))
        total_pipe = len(s) * mfma_cyc                        # pipe cycles per SIMD column (one wave of each wg per SIMD)
        if wino:   # persistent workgroups walk several tiles: count the executed MFMA work itself (64 flop / clk / SIMD)
            total_pipe = flops / 2.25 / 64.0 / 4.0
        if wino:
            flops_exec = flops / 2.25
            print("   (Winograd: executes %.1f GFLOP of MFMA work for %.1f GFLOP of direct convolution)" % (flops_exec / 1e9, flops / 1e9))
        span_cyc = span_us * 1e-6 * clock * 1e9
        print("%s  %dx%dx%d->%dx%dx%d k%ds%d  [%s]" % (sp, h, w, ci, oh, ow, co, k, st, name))
        print("   %.4f ms back-to-back = %.1f TF/s | stamped launch: span %.1f us, clock %.2f GHz, %d workgroups on %d CUs (per CU %d..%d)"
              % (ms, flops / ms / 1e9, span_us, clock, len(s), len(per_cu), per_cu.min(), per_cu.max()))
        print("   start of workgroups after the first: p50 %.1f us  p90 %.1f us  max %.1f us" % (med(start), float(np.percentile(start, 90)), start.max()))
        print("   cycles per workgroup (median): prologue %.0f  fill %.0f  loop %.0f (MFMA alone %d = %.0f %% of it)  epilogue %.0f  | total %.0f"
              % (med(pro), med(fill), med(loop), mfma_cyc, 100.0 * mfma_cyc / med(loop), med(epi), med(s[:, 5] - s[:, 1])))
        fr = start < 2.0   # the workgroups of the first round: nothing to overlap their prologue with
        if fr.any() and not wino:
            print("   first round (%d workgroups): prologue %.0f  fill %.0f cycles -> first MFMA %.1f us after the launch; the others: prologue %.0f"
                  % (int(fr.sum()), med(pro[fr]), med(fill[fr]), med((s[:, 3] - s[:, 1])[fr]) / (clock * 1e3), med(pro[~fr]) if (~fr).any() else 0.0))
        print("   MFMA pipe busy over the span: %.1f %%   (at 2.4 GHz and 100 %% the launch would take %.1f us)"
              % (100.0 * total_pipe / (256.0 * span_cyc), flops / 157.3e12 * 1e6))
        if args.timeline:
            live = raw.reshape(tiles, 8)[:, 6] != 0
            keys = cu_key[live]
            slot = (hw[live] & np.uint64(15)).astype(int)
            one = keys == np.unique(keys)[args.timeline - 1]
            t0 = s[:, 1][one].min()
            order = np.argsort(s[:, 1][one])
            print("   one CU, cycles since its first workgroup started (wave slot | start, loop start, loop end, end):")
            for i in order:
                r = s[one][i]
                print("      slot %d | %8.0f %8.0f %8.0f %8.0f" % (slot[one][i], r[1] - t0, r[3] - t0, r[4] - t0, r[5] - t0))
        for b in (dx, dw, db, dy):
            b.free()


if __name__ == "__main__":
    main()
