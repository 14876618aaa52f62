#!/usr/bin/env python3
"""tools/conv_bench.py -- times si_hip_conv2d_f32 on the distinct conv shapes of a synthesized model
(default: YOLOv5s 640x640, batch 32) with HIP events, back-to-back launches per shape.

    python tools/conv_bench.py [--batch 32] [--reps 20] [--model yolov5s] [--only 3x3]

Prints one line per distinct shape: count in the graph, kernel instantiation, ms, TFLOP/s, algorithmic GB/s,
and the share of the model's conv time.  Development tool; not part of the product or the tests.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from simpleinfer_amd import _native, hipops, modelgen as mg  # noqa: E402
from simpleinfer_amd._native import SiConv2dDesc  # noqa: E402


def conv_shapes(builder):
    shapes = {}
    for ln in builder.lines:
        t = ln.split()
        if t[0] != "nn.Conv2d":
            continue
        kv = dict(x.split("=", 1) for x in t[4 + int(t[2]) + int(t[3]):] if "=" in x)
        n, ci, ih, iw = builder.shapes[t[4]]
        _, co, oh, ow = builder.shapes[t[5]]
        k = tuple(int(v) for v in kv["kernel_size"].strip("()").split(","))
        s = tuple(int(v) for v in kv["stride"].strip("()").split(","))
        p = tuple(int(v) for v in kv["padding"].strip("()").split(","))
        key = (n, ih, iw, ci, oh, ow, co, k, s, p, int(kv["groups"]))
        shapes[key] = shapes.get(key, 0) + 1
    return shapes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--min-ms", type=float, default=0.0, help="sustained timing: repeat until this many ms of launches were timed (DVFS: a "
                    "few-ms burst after idle runs at 1.7-2.0 GHz, a sustained run at ~2.35 GHz), after an equally long warm-up")
    ap.add_argument("--model", default="yolov5s")
    ap.add_argument("--only", default="")
    ap.add_argument("--act", default="silu")
    ap.add_argument("--algo", default="direct", choices=["direct", "wino", "wino43"], help="wino: si_hip_conv2d_wino23_f32 on eligible shapes only")
    ap.add_argument("--shape", action="append", default=[], help="n,h,w,ci,co,k,s,p[,groups] (repeatable): custom shapes instead of a model")
    ap.add_argument("--graph", type=int, default=0, help="(--f16) time a replayed hipGraph of this many launches instead of host-issued launches")
    ap.add_argument("--f16-tile", type=int, default=-1, help="(--f16) SiConvPlan::f16_tile for every launch (-1: the policy; a forced tile also bypasses the slab / patch kernels)")
    ap.add_argument("--f16", action="store_true", help="the fp16 storage path (si_hip_conv2d_f16; SI_CONV_F16_VARIANT picks the tile); stems are skipped")
    args = ap.parse_args()
    H = _native.hip()
    b = mg.build_yolov5s(args.batch, args.size) if args.model == "yolov5s" else mg.build_resnet18(args.batch, 224)
    shapes = conv_shapes(b)
    if args.shape:
        shapes = {}
        for sp in args.shape:
            vals = [int(v) for v in sp.split(",")]
            n, h, w, ci, co, k, st, pd = vals[:8]
            grp = vals[8] if len(vals) > 8 else 1
            oh, ow = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
            shapes[(n, h, w, ci, oh, ow, co, (k, k), (st, st), (pd, pd), grp)] = 1
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0))
    H.si_hip_event_create(C.byref(ev1))
    rows = []
    for key, count in shapes.items():
        n, ih, iw, ci, oh, ow, co, k, s, p, g = key
        tag = "%dx%d" % k
        if args.only and args.only != tag:
            continue
        d = SiConv2dDesc(n, ih, iw, ci, ci, oh, ow, co, co, k[0], k[1], s[0], s[1], 1, 1, p[0], p[1], g, 1,
                         hipops.ACT[args.act], 0, co, 0, 0.0)
        if args.f16:
            if H.si_hip_conv2d_f16_supported(C.byref(d)) != 1:
                continue
            if args.f16_tile >= 0:
                _plan = _native.SiConvPlan(f16_tile=args.f16_tile)
                d.plan = C.pointer(_plan)
            wn = H.si_hip_conv2d_f16_weight_elems(C.byref(d))
            rng = np.random.default_rng(0)
            w32 = ((rng.random((co, ci // g, k[0], k[1]), dtype=np.float32) - 0.5) * 0.1)
            packed = np.zeros(wn, np.float16)
            assert H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), w32.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)) == 0
            dx = hipops.DeviceBuffer.from_numpy(rng.random((n, ih, iw, ci), dtype=np.float32).astype(np.float16))
            dw = hipops.DeviceBuffer.from_numpy(packed)
            db = hipops.DeviceBuffer.from_numpy(rng.random(co, dtype=np.float32))
            dy = hipops.DeviceBuffer(n * oh * ow * co * 2)

            def fn16(stream=None):
                return H.si_hip_conv2d_f16(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, 0, stream)
            for _ in range(2):
                assert fn16() == 0
            H.si_hip_device_sync()
            reps = args.reps
            ms = C.c_float()
            if args.graph:
                # a launch through ctypes costs the host 10-17 us: a kernel shorter than that is timed as a replayed hipGraph of
                # `--graph` launches (round 5: the slab kernels' 17.0 us "back to back" was the host)
                st, gx = C.c_void_p(), C.c_void_p()
                assert H.si_hip_stream_create(C.byref(st)) == 0
                assert H.si_hip_graph_begin_capture(st) == 0
                for _ in range(args.graph):
                    assert fn16(st) == 0
                assert H.si_hip_graph_end_capture(st, C.byref(gx)) == 0
                reps = 2
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
                ms = ms.value / (reps * args.graph)
                H.si_hip_graph_destroy(gx)
                H.si_hip_stream_destroy(st)
            else:
                while True:
                    H.si_hip_event_record(ev0, None)
                    for _ in range(reps):
                        fn16()
                    H.si_hip_event_record(ev1, None)
                    H.si_hip_event_sync(ev1)
                    H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
                    if ms.value >= args.min_ms:
                        break
                    reps = int(reps * max(2.0, 1.2 * args.min_ms / max(ms.value, 1e-3)))
                ms = ms.value / reps
            flops = 2.0 * n * oh * ow * co * k[0] * k[1] * (ci // g)
            byts = 2.0 * (n * ih * iw * ci + n * oh * ow * co + co * (ci // g) * k[0] * k[1])   # (wn counts both packed weight images)
            rows.append((key, count, "f16 v%s" % (args.f16_tile if args.f16_tile >= 0 else "policy"), ms, flops / ms / 1e9, byts / ms / 1e6, flops))
            for buf in (dx, dw, db, dy):
                buf.free()
            continue
        wino = args.algo != "direct"
        fam = {"wino": "wino23", "wino43": "wino43"}.get(args.algo, "")
        if wino and not getattr(H, "si_hip_conv2d_%s_eligible" % fam)(C.byref(d)):
            continue
        wn = getattr(H, "si_hip_conv2d_%s_weight_elems" % fam)(C.byref(d)) if wino else H.si_hip_conv2d_weight_elems(C.byref(d))
        fn = getattr(H, "si_hip_conv2d_%s_f32" % fam) if wino else H.si_hip_conv2d_f32
        rng = np.random.default_rng(0)
        dx = hipops.DeviceBuffer.from_numpy(rng.random((n, ih, iw, ci), dtype=np.float32))
        dw = hipops.DeviceBuffer.from_numpy((rng.random(wn, dtype=np.float32) - 0.5) * 0.1)
        db = hipops.DeviceBuffer.from_numpy(rng.random(co, dtype=np.float32))
        dy = hipops.DeviceBuffer(n * oh * ow * co * 4)
        for _ in range(2):
            rc = fn(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, None)
            assert rc == 0, rc
        H.si_hip_device_sync()
        reps = args.reps
        ms = C.c_float()
        while True:
            H.si_hip_event_record(ev0, None)
            for _ in range(reps):
                fn(C.byref(d), dx.ptr, dw.ptr, db.ptr, None, dy.ptr, None)
            H.si_hip_event_record(ev1, None)
            H.si_hip_event_sync(ev1)
            H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
            if ms.value >= args.min_ms:
                break
            reps = int(reps * max(2.0, 1.2 * args.min_ms / max(ms.value, 1e-3)))   # the short passes double as warm-up
        ms = ms.value / reps
        flops = 2.0 * n * oh * ow * co * k[0] * k[1] * (ci // g)
        byts = 4.0 * (n * ih * iw * ci + n * oh * ow * co + wn)
        name = fam if wino else H.si_hip_conv2d_kernel_name(C.byref(d), dx.ptr).decode().replace("conv_igemm_f32_kernel", "")
        rows.append((key, count, name, ms, flops / ms / 1e9, byts / ms / 1e6, flops))
        for buf in (dx, dw, db, dy):
            buf.free()
    total = sum(r[3] * r[1] for r in rows)
    print("%-34s %3s %-24s %8s %8s %9s %6s" % ("in(HxWxC)->out(HxWxC) k/s", "cnt", "kernel", "ms", "TF/s", "GB/s", "share"))
    for key, count, name, ms, tf, gb, fl in sorted(rows, key=lambda r: -r[3] * r[1]):
        n, ih, iw, ci, oh, ow, co, k, s, p, g = key
        print("%-34s %3d %-24s %8.4f %8.1f %9.1f %5.1f%%" % (
            "%dx%dx%d->%dx%dx%d k%ds%d" % (ih, iw, ci, oh, ow, co, k[0], s[0]), count, name, ms, tf, gb, 100 * ms * count / total))
    tot_fl = sum(r[6] * r[1] for r in rows)
    print("total conv time %.3f ms per forward (batch %d) = %.1f TF/s = %.0f img/s conv-only" % (
        total, args.batch, tot_fl / total / 1e9, args.batch / total * 1e3))


if __name__ == "__main__":
    main()
