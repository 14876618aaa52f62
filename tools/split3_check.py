#!/usr/bin/env python3
"""tools/split3_check.py -- the fp32-by-three-fp16-products kernel (conv_split3.hip) against the true-fp32 implicit GEMM: (1) error of
both against the float64 convolution on the same data (max |err| / max |ref|, small shapes), (2) time per launch on YOLOv5s' two
largest stride-2 layers (replayed hipGraph of 30 launches, sustained, interleaved).  VERDICT r04 item 4's kill criteria: error <= 2x
the fp32 kernel's, >= 1.8x faster standalone."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleinfer_amd import _native, hipops  # noqa: E402
from simpleinfer_amd._native import SiConv2dDesc  # noqa: E402


def ref64(x, w, b, s, p):
    n, ih, iw, ic = x.shape
    oc, _, kh, kw = w.shape
    xp = np.zeros((n, ih + 2 * p, iw + 2 * p, ic), np.float64)
    xp[:, p:p + ih, p:p + iw] = x
    oh, ow = (ih + 2 * p - kh) // s + 1, (iw + 2 * p - kw) // s + 1
    out = np.zeros((n, oh, ow, oc), np.float64)
    w64 = w.astype(np.float64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + s * oh:s, kx:kx + s * ow:s]
            out += np.einsum("nhwc,oc->nhwo", patch, w64[:, :, ky, kx])
    return out + b.astype(np.float64)


def main():
    H = _native.hip()
    print("== error against the float64 convolution (max |err| / max |ref|)")
    for (n, hw, ic, oc, k, s, scale) in [(2, 20, 128, 256, 3, 2, 1.0), (1, 16, 256, 128, 3, 1, 1.0), (2, 12, 512, 128, 1, 1, 1.0), (2, 20, 128, 256, 3, 2, 50.0),
                                       (2, 24, 32, 64, 3, 2, 1.0), (1, 20, 64, 64, 3, 1, 1.0), (2, 12, 96, 160, 1, 1, 1.0), (1, 20, 128, 48, 3, 1, 1.0)]:
        rng = np.random.default_rng(5)
        x = (rng.standard_normal((n, hw, hw, ic)) * scale).astype(np.float32)
        w = (rng.standard_normal((oc, ic, k, k)) * 0.05).astype(np.float32)
        b = rng.standard_normal(oc).astype(np.float32)
        ref = ref64(x, w, b, s, k // 2)
        e32 = np.abs(hipops.conv2d(x, w, b, (s, s), (k // 2, k // 2)).astype(np.float64) - ref).max() / np.abs(ref).max()
        e3 = np.abs(hipops.conv2d_split3(x, w, b, (s, s), (k // 2, k // 2)).astype(np.float64) - ref).max() / np.abs(ref).max()
        e16 = np.abs(hipops.conv2d_f16(x.astype(np.float16), w.astype(np.float16), b, (s, s), (k // 2, k // 2), out_f32=True).astype(np.float64) - ref).max() / np.abs(ref).max()
        print("  %dx%dx%d -> %d k%d s%d, |x| ~ %.0f:  fp32 kernel %.2e   split3 %.2e (%.2fx)   [plain fp16 storage %.1e]" % (hw, hw, ic, oc, k, s, scale, e32, e3, e3 / e32, e16))
    print("== time per launch, batch 32 (graph of 30 launches, sustained)")
    ev0, ev1, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1)); H.si_hip_stream_create(C.byref(st))
    for (n, hw, ic, oc, k, s) in [(32, 80, 128, 256, 3, 2), (32, 40, 256, 512, 3, 2), (32, 160, 64, 128, 3, 2), (32, 20, 512, 512, 1, 1), (32, 40, 256, 256, 1, 1),
                               (32, 320, 32, 64, 3, 2), (32, 80, 64, 64, 3, 1), (32, 160, 64, 64, 1, 1), (32, 80, 128, 64, 1, 1)]:
        p = k // 2
        oh = (hw + 2 * p - k) // s + 1
        d = SiConv2dDesc(n, hw, hw, ic, ic, oh, oh, oc, oc, k, k, s, s, 1, 1, p, p, 1, 1, hipops.ACT["silu"], 0, oc, 0, 0.0)
        rng = np.random.default_rng(0)
        w32 = (rng.standard_normal((oc, ic, k, k)) * 0.05).astype(np.float32)
        pk32 = np.zeros(H.si_hip_conv2d_weight_elems(C.byref(d)), np.float32)
        assert H.si_hip_conv2d_pack_weight_host(C.byref(d), w32.ctypes.data_as(C.c_void_p), pk32.ctypes.data_as(C.c_void_p)) == 0
        pk3 = np.zeros(H.si_hip_conv2d_split3_weight_elems(C.byref(d)), np.float16)
        assert H.si_hip_conv2d_split3_pack_weight_host(C.byref(d), w32.ctypes.data_as(C.c_void_p), pk3.ctypes.data_as(C.c_void_p)) == 0
        dx = hipops.DeviceBuffer.from_numpy(rng.standard_normal((n, hw, hw, ic)).astype(np.float32))
        d32, d3 = hipops.DeviceBuffer.from_numpy(pk32), hipops.DeviceBuffer.from_numpy(pk3)
        db = hipops.DeviceBuffer.from_numpy(rng.standard_normal(oc).astype(np.float32))
        dy = hipops.DeviceBuffer(n * oh * oh * oc * 4)

        def timed(fn):
            gx = C.c_void_p()
            assert H.si_hip_graph_begin_capture(st) == 0
            for _ in range(30):
                assert fn() == 0
            assert H.si_hip_graph_end_capture(st, C.byref(gx)) == 0
            reps, ms = 2, C.c_float()
            while True:
                H.si_hip_event_record(ev0, st)
                for _ in range(reps):
                    H.si_hip_graph_launch(gx, st)
                H.si_hip_event_record(ev1, st)
                H.si_hip_event_sync(ev1)
                H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
                if ms.value >= 300:
                    break
                reps *= 2
            H.si_hip_graph_destroy(gx)
            return ms.value / (reps * 30) * 1e3
        flops = 2.0 * n * oh * oh * oc * k * k * ic
        for r in range(2):
            t32 = timed(lambda: H.si_hip_conv2d_f32(C.byref(d), dx.ptr, d32.ptr, db.ptr, None, dy.ptr, st))
            t3 = timed(lambda: H.si_hip_conv2d_split3_f32(C.byref(d), dx.ptr, d3.ptr, db.ptr, None, dy.ptr, st))
            print("  %dx%dx%d -> %d k%d s%d: fp32 kernel %.1f us (%.0f TF/s)   split3 %.1f us (%.0f TF/s-equivalent)   %.2fx" % (
                hw, hw, ic, oc, k, s, t32, flops / t32 / 1e6, t3, flops / t3 / 1e6, t32 / t3))
        for buf in (dx, d32, d3, db, dy):
            buf.free()


def detect():
    """Detect levels of YOLOv5s batch 32: si_hip_conv2d_yolo_f32 vs si_hip_conv2d_split3_yolo_f32, ms per launch (graph of 30)."""
    from simpleinfer_amd._native import SiYoloLevel
    H = _native.hip()
    n, na, ne = 32, 3, 85
    rows_total = (80 * 80 + 40 * 40 + 20 * 20) * na
    dout = hipops.DeviceBuffer(n * rows_total * ne * 4)
    ev0, ev1, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1)); H.si_hip_stream_create(C.byref(st))
    off = 0
    for hw, c in ((80, 128), (40, 256), (20, 512)):
        rng = np.random.default_rng(0)
        d = SiConv2dDesc(n, hw, hw, c, c, hw, hw, na * ne, na * ne, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, na * ne, 0, 0.0)
        w = ((rng.random((na * ne, c, 1, 1), dtype=np.float32) - 0.5) * 0.1)
        pk32 = np.zeros(H.si_hip_conv2d_weight_elems(C.byref(d)), np.float32)
        assert H.si_hip_conv2d_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), pk32.ctypes.data_as(C.c_void_p)) == 0
        pk3 = np.zeros(H.si_hip_conv2d_split3_weight_elems(C.byref(d)), np.float16)
        assert H.si_hip_conv2d_split3_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), pk3.ctypes.data_as(C.c_void_p)) == 0
        dx = hipops.DeviceBuffer.from_numpy((rng.random((n, hw, hw, c), dtype=np.float32) - 0.5))
        d32, d3 = hipops.DeviceBuffer.from_numpy(pk32), hipops.DeviceBuffer.from_numpy(pk3)
        db = hipops.DeviceBuffer.from_numpy(rng.random(na * ne, dtype=np.float32))
        dg = hipops.DeviceBuffer.from_numpy(rng.random((hw * hw * na, 2), dtype=np.float32))
        da = hipops.DeviceBuffer.from_numpy(rng.random((hw * hw * na, 2), dtype=np.float32))
        lv = SiYoloLevel(na, ne, rows_total, off, 8.0)

        def timed(fn):
            gx = C.c_void_p()
            assert H.si_hip_graph_begin_capture(st) == 0
            for _ in range(30):
                assert fn() == 0
            assert H.si_hip_graph_end_capture(st, C.byref(gx)) == 0
            reps, ms = 2, C.c_float()
            while True:
                H.si_hip_event_record(ev0, st)
                for _ in range(reps):
                    H.si_hip_graph_launch(gx, st)
                H.si_hip_event_record(ev1, st)
                H.si_hip_event_sync(ev1)
                H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
                if ms.value >= 300:
                    break
                reps *= 2
            H.si_hip_graph_destroy(gx)
            return ms.value / (reps * 30) * 1e3
        for r in range(2):
            t32 = timed(lambda: H.si_hip_conv2d_yolo_f32(C.byref(d), dx.ptr, d32.ptr, db.ptr, C.byref(lv), dg.ptr, da.ptr, dout.ptr, st))
            t3 = timed(lambda: H.si_hip_conv2d_split3_yolo_f32(C.byref(d), dx.ptr, d3.ptr, db.ptr, C.byref(lv), dg.ptr, da.ptr, dout.ptr, st))
            print("  Detect level %dx%dx%d: fp32 kernel %.1f us   split3 %.1f us   %.2fx" % (hw, hw, c, t32, t3, t32 / t3))
        off += hw * hw * na


def wino():
    """3x3 stride-1 layers of YOLOv5s (batch 32) and ResNet18 (batch 64): the fp32 Winograd kernel vs its split form, ms per launch."""
    H = _native.hip()
    ev0, ev1, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
    H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1)); H.si_hip_stream_create(C.byref(st))
    for (n, hw, ic, oc) in [(32, 160, 32, 32), (32, 80, 64, 64), (32, 40, 128, 128), (32, 20, 256, 256), (64, 56, 64, 64), (64, 28, 128, 128), (64, 14, 256, 256), (64, 7, 512, 512)]:
        d = SiConv2dDesc(n, hw, hw, ic, ic, hw, hw, oc, oc, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, hipops.ACT["silu"], 0, oc, 0, 0.0)
        rng = np.random.default_rng(0)
        w32 = (rng.standard_normal((oc, ic, 3, 3)) * 0.05).astype(np.float32)
        u32 = np.zeros(H.si_hip_conv2d_wino23_weight_elems(C.byref(d)), np.float32)
        assert H.si_hip_conv2d_wino23_pack_weight_host(C.byref(d), w32.ctypes.data_as(C.c_void_p), u32.ctypes.data_as(C.c_void_p)) == 0
        u3 = np.zeros(H.si_hip_conv2d_wino23_split_weight_elems(C.byref(d)), np.float16)
        assert H.si_hip_conv2d_wino23_split_pack_weight_host(C.byref(d), w32.ctypes.data_as(C.c_void_p), u3.ctypes.data_as(C.c_void_p)) == 0
        dx = hipops.DeviceBuffer.from_numpy(rng.standard_normal((n, hw, hw, ic)).astype(np.float32))
        d32, d3 = hipops.DeviceBuffer.from_numpy(u32), hipops.DeviceBuffer.from_numpy(u3)
        db = hipops.DeviceBuffer.from_numpy(rng.standard_normal(oc).astype(np.float32))
        dy = hipops.DeviceBuffer(n * hw * hw * oc * 4)

        def timed(fn):
            gx = C.c_void_p()
            assert H.si_hip_graph_begin_capture(st) == 0
            for _ in range(30):
                assert fn() == 0
            assert H.si_hip_graph_end_capture(st, C.byref(gx)) == 0
            reps, ms = 2, C.c_float()
            while True:
                H.si_hip_event_record(ev0, st)
                for _ in range(reps):
                    H.si_hip_graph_launch(gx, st)
                H.si_hip_event_record(ev1, st)
                H.si_hip_event_sync(ev1)
                H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
                if ms.value >= 300:
                    break
                reps *= 2
            H.si_hip_graph_destroy(gx)
            return ms.value / (reps * 30) * 1e3
        flops = 2.0 * n * hw * hw * oc * 9 * ic
        t32 = timed(lambda: H.si_hip_conv2d_wino23_f32(C.byref(d), dx.ptr, d32.ptr, db.ptr, None, dy.ptr, st))
        t3 = timed(lambda: H.si_hip_conv2d_wino23_split_f32(C.byref(d), dx.ptr, d3.ptr, db.ptr, None, dy.ptr, st))
        print("  batch %d %dx%dx%d -> %d: fp32 Winograd %.1f us (%.0f TF/s direct-equivalent)   split %.1f us (%.0f)   %.2fx" % (
            n, hw, hw, ic, oc, t32, flops / t32 / 1e6, t3, flops / t3 / 1e6, t32 / t3))


if __name__ == "__main__":
    if "--wino" in sys.argv:
        wino()
    elif "--detect" in sys.argv:
        detect()
    else:
        main()
