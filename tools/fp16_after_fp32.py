#!/usr/bin/env python3
"""tools/fp16_after_fp32.py -- why does the fp16 entry of bench.py's `secondary` leg read 2-3 % below a `--fp16 1` run on the same box?
Times the fp16 engine (a) alone in a fresh process state, (b) with a loaded fp32 engine of the same net alive beside it, (c) after
that engine was released.  usage: python tools/fp16_after_fp32.py [order]  (order: a,b,c letters, default 'abc')"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simpleinfer_amd as si  # noqa: E402
from simpleinfer_amd import _native, hipops  # noqa: E402

H = _native.hip()
mg = si.modelgen


def timed(e, secs=2.0, steps=40):
    for _ in range(20):
        e.forward()
    H.si_hip_device_sync()
    ws, tot = [], 0.0
    while tot < secs:
        t0 = time.perf_counter()
        for _ in range(steps):
            e.forward()
        H.si_hip_device_sync()
        ws.append(time.perf_counter() - t0)
        tot += ws[-1]
    ws.sort()
    return 32 / (ws[len(ws) // 2] / steps)


def make(pp, bp, x, **kw):
    e = si.Engine(device=0, outputs_to_host=0, **kw)
    e.load_model(pp, bp)
    e.input_device(e.input_names()[0], x.ptr)
    return e


with tempfile.TemporaryDirectory() as td:
    b = mg.build_yolov5s(32, 640)
    pp, bp = os.path.join(td, "m.param"), os.path.join(td, "m.bin")
    b.save(pp, bp)
    dx = hipops.DeviceBuffer.from_numpy(mg.synth_input((32, 640, 640, 3), seed=1))
    e16 = make(pp, bp, dx, fp16=1)
    print("fp16 alone            %.0f img/s" % timed(e16))
    e32 = make(pp, bp, dx)
    print("fp32 (beside)         %.0f img/s" % timed(e32))
    print("fp16, fp32 alive      %.0f img/s" % timed(e16))
    e16b = make(pp, bp, dx, fp16=1)
    print("fp16 created second   %.0f img/s" % timed(e16b))
    e32.release()
    print("fp16 (2nd), fp32 gone %.0f img/s" % timed(e16b))
    print("fp16 (1st) again      %.0f img/s" % timed(e16))
    e32 = make(pp, bp, dx)
    timed(e32, 1.0)
    e32.profile()
    print("fp16 after fp32.profile() %.0f img/s" % timed(e16))
    e16c = make(pp, bp, dx, fp16=1, graph=0, winograd=1)
    print("fp16 (3rd, created after profile, explicit graph=0 winograd=1) %.0f img/s" % timed(e16c))
    e16c.profile()
    print("fp16 (3rd) after its own profile() %.0f img/s" % timed(e16c))
    print("fp16 (1st) again      %.0f img/s" % timed(e16))
