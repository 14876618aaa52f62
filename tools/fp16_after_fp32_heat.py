#!/usr/bin/env python3
"""tools/fp16_after_fp32_heat.py -- does ten seconds of fp32 MFMA load in front of it lower the fp16 figure (power / thermal state)?
fp16 1 s -> fp32 10 s -> fp16 1 s at once -> idle 5 s -> fp16 1 s; twice."""
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


def timed(e, secs, steps=40):
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


with tempfile.TemporaryDirectory() as td:
    b = mg.build_yolov5s(32, 640)
    pp, bp = os.path.join(td, "m.param"), os.path.join(td, "m.bin")
    b.save(pp, bp)
    dx = hipops.DeviceBuffer.from_numpy(mg.synth_input((32, 640, 640, 3), seed=1))
    e16 = si.Engine(device=0, outputs_to_host=0, fp16=1); e16.load_model(pp, bp); e16.input_device(e16.input_names()[0], dx.ptr)
    e32 = si.Engine(device=0, outputs_to_host=0); e32.load_model(pp, bp); e32.input_device(e32.input_names()[0], dx.ptr)
    timed(e16, 0.5); timed(e32, 0.5)
    for r in range(2):
        print("fp16 (1 s)                      %.0f" % timed(e16, 1.0))
        print("fp32 (10 s)                     %.0f" % timed(e32, 10.0, 20))
        print("fp16 right behind it (1 s)      %.0f" % timed(e16, 1.0))
        print("fp16 second second              %.0f" % timed(e16, 1.0))
        time.sleep(5)
        print("fp16 after 5 s idle (1 s)       %.0f" % timed(e16, 1.0))
        print("fp16 sustained (5 s)            %.0f" % timed(e16, 5.0))
