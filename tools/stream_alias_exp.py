#!/usr/bin/env python3
"""tools/stream_alias_exp.py -- does the number of HIP streams alive in the process change an engine's speed?  (R5.17's hypothesis: HIP
multiplexes streams onto a few hardware queues; an engine created behind others may find its Detect side stream on the queue of its main
stream and lose the overlap.)  For N = 0, 1, 2, 3, 4, 6, 8 idle streams created FIRST: a fresh fp16 engine, 1.5 s of 40-step windows;
then the same with detect_stream=0 (Detect on the main stream: nothing to lose)."""
import ctypes as C
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


def timed(e, secs=1.5, steps=40):
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


with tempfile.TemporaryDirectory() as td:
    b = mg.build_yolov5s(32, 640)
    pp, bp = os.path.join(td, "m.param"), os.path.join(td, "m.bin")
    b.save(pp, bp)
    dx = hipops.DeviceBuffer.from_numpy(mg.synth_input((32, 640, 640, 3), seed=1))
    streams = []
    for n in (0, 1, 2, 3, 4, 6, 8):
        while len(streams) < n:
            st = C.c_void_p()
            assert H.si_hip_stream_create(C.byref(st)) == 0
            streams.append(st)
        res = []
        for ds in (1, 0):
            e = si.Engine(device=0, outputs_to_host=0, fp16=1, detect_stream=ds)
            e.load_model(pp, bp)
            e.input_device(e.input_names()[0], dx.ptr)
            res.append(timed(e))
            e.release()
        print("%d idle streams alive: fp16 %.0f img/s   with detect_stream=0 %.0f" % (n, res[0], res[1]))
