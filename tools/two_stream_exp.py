#!/usr/bin/env python3
"""tools/two_stream_exp.py -- does running the batch as S concurrent sub-batches (one engine + HIP stream each) beat one
batch-B engine?  Kernel tails / launch ramps of one stream overlap the other's work.  Development experiment."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simpleinfer_amd as si  # noqa: E402
from simpleinfer_amd import hipops, _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--splits", default="1,2,4")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--fp16", type=int, default=0)
    ap.add_argument("--graph", type=int, default=0)
    args = ap.parse_args()
    H = _native.hip()
    mg = si.modelgen
    with tempfile.TemporaryDirectory() as td:
        pp, bp = os.path.join(td, "m.param"), os.path.join(td, "m.bin")
        mg.build_yolov5s(1, 640).save(pp, bp)
        for S in [int(v) for v in args.splits.split(",")]:
            per = args.batch // S
            engines, bufs = [], []
            for k in range(S):
                e = si.Engine(batch=per, outputs_to_host=0, fp16=args.fp16, graph=args.graph)
                e.load_model(pp, bp)
                dx = hipops.DeviceBuffer.from_numpy(mg.synth_input((per, 640, 640, 3), seed=1 + k))
                e.input_device(e.input_names()[0], dx.ptr)
                e.forward()
                engines.append(e)
                bufs.append(dx)
            def step():
                for e in engines:
                    e.forward_async()
                for e in engines:
                    e.sync()
            for _ in range(10):
                step()
            H.si_hip_device_sync()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < args.seconds:
                step()
                n += 1
            H.si_hip_device_sync()
            dt = time.perf_counter() - t0
            print("sub-batches %d x %d: %.3f ms per %d images = %.0f img/s" % (S, per, dt / n * 1e3, args.batch, args.batch * n / dt), flush=True)
            for e in engines:
                e.release()


if __name__ == "__main__":
    main()
