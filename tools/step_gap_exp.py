#!/usr/bin/env python3
"""tools/step_gap_exp.py [--batch B] -- what the boundary between two synchronous Forward() calls costs at small batches.

YOLOv5s 640x640 fp32, device-resident input.  Modes, interleaved, median of several windows each:
  sync       e.forward() per step (what bench.py times: Engine::Forward is synchronous)
  async      e.forward_async() per step, ONE e.sync() per window (launches of step k+1 are queued behind step k)
  graph      the same two with SetOption("graph", 1)
The difference sync - async is what the host-side wait + relaunch costs per step.
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simpleinfer_amd as si  # noqa: E402
from simpleinfer_amd import hipops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--windows", type=int, default=7)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    mg = si.modelgen
    with tempfile.TemporaryDirectory() as td:
        b = mg.build_yolov5s(args.batch, 640)
        pp, bp = os.path.join(td, "m.param"), os.path.join(td, "m.bin")
        b.save(pp, bp)
        engines = {}
        extra = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.opt}
        for name, g in (("eager", 0), ("graph", 1)):
            e = si.Engine(device=0, outputs_to_host=0, graph=g, **extra)
            e.load_model(pp, bp)
            x = mg.synth_input((args.batch, 640, 640, 3), seed=1)
            dx = hipops.DeviceBuffer.from_numpy(x)
            e.input_device(e.input_names()[0], dx.ptr)
            for _ in range(20):
                e.forward()
            engines[name] = (e, dx)
        res = {}
        for w in range(args.windows):
            for name, (e, _) in engines.items():
                for mode in ("sync", "async"):
                    e.sync()
                    t0 = time.perf_counter()
                    if mode == "sync":
                        for _ in range(args.steps):
                            e.forward()
                    else:
                        for _ in range(args.steps):
                            e.forward_async()
                        e.sync()
                    dt = (time.perf_counter() - t0) / args.steps
                    res.setdefault((name, mode), []).append(dt * 1e6)
        for k, v in res.items():
            v.sort()
            print("batch %d %-6s %-6s median %.1f us/step  min %.1f  (%.0f img/s)" % (args.batch, k[0], k[1], v[len(v) // 2], v[0], args.batch / v[len(v) // 2] * 1e6))


if __name__ == "__main__":
    main()
