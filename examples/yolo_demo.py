"""examples/yolo_demo.py -- the reference's demo (test/test_yolo/test_yolo.cpp) end to end on the device:
letterbox packing -> Engine::Forward -> confidence filter / NMS / un-letterbox, with synthetic "photos" (no image codec
or model files in this environment: the network has seeded random weights, so the boxes mean nothing -- the point is the
data flow and that only a few KB per image ever leave HBM).

    python examples/yolo_demo.py [--size 640] [--images 4]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simpleinfer_amd as si  # noqa: E402
from simpleinfer_amd import hipops  # noqa: E402


def run(size=640, images=((480, 640), (1080, 810), (375, 500), (720, 1280)), seed=0):
    mg = si.modelgen
    n = len(images)
    rng = np.random.Generator(np.random.Philox(seed))
    with tempfile.TemporaryDirectory() as td:
        pp, bp = os.path.join(td, "m.pnnx.param"), os.path.join(td, "m.pnnx.bin")
        mg.build_yolov5s(n, size).save(pp, bp)
        e = si.Engine(outputs_to_host=0)
        e.load_model(pp, bp)
        x = np.empty((n, size, size, 3), np.float32)
        adjust = np.empty((n, 5), np.float32)
        for b, (h, w) in enumerate(images):
            hr, wr, scale, pt, pl = hipops.letterbox_geometry(h, w, size, size)
            resized = rng.integers(0, 256, (hr, wr, 3), dtype=np.uint8)   # stands in for cv::resize(imread(...))
            x[b] = hipops.letterbox(resized, size, size, pt, pl)           # pad / BGR->RGB / float / /255 on the device
            adjust[b] = (pl, pt, scale, w, h)
        e.input("0", x)
        e.forward()
        oname = e.output_names()[0]
        pred = e.extract(oname)                                            # (the demo keeps it on the device; see below)
        dets, counts = hipops.yolo_postprocess(pred, 0.25, 0.45, adjust=adjust, max_det=300)
    return x, adjust, pred, dets, counts


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--images", type=int, default=4)
    a = ap.parse_args()
    shapes = ((480, 640), (1080, 810), (375, 500), (720, 1280))[:a.images]
    _, _, pred, dets, counts = run(a.size, shapes)
    for b, d in enumerate(dets):
        print("image %d (%dx%d): %d boxes kept of %d predictions; first: %s" % (
            b, shapes[b][1], shapes[b][0], counts[b], pred.shape[1], np.array2string(d[0], precision=2) if len(d) else "-"))
