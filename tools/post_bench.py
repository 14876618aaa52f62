"""tools/post_bench.py -- time si_hip_yolo_postprocess_f32 on (a) realistic clustered predictions and (b) the worst
case where every row survives the confidence filter.  Run on the GPU box; under rocprofv3 --kernel-trace --stats it
gives the per-kernel split."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from simpleinfer_amd import _native, hipops  # noqa: E402
from util import synthetic_predictions  # noqa: E402


def run(pred, thr, reps=5):
    H = _native.hip()
    n, rows, ne = pred.shape
    d = hipops.DeviceBuffer.from_numpy(pred)
    wsb = H.si_hip_yolo_postprocess_workspace_bytes(n, rows, ne)
    ws, dets, cnt = hipops.DeviceBuffer(wsb), hipops.DeviceBuffer(n * 300 * 24), hipops.DeviceBuffer(n * 4)
    ts = []
    for _ in range(reps + 1):
        H.si_hip_device_sync()
        t0 = time.perf_counter()
        rc = H.si_hip_yolo_postprocess_f32(d.ptr, n, rows, ne, thr, 0.45, 0, None, dets.ptr, cnt.ptr, 300, ws.ptr, wsb, None)
        H.si_hip_device_sync()
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    return float(np.mean(ts[1:])), cnt.to_numpy((n,), np.int32)


if __name__ == "__main__":
    n = 32
    pred = synthetic_predictions(3, n, 25200, nc=80, n_gt=8, hot_frac=0.03)
    ms, c = run(pred, 0.25)
    print("clustered (%.0f candidates/img -> %.1f boxes/img): %.3f ms per batch of %d" % (
        float((pred[..., 4] >= 0.5).sum()) / n, c.mean(), ms, n))
    r = np.random.Generator(np.random.Philox(1))
    pred = r.random((n, 25200, 85), dtype=np.float32)
    pred[..., 0:2] *= 640
    pred[..., 2:4] *= 100
    ms, c = run(pred, -1.0)
    print("every row survives (25200 candidates/img -> %.1f boxes/img): %.3f ms per batch of %d" % (c.mean(), ms, n))
