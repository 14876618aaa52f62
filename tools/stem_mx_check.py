#!/usr/bin/env python3
"""tools/stem_mx_check.py <out.npy> -- the YOLOv5s stem (6x6 s2 p2, 3 -> 32, SiLU) on seeded inputs at three sizes (one with a ragged
last column tile), outputs concatenated into one file: run once per build / SI_STEM_MX setting and compare the files byte for byte."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpleinfer_amd import hipops  # noqa: E402

outs = []
for i, (n, h, w) in enumerate(((3, 640, 640), (2, 96, 652), (1, 64, 1300))):
    rng = np.random.default_rng(10 + i)
    x = rng.uniform(-1, 1, (n, h, w, 3)).astype(np.float32)
    wt = rng.uniform(-0.3, 0.3, (32, 3, 6, 6)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (32,)).astype(np.float32)
    y = hipops.conv2d(x, wt, b, (2, 2), (2, 2), act1="silu")
    print(y.shape, float(np.abs(y).max()), hipops.conv2d_kernel_name(x.shape, wt.shape, (2, 2), (2, 2)))
    outs.append(y.ravel())
np.save(sys.argv[1], np.concatenate(outs))
