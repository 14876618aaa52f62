import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simpleinfer_amd import hipops
import torch
def ref_conv(x, w):
    return torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(w).double(), stride=2, padding=1).permute(0, 2, 3, 1).numpy()
rng = np.random.default_rng(0)
for (n, hw, ic, oc) in ((1, 8, 16, 16), (1, 16, 16, 64), (2, 16, 32, 64)):
    x = rng.standard_normal((n, hw, hw, ic)).astype(np.float32)
    w = (rng.standard_normal((oc, ic, 3, 3)) * 0.3).astype(np.float32)
    got = hipops.conv2d_s2poly(x, w)
    ref = ref_conv(x, w)
    err = np.abs(got - ref)
    print("case", n, hw, ic, oc, "max rel", err.max() / np.abs(ref).max())
    print(" err by (oy%2, ox%2):", [[float(err[:, a::2, b::2].max()) for b in range(2)] for a in range(2)])
    print(" err by channel group of 16:", [float(err[..., g * 16:(g + 1) * 16].max()) for g in range(oc // 16)])
    print(" err by tile row:", [float(err[:, 2 * t:2 * t + 2].max()) for t in range(hw // 4)])
    # single-tap filters: which taps are wrong?
    for ky in range(3):
        for kx in range(3):
            w1 = np.zeros_like(w); w1[:, :, ky, kx] = w[:, :, ky, kx]
            e1 = np.abs(hipops.conv2d_s2poly(x, w1) - ref_conv(x, w1)).max()
            print("  tap", ky, kx, "err %.3e" % e1, end=";")
    print()
    # single-channel input: which channels are wrong?
    bad = []
    for c in range(ic):
        x1 = np.zeros_like(x); x1[..., c] = x[..., c]
        e1 = np.abs(hipops.conv2d_s2poly(x1, w) - ref_conv(x1, w)).max()
        if e1 > 1e-3: bad.append(c)
    print(" bad input channels:", bad)
