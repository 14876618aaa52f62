"""Generates tests/golden/*.npz -- seeded inputs and expected outputs for the operator set.

Run in the build container only:  python tests/golden/make_golden.py
Expected outputs come from torch-CPU in float64 (PyTorch ops are the semantics pnnx / the reference
mirror: nn.Conv2d, nn.MaxPool2d, nn.Upsample(nearest), torch.cat, ...), converted NCHW<->NHWC; the
pnnx loader fixtures come from the REFERENCE's own loader (oracle/_ref/ref_pnnx_dump, compiled from
/root/reference/src/pnnx).  Nothing of the reference's source text is stored: fixtures are data.
"""
import os
import subprocess
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from simpleinfer_amd import modelgen as mg  # noqa: E402


def u(seed, shape, lo=0.0, hi=1.0):
    r = np.random.Generator(np.random.Philox(seed))
    return (lo + (hi - lo) * r.random(shape, dtype=np.float32)).astype(np.float32)


def nchw(x):
    return torch.from_numpy(x).permute(0, 3, 1, 2).double()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().float().numpy()


def main():
    out = {}
    # ---- conv2d cases: (n,h,w,ci,co,k,s,p,d,g,bias)
    conv_cases = [
        ("conv_3x3_s1_p1", (1, 16, 16, 8, 12, 3, 1, 1, 1, 1, True)),      # Winograd-eligible in the reference
        ("conv_3x3_s1_p0", (2, 9, 11, 5, 7, 3, 1, 0, 1, 1, True)),        # odd sizes, ic%4 != 0, oc%4 != 0
        ("conv_3x3_s2_p1", (2, 15, 17, 16, 24, 3, 2, 1, 1, 1, True)),
        ("conv_6x6_s2_p2", (1, 32, 32, 3, 8, 6, 2, 2, 1, 1, True)),       # YOLOv5 stem shape family
        ("conv_1x1", (2, 10, 10, 32, 255, 1, 1, 0, 1, 1, True)),          # Detect 1x1: oc = 255
        ("conv_grouped", (1, 12, 12, 8, 16, 3, 1, 1, 1, 2, True)),
        ("conv_depthwise", (1, 12, 12, 8, 8, 3, 1, 1, 1, 8, False)),
        ("conv_dilated", (1, 14, 14, 4, 6, 3, 1, 2, 2, 1, True)),
        ("conv_7x7_s2_p3", (1, 20, 20, 3, 16, 7, 2, 3, 1, 1, True)),      # ResNet stem family
    ]
    for i, (name, (n, h, w, ci, co, k, s, p, d, g, has_b)) in enumerate(conv_cases):
        x = u(100 + i, (n, h, w, ci))
        wt = u(200 + i, (co, ci // g, k, k), -0.5, 0.5)
        b = u(300 + i, (co,), -0.5, 0.5) if has_b else None
        y = F.conv2d(nchw(x), torch.from_numpy(wt).double(), None if b is None else torch.from_numpy(b).double(),
                     stride=s, padding=p, dilation=d, groups=g)
        out[name + "/x"], out[name + "/w"], out[name + "/y"] = x, wt, nhwc(y)
        out[name + "/cfg"] = np.array([s, p, d, g], np.int32)
        if b is not None:
            out[name + "/b"] = b
    # ---- pooling / resampling / elementwise
    x = u(1, (2, 20, 20, 16), -1, 1)
    out["maxpool_k5s1p2/x"] = x
    out["maxpool_k5s1p2/y"] = nhwc(F.max_pool2d(nchw(x), 5, 1, 2))
    x = u(2, (1, 15, 15, 5), -1, 1)
    out["maxpool_k3s2p1/x"] = x
    out["maxpool_k3s2p1/y"] = nhwc(F.max_pool2d(nchw(x), 3, 2, 1))
    x = u(3, (2, 7, 7, 12), -1, 1)
    out["gap/x"] = x
    out["gap/y"] = nhwc(F.adaptive_avg_pool2d(nchw(x), 1))
    x = u(4, (2, 10, 10, 8), -1, 1)
    out["upsample2/x"] = x
    out["upsample2/y"] = nhwc(F.interpolate(nchw(x), scale_factor=2.0, mode="nearest"))
    xs = [u(5, (1, 6, 6, 3)), u(6, (1, 6, 6, 2)), u(7, (1, 6, 6, 4))]
    out["cat/x0"], out["cat/x1"], out["cat/x2"] = xs
    out["cat/y"] = nhwc(torch.cat([nchw(v) for v in xs], 1))
    x = u(8, (1, 9, 9, 6), -6, 6)
    out["act/x"] = x
    t = torch.from_numpy(x).double()
    out["act/silu"] = F.silu(t).float().numpy()
    out["act/relu"] = F.relu(t).float().numpy()
    out["act/sigmoid"] = torch.sigmoid(t).float().numpy()
    out["act/hardsigmoid"] = F.hardsigmoid(t).float().numpy()
    out["act/hardswish"] = F.hardswish(t).float().numpy()
    a, b = u(9, (2, 5, 5, 8), -1, 1), u(10, (2, 1, 1, 8), -1, 1)
    out["binary/a"], out["binary/b"] = a, b
    out["binary/add"], out["binary/mul"] = a + b, a * b
    x = u(11, (2, 6, 6, 5), -1, 1)
    m, v, g_, be = u(12, (5,), -0.5, 0.5), u(13, (5,), 0.5, 1.5), u(14, (5,), 0.5, 1.5), u(15, (5,), -0.5, 0.5)
    out["bn/x"], out["bn/mean"], out["bn/var"], out["bn/gamma"], out["bn/beta"] = x, m, v, g_, be
    out["bn/y"] = nhwc(F.batch_norm(nchw(x), torch.from_numpy(m).double(), torch.from_numpy(v).double(),
                                    torch.from_numpy(g_).double(), torch.from_numpy(be).double(), False, 0.0, 1e-5))
    x = u(16, (2, 3, 2, 4), -1, 1)
    out["flatten/x"] = x
    out["flatten/y"] = torch.flatten(nchw(x), 1).float().numpy()
    x, w, b = u(17, (3, 40), -1, 1), u(18, (10, 40), -0.3, 0.3), u(19, (10,), -0.1, 0.1)
    out["linear/x"], out["linear/w"], out["linear/b"] = x, w, b
    out["linear/y"] = F.linear(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                               torch.from_numpy(b).double()).float().numpy()
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **out)
    print("ops_golden.npz:", len(out), "arrays")

    # ---- pnnx loader fixtures from the reference's own loader
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_pnnx_dump")
    for name, b in (("toy_yolo", mg.build_toy_yolo(2, 64)), ("toy_classifier", mg.build_toy_classifier(2, 32))):
        pp, bp = "/tmp/%s.pnnx.param" % name, "/tmp/%s.pnnx.bin" % name
        b.save(pp, bp)
        for tag, extra in (("raw", []), ("expanded", ["--expand"])):
            txt = subprocess.run([ref, pp, bp] + extra, check=True, capture_output=True, text=True).stdout
            with open(os.path.join(HERE, "%s.%s.refdump.txt" % (name, tag)), "w") as f:
                f.write(txt)
    print("reference loader dumps written")


if __name__ == "__main__":
    main()
