"""pnnx model synthesizer (tooling, host-only, numpy).

The reference ships no model files (its ``3rdparty/tmp`` submodule is absent),
and batch size is baked into a ``.pnnx.param`` (reference
``src/pnnx/ir.cpp:597-651``: every operand shape comes from ``#name=(...)f32``),
so benchmarks and parity tests need to *write* models.  This module emits the
two files the reference's loader reads:

* ``*.pnnx.param`` -- text: magic ``7767517``, ``<ops> <operands>``, one line
  per operator (``src/pnnx/ir.cpp:709-815``; value syntax ``:479-550``).
* ``*.pnnx.bin``  -- ZIP, stored-only, entries ``<opname>.<attr>`` holding raw
  little-endian tensors (``src/pnnx/storezip.cpp:117-229``).

Graphs follow SURVEY.md Appendix A (YOLOv5s v6) / A2 (torchvision ResNet18, BN
folded).  Weights come from a portable counter-based generator (splitmix64
keyed by ``fnv1a("<op>.<attr>")``), so the same seeds give the same bytes on
any machine; nothing is trained, nothing is downloaded.
"""
from __future__ import annotations

import math
import zipfile
from typing import Dict, List, Sequence, Tuple

import numpy as np

_U64 = np.uint64


def fnv1a64(s: str) -> int:
    h = 1469598103934665603
    for ch in s.encode():
        h ^= ch
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix_uniform(seed: int, count: int) -> np.ndarray:
    """``count`` floats in [0,1): splitmix64 of (seed + (i+1)*golden), top 24 bits."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=_U64)
        z = _U64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        z = z ^ (z >> _U64(31))
    return ((z >> _U64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def seeded_uniform(key: str, shape: Sequence[int], lo: float, hi: float, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = splitmix_uniform(fnv1a64(key) ^ (seed * 0x2545F4914F6CDD1D), n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def synth_input(shape_nhwc: Sequence[int], seed: int = 1) -> np.ndarray:
    """NHWC fp32 U[0,1) -- mimics the /255 normalisation of test_yolo.cpp:252."""
    return seeded_uniform("input", shape_nhwc, 0.0, 1.0, seed)


def _fmt_shape(shape: Sequence[int]) -> str:
    return "(" + ",".join(str(int(s)) for s in shape) + ")f32"


def _fmt_val(v) -> str:
    if isinstance(v, bool):
        return "True" if v else "False"
    if isinstance(v, int):
        return str(v)
    if isinstance(v, float):
        return "%e" % v
    if isinstance(v, str):
        return v
    if isinstance(v, (tuple, list)):
        def one(x):
            if isinstance(x, float):
                return repr(float(x)) if "." in repr(float(x)) or "e" in repr(float(x)) else "%.1f" % x
            return str(x)
        return "(" + ",".join(one(x) for x in v) + ")"
    raise TypeError(type(v))


class PnnxBuilder:
    """Accumulates operators/operands in NCHW (the file's convention) and writes the pair."""

    def __init__(self, seed: int = 0):
        self.seed = seed
        self.lines: List[str] = []
        self.attrs: Dict[str, np.ndarray] = {}
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        self._n_operand = 0
        self._counts: Dict[str, int] = {}
        self.n_ops = 0

    # -- plumbing -----------------------------------------------------------
    def _new_operand(self, shape: Sequence[int]) -> str:
        name = str(self._n_operand)
        self._n_operand += 1
        self.shapes[name] = tuple(int(s) for s in shape)
        return name

    def _opname(self, prefix: str) -> str:
        i = self._counts.get(prefix, 0)
        self._counts[prefix] = i + 1
        return "%s_%d" % (prefix, i)

    def _emit(self, typ: str, name: str, ins: Sequence[str], outs: Sequence[str],
              params: Dict[str, object] = None, attrs: Dict[str, np.ndarray] = None):
        toks = [typ, name, str(len(ins)), str(len(outs)), *ins, *outs]
        for k, v in (params or {}).items():
            toks.append("%s=%s" % (k, _fmt_val(v)))
        for k, arr in (attrs or {}).items():
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            toks.append("@%s=%s" % (k, _fmt_shape(arr.shape)))
            self.attrs["%s.%s" % (name, k)] = arr
        for r in list(ins) + list(outs):
            toks.append("#%s=%s" % (r, _fmt_shape(self.shapes[r])))
        self.lines.append(" ".join(toks))
        self.n_ops += 1

    # -- operators ----------------------------------------------------------
    def input(self, shape_nchw: Sequence[int]) -> str:
        out = self._new_operand(shape_nchw)
        self._emit("pnnx.Input", self._opname("pnnx_input"), [], [out])
        return out

    def output(self, x: str):
        self._emit("pnnx.Output", self._opname("pnnx_output"), [x], [])

    def conv(self, x: str, cout: int, k, s=1, p=None, d=1, groups: int = 1, bias: bool = True,
             name: str = None) -> str:
        n, cin, h, w = self.shapes[x]
        kh, kw = (k, k) if isinstance(k, int) else k
        sh, sw = (s, s) if isinstance(s, int) else s
        dh, dw = (d, d) if isinstance(d, int) else d
        if p is None:
            p = (kh // 2, kw // 2)
        ph, pw = (p, p) if isinstance(p, int) else p
        oh = (h + 2 * ph - ((kh - 1) * dh + 1)) // sh + 1
        ow = (w + 2 * pw - ((kw - 1) * dw + 1)) // sw + 1
        name = name or self._opname("conv")
        fan_in = (cin // groups) * kh * kw
        a = math.sqrt(3.0 / fan_in)
        attrs = {"weight": seeded_uniform(name + ".weight", (cout, cin // groups, kh, kw), -a, a, self.seed)}
        if bias:
            attrs["bias"] = seeded_uniform(name + ".bias", (cout,), -0.1, 0.1, self.seed)
        out = self._new_operand((n, cout, oh, ow))
        self._emit("nn.Conv2d", name, [x], [out],
                   dict(bias=bool(bias), dilation=(dh, dw), groups=groups, in_channels=cin,
                        kernel_size=(kh, kw), out_channels=cout, padding=(ph, pw),
                        padding_mode="zeros", stride=(sh, sw)), attrs)
        return out

    def _unary(self, typ: str, prefix: str, x: str, params=None) -> str:
        out = self._new_operand(self.shapes[x])
        self._emit(typ, self._opname(prefix), [x], [out], params or {})
        return out

    def silu(self, x): return self._unary("nn.SiLU", "silu", x)
    def relu(self, x): return self._unary("nn.ReLU", "relu", x)
    def sigmoid(self, x): return self._unary("nn.Sigmoid", "sigmoid", x)
    def hardsigmoid(self, x): return self._unary("nn.Hardsigmoid", "hsigmoid", x)
    def hardswish(self, x): return self._unary("nn.Hardswish", "hswish", x)

    def maxpool(self, x: str, k: int, s: int, p: int) -> str:
        n, c, h, w = self.shapes[x]
        oh = (h + 2 * p - k) // s + 1
        ow = (w + 2 * p - k) // s + 1
        out = self._new_operand((n, c, oh, ow))
        self._emit("nn.MaxPool2d", self._opname("maxpool"), [x], [out],
                   dict(ceil_mode=False, dilation=(1, 1), kernel_size=(k, k), padding=(p, p),
                        return_indices=False, stride=(s, s)))
        return out

    def adaptive_avgpool(self, x: str, out_hw=(1, 1)) -> str:
        n, c, h, w = self.shapes[x]
        out = self._new_operand((n, c, out_hw[0], out_hw[1]))
        self._emit("nn.AdaptiveAvgPool2d", self._opname("avgpool"), [x], [out],
                   dict(output_size=(int(out_hw[0]), int(out_hw[1]))))
        return out

    def upsample(self, x: str, scale: float = 2.0) -> str:
        n, c, h, w = self.shapes[x]
        out = self._new_operand((n, c, int(h * scale), int(w * scale)))
        self._emit("nn.Upsample", self._opname("upsample"), [x], [out],
                   dict(mode="nearest", scale_factor=(float(scale), float(scale)), size="None"))
        return out

    def cat(self, xs: Sequence[str], dim: int = 1) -> str:
        shp = list(self.shapes[xs[0]])
        shp[dim] = sum(self.shapes[x][dim] for x in xs)
        out = self._new_operand(shp)
        self._emit("torch.cat", self._opname("cat"), list(xs), [out], dict(dim=dim))
        return out

    def expression(self, expr: str, xs: Sequence[str], out_shape=None) -> str:
        out = self._new_operand(out_shape or self.shapes[xs[0]])
        self._emit("pnnx.Expression", self._opname("pnnx_expr"), list(xs), [out], dict(expr=expr))
        return out

    def add(self, a: str, b: str) -> str:
        return self.expression("add(@0,@1)", [a, b])

    def mul(self, a: str, b: str) -> str:
        sa, sb = self.shapes[a], self.shapes[b]
        return self.expression("mul(@0,@1)", [a, b], tuple(max(x, y) for x, y in zip(sa, sb)))

    def batchnorm(self, x: str, eps: float = 1e-5) -> str:
        n, c, h, w = self.shapes[x]
        name = self._opname("bn")
        attrs = dict(
            running_mean=seeded_uniform(name + ".running_mean", (c,), -0.5, 0.5, self.seed),
            running_var=seeded_uniform(name + ".running_var", (c,), 0.5, 1.5, self.seed),
            weight=seeded_uniform(name + ".weight", (c,), 0.5, 1.5, self.seed),
            bias=seeded_uniform(name + ".bias", (c,), -0.5, 0.5, self.seed))
        out = self._new_operand((n, c, h, w))
        self._emit("nn.BatchNorm2d", name, [x], [out], dict(affine=True, eps=float(eps), num_features=c), attrs)
        return out

    def flatten(self, x: str) -> str:
        shp = self.shapes[x]
        out = self._new_operand((shp[0], int(np.prod(shp[1:]))))
        self._emit("torch.flatten", self._opname("flatten"), [x], [out], dict(end_dim=-1, start_dim=1))
        return out

    def linear(self, x: str, out_f: int, bias: bool = True) -> str:
        n, in_f = self.shapes[x]
        name = self._opname("linear")
        a = math.sqrt(3.0 / in_f)
        attrs = {"weight": seeded_uniform(name + ".weight", (out_f, in_f), -a, a, self.seed),
                 # the reference requires the attribute even for bias=False
                 "bias": seeded_uniform(name + ".bias", (out_f,), -0.1, 0.1, self.seed)}
        out = self._new_operand((n, out_f))
        self._emit("nn.Linear", name, [x], [out], dict(bias=bool(bias), in_features=in_f, out_features=out_f), attrs)
        return out

    def detect(self, xs: Sequence[str], strides=(8.0, 16.0, 32.0), anchors=None, nc: int = 80) -> str:
        """models.yolo.Detect -- attribute names per reference src/layer/yolo_detect.cpp:19-145."""
        anchors = anchors or [[(10, 13), (16, 30), (33, 23)], [(30, 61), (62, 45), (59, 119)],
                              [(116, 90), (156, 198), (373, 326)]]
        ne, na = nc + 5, 3
        name = self._opname("detect")
        attrs = {"pnnx_5": np.asarray(strides, dtype=np.float32)}
        anchor_idx, grid_idx = (4, 2, 0), (6, 3, 1)
        rows = 0
        n = self.shapes[xs[0]][0]
        for i, x in enumerate(xs):
            _, c, h, w = self.shapes[x]
            a = math.sqrt(3.0 / c)
            attrs["m.%d.weight" % i] = seeded_uniform("%s.m.%d.weight" % (name, i), (na * ne, c, 1, 1), -a, a, self.seed)
            attrs["m.%d.bias" % i] = seeded_uniform("%s.m.%d.bias" % (name, i), (na * ne,), -0.1, 0.1, self.seed)
            gy, gx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
            grid = np.stack([gx - 0.5, gy - 0.5], axis=-1)  # [h,w,2], (x,y)
            attrs["pnnx_%d" % grid_idx[i]] = np.broadcast_to(grid[None, None], (1, na, h, w, 2)).copy()
            ag = np.asarray(anchors[i], dtype=np.float32).reshape(1, na, 1, 1, 2)
            attrs["pnnx_%d" % anchor_idx[i]] = np.broadcast_to(ag, (1, na, h, w, 2)).copy()
            rows += na * h * w
        out = self._new_operand((n, rows, ne))
        self._emit("models.yolo.Detect", name, list(xs), [out], {}, attrs)
        return out

    # -- writing ------------------------------------------------------------
    def save(self, param_path: str, bin_path: str):
        with open(param_path, "w") as f:
            f.write("7767517\n%d %d\n" % (self.n_ops, self._n_operand))
            for ln in self.lines:
                f.write(ln + "\n")
        with zipfile.ZipFile(bin_path, "w", compression=zipfile.ZIP_STORED) as z:
            for k, arr in self.attrs.items():
                zi = zipfile.ZipInfo(k, date_time=(1980, 1, 1, 0, 0, 0))
                zi.compress_type = zipfile.ZIP_STORED
                z.writestr(zi, arr.astype("<f4").tobytes())


# ---------------------------------------------------------------------------
# YOLOv5 building blocks
# ---------------------------------------------------------------------------
def _Conv(b: PnnxBuilder, x, c2, k=1, s=1):
    return b.silu(b.conv(x, c2, k, s, k // 2))


def _C3(b: PnnxBuilder, x, c2, n=1, shortcut=True):
    c_ = c2 // 2
    y1 = _Conv(b, x, c_, 1)
    for _ in range(n):
        t = _Conv(b, _Conv(b, y1, c_, 1), c_, 3)
        y1 = b.add(y1, t) if shortcut else t
    y2 = _Conv(b, x, c_, 1)
    return _Conv(b, b.cat([y1, y2], 1), c2, 1)


def _SPPF(b: PnnxBuilder, x, c2, k=5):
    c1 = b.shapes[x][1]
    x = _Conv(b, x, c1 // 2, 1)
    y1 = b.maxpool(x, k, 1, k // 2)
    y2 = b.maxpool(y1, k, 1, k // 2)
    y3 = b.maxpool(y2, k, 1, k // 2)
    return _Conv(b, b.cat([x, y1, y2, y3], 1), c2, 1)


def build_yolov5s(batch: int, size: int = 640, seed: int = 0, width: float = 0.5,
                  depth: float = 1.0 / 3.0, nc: int = 80) -> PnnxBuilder:
    def ch(c):
        return int(math.ceil(c * width / 8) * 8)

    def dn(n):
        return max(round(n * depth), 1)

    b = PnnxBuilder(seed)
    x = b.input((batch, 3, size, size))
    x = b.silu(b.conv(x, ch(64), 6, 2, 2))         # 0: Conv(3,32,k6,s2,p2)
    x = _Conv(b, x, ch(128), 3, 2)                 # 1
    x = _C3(b, x, ch(128), dn(3))                  # 2
    x = _Conv(b, x, ch(256), 3, 2)                 # 3
    p3 = _C3(b, x, ch(256), dn(6))                 # 4
    x = _Conv(b, p3, ch(512), 3, 2)                # 5
    p4 = _C3(b, x, ch(512), dn(9))                 # 6
    x = _Conv(b, p4, ch(1024), 3, 2)               # 7
    x = _C3(b, x, ch(1024), dn(3))                 # 8
    x = _SPPF(b, x, ch(1024), 5)                   # 9
    h10 = _Conv(b, x, ch(512), 1, 1)               # 10
    x = b.upsample(h10, 2.0)                       # 11
    x = b.cat([x, p4], 1)                          # 12
    x = _C3(b, x, ch(512), dn(3), False)           # 13
    h14 = _Conv(b, x, ch(256), 1, 1)               # 14
    x = b.upsample(h14, 2.0)                       # 15
    x = b.cat([x, p3], 1)                          # 16
    d3 = _C3(b, x, ch(256), dn(3), False)          # 17 -> P3
    x = _Conv(b, d3, ch(256), 3, 2)                # 18
    x = b.cat([x, h14], 1)                         # 19
    d4 = _C3(b, x, ch(512), dn(3), False)          # 20 -> P4
    x = _Conv(b, d4, ch(512), 3, 2)                # 21
    x = b.cat([x, h10], 1)                         # 22
    d5 = _C3(b, x, ch(1024), dn(3), False)         # 23 -> P5
    out = b.detect([d3, d4, d5], nc=nc)            # 24
    b.output(out)
    return b


def build_resnet18(batch: int, size: int = 224, num_classes: int = 1000, seed: int = 0,
                   base: int = 64) -> PnnxBuilder:
    """torchvision ResNet18 with BN folded into the convs."""
    b = PnnxBuilder(seed)
    x = b.input((batch, 3, size, size))
    x = b.relu(b.conv(x, base, 7, 2, 3))
    x = b.maxpool(x, 3, 2, 1)
    cin = base
    for stage, c in enumerate((base, base * 2, base * 4, base * 8)):
        for blk in range(2):
            s = 2 if (stage > 0 and blk == 0) else 1
            idt = x
            y = b.relu(b.conv(x, c, 3, s, 1))
            y = b.conv(y, c, 3, 1, 1)
            if s != 1 or cin != c:
                idt = b.conv(x, c, 1, s, 0)
            x = b.relu(b.add(y, idt))
            cin = c
    x = b.adaptive_avgpool(x, (1, 1))
    x = b.flatten(x)
    x = b.linear(x, num_classes)
    b.output(x)
    return b


def build_toy_yolo(batch: int = 2, size: int = 64, seed: int = 0) -> PnnxBuilder:
    """A narrow YOLOv5 (width 0.125 -> channels 8..128) for fast graph-level parity tests."""
    return build_yolov5s(batch, size, seed, width=0.125, depth=1.0 / 3.0, nc=3)


def build_toy_classifier(batch: int = 2, size: int = 32, seed: int = 0) -> PnnxBuilder:
    """Small net touching the MobileNet-side ops: BN, hardswish, hardsigmoid, SE-style broadcast mul,
    grouped conv, sigmoid, avgpool, flatten, linear."""
    b = PnnxBuilder(seed)
    x = b.input((batch, 3, size, size))
    x = b.hardswish(b.batchnorm(b.conv(x, 16, 3, 2, 1, bias=False)))
    y = b.relu(b.conv(x, 16, 3, 1, 1, groups=16))          # depthwise
    se = b.adaptive_avgpool(y, (1, 1))
    se = b.relu(b.conv(se, 8, 1))
    se = b.hardsigmoid(b.conv(se, 16, 1))
    y = b.mul(y, se)                                        # broadcast over H, W
    y = b.conv(y, 16, 1)
    x = b.add(x, y)
    x = b.sigmoid(b.conv(x, 24, 3, 2, 1, groups=2))
    x = b.adaptive_avgpool(x, (1, 1))
    x = b.flatten(x)
    x = b.linear(x, 10)
    b.output(x)
    return b


def build_mobilenetv3_small(batch: int, size: int = 224, num_classes: int = 1000, seed: int = 0) -> PnnxBuilder:
    """torchvision MobileNetV3-Small topology with BatchNorm folded into the convolutions (nn.Conv2d bias=True), the
    model family of the reference's test_classify (test/test_classify/test_classify.cpp:12-15): 3x3 / 5x5 depthwise
    convs, squeeze-excite (global average pool -> 1x1 -> ReLU -> 1x1 -> Hardsigmoid -> broadcast mul), Hardswish,
    residual adds."""
    b = PnnxBuilder(seed)

    def act(x, kind):
        return b.hardswish(x) if kind == "HS" else b.relu(x)

    def make_div(v, d=8):
        return max(d, int(v + d / 2) // d * d)

    x = b.input((batch, 3, size, size))
    x = b.hardswish(b.conv(x, 16, 3, 2, 1))
    cin = 16
    # kernel, expanded channels, output channels, squeeze-excite, activation, stride
    cfg = [(3, 16, 16, True, "RE", 2), (3, 72, 24, False, "RE", 2), (3, 88, 24, False, "RE", 1), (5, 96, 40, True, "HS", 2),
           (5, 240, 40, True, "HS", 1), (5, 240, 40, True, "HS", 1), (5, 120, 48, True, "HS", 1), (5, 144, 48, True, "HS", 1),
           (5, 288, 96, True, "HS", 2), (5, 576, 96, True, "HS", 1), (5, 576, 96, True, "HS", 1)]
    for k, exp, cout, se, nl, s in cfg:
        y = x
        if exp != cin:
            y = act(b.conv(y, exp, 1), nl)
        y = act(b.conv(y, exp, k, s, k // 2, groups=exp), nl)          # depthwise
        if se:
            q = b.adaptive_avgpool(y, (1, 1))
            q = b.relu(b.conv(q, make_div(exp // 4), 1))
            q = b.hardsigmoid(b.conv(q, exp, 1))
            y = b.mul(y, q)                                            # broadcast over H, W
        y = b.conv(y, cout, 1)
        x = b.add(x, y) if (s == 1 and cin == cout) else y
        cin = cout
    x = b.hardswish(b.conv(x, 576, 1))
    x = b.adaptive_avgpool(x, (1, 1))
    x = b.flatten(x)
    x = b.hardswish(b.linear(x, 1024))
    x = b.linear(x, num_classes)
    b.output(x)
    return b


def conv_flops(builder: PnnxBuilder) -> int:
    """Direct-convolution FLOPs (2*MAC) of every nn.Conv2d + Detect 1x1 conv in the graph
    (sum N*OH*OW*KH*KW*(Cin/g)*Cout)."""
    total = 0
    for ln in builder.lines:
        t = ln.split()
        if t[0] == "nn.Conv2d":
            kv = dict(x.split("=", 1) for x in t[4 + int(t[2]) + int(t[3]):])
            out = t[4 + int(t[2])]
            n, co, oh, ow = builder.shapes[out]
            kh, kw = (int(v) for v in kv["kernel_size"].strip("()").split(","))
            total += 2 * n * oh * ow * kh * kw * (int(kv["in_channels"]) // int(kv["groups"])) * co
        elif t[0] == "models.yolo.Detect":
            nin = int(t[2])
            out = t[4 + nin]
            ne = builder.shapes[out][2]
            for x in t[4:4 + nin]:
                n, c, h, w = builder.shapes[x]
                total += 2 * n * h * w * c * 3 * ne
    return total
