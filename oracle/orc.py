"""ctypes front-end of oracle/liboracle.so + a whole-graph CPU oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg -- never by the product package.

``run_graph`` walks a ``.pnnx.param/.bin`` pair with its *own* small parser
(independent of the product's C++ loader) and executes every operator with the
C restatement in ``si_oracle.c`` in the order and semantics of the reference's
``EngineImpl`` (NCHW file shapes -> NHWC tensors, reference
``src/engine_impl.cpp:182-189``; ``pnnx.Expression`` lowered as
``src/pnnx/expand_expression.cpp`` does for add/mul).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import zipfile
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None

F32P = C.POINTER(C.c_float)


class OrcConv2d(C.Structure):
    _fields_ = [(k, C.c_int) for k in
                ("n", "ih", "iw", "ic", "oc", "kh", "kw", "sh", "sw", "dh", "dw",
                 "pt", "pb", "pl", "pr", "groups", "use_bias")]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "si_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(F32P)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i4(shape: Sequence[int]):
    return (C.c_int * 4)(*[int(s) for s in shape])


def pad4(shape: Sequence[int]) -> List[int]:
    """rank-adapting view of reference include/eigen_helper.h:32-63 (leading dims folded / padded with 1)."""
    shape = list(shape)
    if len(shape) >= 4:
        lead = int(np.prod(shape[:len(shape) - 3]))
        return [lead] + shape[-3:]
    return [1] * (4 - len(shape)) + shape


# ---------------------------------------------------------------------------
# per-op wrappers (NHWC numpy in / out)
# ---------------------------------------------------------------------------
def conv_desc(x_shape, w_shape, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1, use_bias=True):
    n, ih, iw, ic = x_shape
    oc, _, kh, kw = w_shape
    d = OrcConv2d(n, ih, iw, ic, oc, kh, kw, stride[0], stride[1], dilation[0], dilation[1],
                  padding[0], padding[0], padding[1], padding[1], groups, 1 if use_bias else 0)
    oh, ow = C.c_int(), C.c_int()
    lib().orc_conv2d_out_shape(C.byref(d), C.byref(oh), C.byref(ow))
    return d, oh.value, ow.value


def conv2d(x, w_oihw, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1, path="auto",
           q1_bug=False, acc64=True):
    """path: auto (reference dispatch) | im2col | winograd | naive | chain (the device kernel's fma order)"""
    x, w_oihw = _f32(x), _f32(w_oihw)
    bias = None if bias is None else _f32(bias)
    d, oh, ow = conv_desc(x.shape, w_oihw.shape, stride, padding, dilation, groups, bias is not None)
    out = np.empty((x.shape[0], oh, ow, w_oihw.shape[0]), np.float32)
    L = lib()
    if path == "auto":
        rc = L.orc_conv2d_forward(C.byref(d), _p(x), _p(w_oihw), _p(bias), _p(out))
    elif path == "im2col":
        rc = L.orc_conv2d_im2col(C.byref(d), _p(x), _p(w_oihw), _p(bias), _p(out))
    elif path == "winograd":
        rc = L.orc_conv2d_winograd23(C.byref(d), _p(x), _p(w_oihw), _p(bias), _p(out), 1 if q1_bug else 0)
    elif path == "naive":
        rc = L.orc_conv2d_naive(C.byref(d), _p(x), _p(w_oihw), _p(bias), _p(out), 1 if acc64 else 0)
    elif path == "chain":   # the device implicit-GEMM kernel's fma chain (bit-exact predictor, not a reference algorithm)
        rc = L.orc_conv2d_chain(C.byref(d), _p(x), _p(w_oihw), _p(bias), _p(out))
    else:
        raise ValueError(path)
    if rc != 0:
        raise RuntimeError("oracle conv2d(%s) rc=%d" % (path, rc))
    return out


def gemm_pack4(A, Bp, M, N, K, ref=False):
    A, Bp = _f32(A), _f32(Bp)
    Cm = np.zeros((M, N), np.float32)
    fn = lib().orc_gemm_pack4_f32_ref if ref else lib().orc_gemm_pack4_f32
    fn(C.c_size_t(M), C.c_size_t(N), C.c_size_t(K), _p(A), C.c_size_t(K), _p(Bp), _p(Cm), C.c_size_t(N))
    return Cm


def linear(x, w, b=None):
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    y = np.empty((x.shape[0], w.shape[0]), np.float32)
    lib().orc_linear(_p(x), x.shape[0], x.shape[1], _p(w), _p(b), w.shape[0], _p(y))
    return y


def maxpool2d(x, k, s, p, d=(1, 1)):
    x = _f32(x)
    n, ih, iw, c = x.shape
    oh = (ih + 2 * p[0] - ((k[0] - 1) * d[0] + 1)) // s[0] + 1
    ow = (iw + 2 * p[1] - ((k[1] - 1) * d[1] + 1)) // s[1] + 1
    out = np.empty((n, oh, ow, c), np.float32)
    lib().orc_maxpool2d(_p(x), n, ih, iw, c, k[0], k[1], s[0], s[1], d[0], d[1], p[0], p[1], _p(out), oh, ow)
    return out


def adaptive_avgpool2d(x, out_hw):
    x = _f32(x)
    n, ih, iw, c = x.shape
    out = np.empty((n, out_hw[0], out_hw[1], c), np.float32)
    rc = lib().orc_adaptive_avgpool2d(_p(x), n, ih, iw, c, _p(out), out_hw[0], out_hw[1])
    if rc != 0:
        raise RuntimeError("adaptive_avgpool2d unsupported shape")
    return out


def upsample_nearest(x, scale_h, scale_w, out_hw=None):
    x = _f32(x)
    n, ih, iw, c = x.shape
    oh, ow = out_hw if out_hw else (int(ih * scale_h), int(iw * scale_w))
    out = np.empty((n, oh, ow, c), np.float32)
    lib().orc_upsample_nearest(_p(x), n, ih, iw, c, C.c_float(scale_h), C.c_float(scale_w), _p(out), oh, ow)
    return out


def cat(xs, axis):
    """axis is the NHWC axis (reference cat.cpp:75-84 maps NCHW dim 1->3, 2->1, 3->2)."""
    xs = [_f32(x) for x in xs]
    shp = list(xs[0].shape)
    shp[axis] = sum(x.shape[axis] for x in xs)
    out = np.empty(shp, np.float32)
    off = 0
    for x in xs:
        lib().orc_cat_axis(_p(x), _i4(x.shape), _p(out), _i4(shp), axis, off)
        off += x.shape[axis]
    return out


def binary_op(op, a, b, out_shape=None):
    a, b = _f32(a), _f32(b)
    a4, b4 = pad4(a.shape), pad4(b.shape)
    o4 = pad4(out_shape) if out_shape is not None else [max(x, y) for x, y in zip(a4, b4)]
    out = np.empty(o4, np.float32)
    rc = lib().orc_binary_op(op, _p(a), _i4(a4), _p(b), _i4(b4), _p(out), _i4(o4))
    if rc != 0:
        raise RuntimeError("binary_op rc=%d" % rc)
    return out.reshape(out_shape if out_shape is not None else o4)


ACT = {"relu": 1, "silu": 2, "sigmoid": 3, "hardsigmoid": 4, "hardswish": 5}


def activation(kind, x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_activation(ACT[kind], _p(x), _p(out), C.c_size_t(x.size))
    return out


def batchnorm2d(x, mean, var, gamma, beta, eps):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_batchnorm2d(_p(x), C.c_size_t(x.size // x.shape[-1]), x.shape[-1], _p(_f32(mean)), _p(_f32(var)),
                          _p(_f32(gamma)), _p(_f32(beta)), C.c_float(eps), _p(out))
    return out


def flatten_nhwc(x):
    x = _f32(x)
    n, h, w, c = x.shape
    out = np.empty((n, c * h * w), np.float32)
    lib().orc_flatten_nhwc(_p(x), n, h, w, c, _p(out))
    return out


def yolo_detect(feats, weights, biases, grids, anchor_grids, strides, na=3):
    feats = [_f32(f) for f in feats]
    n = feats[0].shape[0]
    ne = weights[0].shape[0] // na
    rows_total = sum(f.shape[1] * f.shape[2] * na for f in feats)
    out = np.empty((n, rows_total, ne), np.float32)
    off = 0
    for f, w, b, g, a, s in zip(feats, weights, biases, grids, anchor_grids, strides):
        _, h, wd, cin = f.shape
        lib().orc_yolo_detect_level(_p(f), n, h, wd, cin, _p(_f32(w)), _p(_f32(b)), na, ne, _p(_f32(g)),
                                    _p(_f32(a)), C.c_float(float(s)), _p(out), rows_total, off)
        off += h * wd * na
    return out


def letterbox_geometry(height_origin, width_origin, height_new, width_new):
    hr, wr, pt, pl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    sc = C.c_float()
    lib().orc_letterbox_geometry(height_origin, width_origin, height_new, width_new, C.byref(hr), C.byref(wr),
                                 C.byref(sc), C.byref(pt), C.byref(pl))
    return hr.value, wr.value, sc.value, pt.value, pl.value


def letterbox(resized_bgr, height_new, width_new, padding_t, padding_l):
    src = np.ascontiguousarray(resized_bgr, dtype=np.uint8)
    out = np.empty((height_new, width_new, 3), np.float32)
    lib().orc_letterbox_u8(src.ctypes.data_as(C.c_void_p), int(src.shape[0]), int(src.shape[1]), _p(out), height_new,
                           width_new, padding_t, padding_l)
    return out


def resize_bilinear_u8c3(image, dst_h, dst_w):
    """The cv::resize of PreProcess (test/test_yolo/test_yolo.cpp:213-216), 8-bit, 3 channels, INTER_LINEAR.  PARITY UNPINNED
    against the reference: cv::resize lives in its simpleocv submodule, which is absent from /root/reference (empty 3rdparty
    directory).  This restates the PUBLISHED fixed-point algorithm that library shares with OpenCV and ncnn (resize_bilinear_c3):
    half-pixel centres, f = (d + 0.5) * (src / dst) - 0.5 in double rounded to float, s = floor(f) clamped (left: f = 0; right:
    s = src - 2, f = 1), weights a = (short)(int)(w * 2048 + 0.5), horizontal pass (S[s] * a0 + S[s + 1] * a1) >> 4, vertical
    pass (((b0 * r0) >> 16) + ((b1 * r1) >> 16) + 2) >> 2.  Plain numpy integer arithmetic; image u8 [h][w][3]."""
    src = np.ascontiguousarray(image, dtype=np.uint8)
    sh, sw = int(src.shape[0]), int(src.shape[1])

    def axis(dst, n_src):
        d = np.arange(dst, dtype=np.float64)
        f = ((d + 0.5) * (np.float64(n_src) / np.float64(dst)) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        s[lo], f[lo] = 0, 0.0
        hi = s >= n_src - 1
        s[hi], f[hi] = n_src - 2, 1.0
        if n_src == 1:
            s[:], f[:] = 0, 0.0
        a0 = ((np.float32(1.0) - f) * np.float32(2048.0) + np.float32(0.5)).astype(np.int64)
        a1 = (f * np.float32(2048.0) + np.float32(0.5)).astype(np.int64)
        return s, np.minimum(s + 1, n_src - 1) if n_src > 1 else s, a0, a1

    x0, x1, ax0, ax1 = axis(dst_w, sw)
    y0, y1, ay0, ay1 = axis(dst_h, sh)
    S = src.astype(np.int64)
    rows = (S[:, x0, :] * ax0[None, :, None] + S[:, x1, :] * ax1[None, :, None]) >> 4          # [sh][dst_w][3], fits 16 bits
    r0, r1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * r0) >> 16) + ((ay1[:, None, None] * r1) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def yolo_postprocess(pred, prob_threshold=0.25, nms_threshold=0.45, agnostic=False, adjust=None, max_det=None):
    """test_yolo.cpp:337-428 per image; returns (list of [k][6] arrays, counts)."""
    pred = _f32(pred)
    n, rows, ne = pred.shape
    if max_det is None:
        max_det = max(rows, 1)
    L = lib()
    L.orc_yolo_postprocess.restype = C.c_int
    outs, cnts = [], np.zeros((n,), np.int32)
    for b in range(n):
        dets = np.zeros((max(max_det, 1), 6), np.float32)
        adj = _f32(adjust[b]) if adjust is not None else None
        k = L.orc_yolo_postprocess(_p(pred[b]), rows, ne, C.c_float(prob_threshold), C.c_float(nms_threshold),
                                   int(bool(agnostic)), _p(adj), _p(dets), max_det)
        cnts[b] = k
        outs.append(dets[:min(k, max_det)].copy())
    return outs, cnts


# ---------------------------------------------------------------------------
# independent .pnnx.param/.bin reader + graph walk
# ---------------------------------------------------------------------------
def _parse_value(v: str):
    """reference src/pnnx/ir.cpp:479-550"""
    if v in ("None", "()", "[]"):
        return None
    if v in ("True", "False"):
        return v == "True"
    if v[0] in "([":
        items = v[1:-1].split(",")
        out = []
        for e in items:
            if not (e[0].isdigit() or (e[0] == "-" and len(e) > 1 and e[1].isdigit())):
                out.append(e)
            elif "." in e or "e" in e:
                out.append(float(e))
            else:
                out.append(int(e))
        return out
    if not (v[0].isdigit() or (v[0] == "-" and len(v) > 1 and v[1].isdigit())):
        return v
    if "." in v or "e" in v:
        return float(np.float32(v))
    return int(v)


class Op:
    def __init__(self, typ, name, ins, outs):
        self.type, self.name, self.inputs, self.outputs = typ, name, ins, outs
        self.params: Dict[str, object] = {}
        self.attrs: Dict[str, np.ndarray] = {}


def load_pnnx(param_path: str, bin_path: str):
    ops: List[Op] = []
    shapes: Dict[str, List[int]] = {}
    z = zipfile.ZipFile(bin_path)
    names = set(z.namelist())
    with open(param_path) as f:
        assert int(f.readline().split()[0]) == 7767517
        n_ops = int(f.readline().split()[0])
        for _ in range(n_ops):
            t = f.readline().split()
            ni, no = int(t[2]), int(t[3])
            op = Op(t[0], t[1], t[4:4 + ni], t[4 + ni:4 + ni + no])
            for kv in t[4 + ni + no:]:
                k, v = kv.split("=", 1)
                if k[0] == "@":
                    shp = [int(s) for s in v[1:v.rindex(")")].split(",")]
                    key = "%s.%s" % (op.name, k[1:])
                    if key in names:
                        op.attrs[k[1:]] = np.frombuffer(z.read(key), dtype="<f4").reshape(shp).copy()
                elif k[0] == "#":
                    shapes[k[1:]] = [(-1 if s == "?" else int(s)) for s in v[1:v.rindex(")")].split(",")]
                elif k[0] == "$":
                    pass
                else:
                    op.params[k] = _parse_value(v)
            ops.append(op)
    return ops, shapes


def nhwc_shape(shape_nchw: Sequence[int]) -> List[int]:
    """reference src/engine_impl.cpp:182-189"""
    s = list(shape_nchw)
    if len(s) > 3:
        d = len(s)
        s[d - 1], s[d - 2], s[d - 3] = shape_nchw[d - 3], shape_nchw[d - 1], shape_nchw[d - 2]
    return s


UNARY_CODES = {"abs": 0, "neg": 1, "floor": 2, "ceil": 3, "square": 4, "sqrt": 5, "rsqrt": 6, "exp": 7, "log": 8, "sin": 9,
               "cos": 10, "tan": 11, "asin": 12, "acos": 13, "atan": 14, "reciprocal": 15, "tanh": 16, "log10": 17}
BINARY_CODES = {"add": 0, "sub": 1, "mul": 2, "div": 3, "pow": 6, "atan2": 10}
BINARY_REVERSED = {"sub": 7, "div": 8, "pow": 9, "atan2": 11}


def unary_op(op: int, x: np.ndarray) -> np.ndarray:
    x = _f32(x)
    out = np.empty_like(x)
    rc = lib().orc_unary_op(op, _p(x), _p(out), C.c_size_t(x.size))
    if rc != 0:
        raise RuntimeError("unary_op rc=%d" % rc)
    return out


def binary_scalar(op: int, x: np.ndarray, scalar: float) -> np.ndarray:
    x = _f32(x)
    out = np.empty_like(x)
    rc = lib().orc_binary_scalar(op, _p(x), C.c_float(scalar), _p(out), C.c_size_t(x.size))
    if rc != 0:
        raise RuntimeError("binary_scalar rc=%d" % rc)
    return out


def _eval_expr(expr: str, args: List[np.ndarray], out_shape):
    """A pnnx.Expression evaluated as the reference's loader lowers it (src/pnnx/expand_expression.cpp:65-307): prefix
    expression scanned right to left; unary functions -> UnaryOp, binary ones -> BinaryOp, a literal operand -> the scalar
    form (the literal first: the operand-reversed code), pow(x, 2) -> square.  (The reference's BinaryOp layer then only runs
    add and mul of two tensors, src/layer/binary_op.cpp:17-31; the rest has float libm semantics here.)"""
    toks = expr.replace("(", " ").replace(")", " ").replace(",", " ").split()
    stack: list = []
    is_lit = lambda v: isinstance(v, float)
    for t in reversed(toks):
        if t in UNARY_CODES:
            a = stack.pop()
            stack.append(unary_op(UNARY_CODES[t], a))
        elif t in BINARY_CODES:
            a, b = stack.pop(), stack.pop()
            if is_lit(a) and is_lit(b):
                raise NotImplementedError("constant folding is pnnx's job, not the loader's")
            if is_lit(a):
                stack.append(binary_scalar(BINARY_REVERSED.get(t, BINARY_CODES[t]), b, a))
            elif is_lit(b):
                stack.append(unary_op(4, a) if (t == "pow" and b == 2.0) else binary_scalar(BINARY_CODES[t], a, b))
            else:
                o4 = [max(x, y) for x, y in zip(pad4(a.shape), pad4(b.shape))]
                stack.append(binary_op(BINARY_CODES[t], a, b, o4))
        elif t[0] == "@":
            stack.append(args[int(t[1:])])
        else:
            try:
                stack.append(float(t))
            except ValueError:
                raise NotImplementedError("expression token %r (the reference's lowering gives up on it too)" % t)
    return stack.pop().reshape(out_shape)


def run_graph(param_path: str, bin_path: str, inputs: Dict[str, np.ndarray], keep_all: bool = False,
              conv_path: str = "auto") -> Dict[str, np.ndarray]:
    """Execute the graph on the CPU; returns {operand name: NHWC array} for outputs (or all operands)."""
    ops, shapes = load_pnnx(param_path, bin_path)
    vals: Dict[str, np.ndarray] = {}
    outputs: List[str] = []
    for op in ops:
        t, P, A = op.type, op.params, op.attrs
        x = [vals[i] for i in op.inputs] if t != "pnnx.Input" else []
        if t == "pnnx.Input":
            name = op.outputs[0]
            arr = _f32(inputs[name])
            assert list(arr.shape) == nhwc_shape(shapes[name]), (arr.shape, shapes[name])
            vals[name] = arr
            continue
        if t == "pnnx.Output":
            outputs.append(op.inputs[0])
            continue
        if t == "nn.Conv2d":
            y = conv2d(x[0], A["weight"], A.get("bias") if P["bias"] else None, P["stride"], P["padding"],
                       P["dilation"], P["groups"], path=conv_path)
        elif t in ("nn.SiLU", "nn.ReLU", "nn.Sigmoid", "nn.Hardsigmoid", "nn.Hardswish"):
            y = activation({"nn.SiLU": "silu", "nn.ReLU": "relu", "nn.Sigmoid": "sigmoid",
                            "nn.Hardsigmoid": "hardsigmoid", "nn.Hardswish": "hardswish"}[t], x[0])
        elif t == "nn.MaxPool2d":
            y = maxpool2d(x[0], P["kernel_size"], P["stride"], P["padding"], P["dilation"])
        elif t == "nn.AdaptiveAvgPool2d":
            y = adaptive_avgpool2d(x[0], P["output_size"])
        elif t == "nn.Upsample":
            oshape = nhwc_shape(shapes[op.outputs[0]])
            y = upsample_nearest(x[0], P["scale_factor"][0], P["scale_factor"][1], (oshape[1], oshape[2]))
        elif t == "torch.cat":
            y = cat(x, {1: 3, 2: 1, 3: 2}.get(P["dim"], P["dim"]))
        elif t == "pnnx.Expression":
            y = _eval_expr(P["expr"], x, nhwc_shape(shapes[op.outputs[0]]))
        elif t == "nn.BatchNorm2d":
            y = batchnorm2d(x[0], A["running_mean"], A["running_var"], A["weight"], A["bias"], P["eps"])
        elif t == "torch.flatten":
            y = flatten_nhwc(x[0]) if x[0].ndim == 4 else x[0].reshape(nhwc_shape(shapes[op.outputs[0]]))
        elif t == "nn.Linear":
            y = linear(x[0], A["weight"], A["bias"] if P["bias"] else None)
        elif t == "models.yolo.Detect":
            y = yolo_detect(x, [A["m.%d.weight" % i] for i in range(3)], [A["m.%d.bias" % i] for i in range(3)],
                            [A["pnnx_%d" % i] for i in (6, 3, 1)], [A["pnnx_%d" % i] for i in (4, 2, 0)],
                            A["pnnx_5"].tolist())
        else:
            raise NotImplementedError("operator %s is not registered in the reference (layer_registry.cpp:33-49)" % t)
        vals[op.outputs[0]] = y
    return vals if keep_all else {k: vals[k] for k in outputs}
