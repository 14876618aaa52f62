"""numpy-in / numpy-out wrappers over the kernel C-ABI (include/si_hip.h).

Each helper uploads its operands to HBM, launches exactly one C-ABI kernel entry point and downloads
the result, so the parity tests exercise the same symbols a C / cgo / JNI caller would bind.  No
computation happens in Python; without a HIP device every call raises HipError.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

import contextlib

from . import _native
from ._native import SiConvPlan, SiPool2dDesc

# The kernel-form plan attached to every conv descriptor THIS MODULE builds (tests and sweeps hold every tile / form of a kernel family to the
# same bits through it).  State of this Python test helper only: the C-ABI itself has no process-global switch -- a plan travels inside the
# descriptor of one call (include/si_hip.h SiConv2dDesc::plan).
_PLAN = None


@contextlib.contextmanager
def plan(**fields):
    """with hipops.plan(f32_tile=4): ...  -- every conv launched through this module inside the block carries SiConvPlan(**fields)"""
    global _PLAN
    prev, _PLAN = _PLAN, SiConvPlan(**fields)
    try:
        yield _PLAN
    finally:
        _PLAN = prev


def set_plan(**fields):
    """the non-scoped form of plan(): every later conv launched through this module carries SiConvPlan(**fields); no fields = no plan"""
    global _PLAN
    _PLAN = SiConvPlan(**fields) if fields else None


def SiConv2dDesc(*a):
    d = _native.SiConv2dDesc(*a)
    if _PLAN is not None:
        d.plan = C.pointer(_PLAN)
    return d

ACT = {"none": 0, "relu": 1, "silu": 2, "sigmoid": 3, "hardsigmoid": 4, "hardswish": 5, "leakyrelu": 6}


class HipError(RuntimeError):
    pass


def _chk(rc: int, what: str):
    if rc != 0:
        raise HipError("%s: %s (code %d)" % (what, _native.hip().si_hip_error_string(rc).decode(), rc))


class DeviceBuffer:
    """HBM allocation owned by Python (hipMalloc / hipFree through the C-ABI)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _chk(_native.hip().si_hip_malloc(C.byref(p), max(self.nbytes, 16)), "si_hip_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a: np.ndarray, stream=None) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        _chk(_native.hip().si_hip_memcpy_h2d(b.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, stream), "h2d")
        _chk(_native.hip().si_hip_stream_sync(stream), "sync")
        return b

    @classmethod
    def view(cls, ptr: int, nbytes: int) -> "DeviceBuffer":
        """Non-owning handle on device memory somebody else allocated (an engine output, a gathered buffer)."""
        b = cls.__new__(cls)
        b.nbytes, b.ptr, b._borrowed = int(nbytes), int(ptr), True
        return b

    def to_numpy(self, shape, dtype=np.float32, stream=None) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= max(self.nbytes, 16)
        _chk(_native.hip().si_hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, stream), "d2h")
        _chk(_native.hip().si_hip_stream_sync(stream), "sync")
        return out

    def fill(self, byte: int = 0):
        _chk(_native.hip().si_hip_memset_async(self.ptr, byte, self.nbytes, None), "memset")
        _chk(_native.hip().si_hip_stream_sync(None), "sync")

    def free(self):
        if getattr(self, "ptr", None):
            if not getattr(self, "_borrowed", False):
                _native.hip().si_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync():
    _chk(_native.hip().si_hip_device_sync(), "device sync")


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i4(shape: Sequence[int]):
    return (C.c_int * 4)(*[int(s) for s in shape])


def pad4(shape: Sequence[int]):
    shape = list(shape)
    if len(shape) >= 4:
        return [int(np.prod(shape[:len(shape) - 3]))] + shape[-3:]
    return [1] * (4 - len(shape)) + shape


def conv_out_hw(ih, iw, k, s, p, d):
    oh = (ih + 2 * p[0] - ((k[0] - 1) * d[0] + 1)) // s[0] + 1
    ow = (iw + 2 * p[1] - ((k[1] - 1) * d[1] + 1)) // s[1] + 1
    return oh, ow


def conv2d(x, w_oihw, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1, act1="none",
           residual=None, act2="none", act_param=0.0, in_ld: Optional[int] = None, out_ld: Optional[int] = None,
           out_c_off: int = 0, in_fill: float = 0.0):
    """si_hip_conv2d_f32.  in_ld/out_ld > C exercise the strided (concat-slice) addressing: the input is
    embedded in / the output is written into a wider zero-filled buffer and sliced back."""
    H = _native.hip()
    x, w_oihw = _f32(x), _f32(w_oihw)
    n, ih, iw, ic = x.shape
    oc, _, kh, kw = w_oihw.shape
    oh, ow = conv_out_hw(ih, iw, (kh, kw), stride, padding, dilation)
    in_ld = in_ld or ic
    out_ld = out_ld or oc
    d = SiConv2dDesc(n, ih, iw, ic, in_ld, oh, ow, oc, out_ld, kh, kw, stride[0], stride[1], dilation[0], dilation[1],
                     padding[0], padding[1], groups, 1 if bias is not None else 0, ACT[act1],
                     1 if residual is not None else 0, oc, ACT[act2], float(act_param))
    packed = np.zeros(H.si_hip_conv2d_weight_elems(C.byref(d)), np.float32)
    _chk(H.si_hip_conv2d_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)),
         "pack weight")
    if in_ld != ic:
        xw = np.full((n, ih, iw, in_ld), in_fill, np.float32)  # in_fill: what lies between the pixels' channels
        xw[..., :ic] = x
        x = xw
    dx, dw = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(packed)
    db = DeviceBuffer.from_numpy(_f32(bias)) if bias is not None else None
    dr = DeviceBuffer.from_numpy(_f32(residual)) if residual is not None else None
    dy = DeviceBuffer(n * oh * ow * out_ld * 4)
    dy.fill(0)
    _chk(H.si_hip_conv2d_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr if db else None, dr.ptr if dr else None,
                             dy.ptr + 4 * out_c_off, None), "si_hip_conv2d_f32")
    y = dy.to_numpy((n, oh, ow, out_ld))
    return y[..., out_c_off:out_c_off + oc].copy() if out_ld != oc else y


def _range_flag(d, want):
    """a zeroed device word handed to an f32_split launch as SiConv2dDesc::range_flag (the kernel writes 1 when an operand left fp16's range)"""
    if not want:
        return None
    f = DeviceBuffer.from_numpy(np.zeros(1, np.uint32))
    d.range_flag = f.ptr
    return f


def conv2d_split3(x, w_oihw, bias=None, stride=(1, 1), padding=(0, 0), act1="none", residual=None, act2="none", return_flag=False, in_ld=None):
    """si_hip_conv2d_split3_f32: fp32 conv on the fp16 matrix cores by operand splitting (three fp16 MFMAs per product, fp32 accumulate).
    return_flag: also return the range-guard word (1: an operand overflowed fp16 on its way through the split).  in_ld: the input as a channel
    slice of a wider tensor (pixel stride in_ld > ic; what lies between is NaN)"""
    H = _native.hip()
    x, w_oihw = _f32(x), _f32(w_oihw)
    n, ih, iw, ic = x.shape
    oc, _, kh, kw = w_oihw.shape
    oh, ow = conv_out_hw(ih, iw, (kh, kw), stride, padding, (1, 1))
    if in_ld and in_ld != ic:
        xw = np.full((n, ih, iw, in_ld), np.nan, np.float32)
        xw[..., :ic] = x
        x = xw
    d = SiConv2dDesc(n, ih, iw, ic, in_ld or ic, oh, ow, oc, oc, kh, kw, stride[0], stride[1], 1, 1, padding[0], padding[1], 1,
                     1 if bias is not None else 0, ACT[act1], 1 if residual is not None else 0, oc, ACT[act2], 0.0)
    if not H.si_hip_conv2d_split3_supported(C.byref(d)):
        raise HipError("si_hip_conv2d_split3_f32: unsupported shape")
    packed = np.zeros(H.si_hip_conv2d_split3_weight_elems(C.byref(d)), np.float16)
    _chk(H.si_hip_conv2d_split3_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack split3")
    dx, dw = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(packed)
    db = DeviceBuffer.from_numpy(_f32(bias)) if bias is not None else None
    dr = DeviceBuffer.from_numpy(_f32(residual)) if residual is not None else None
    dy = DeviceBuffer(n * oh * ow * oc * 4)
    flag = _range_flag(d, return_flag)
    _chk(H.si_hip_conv2d_split3_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr if db else None, dr.ptr if dr else None, dy.ptr, None), "si_hip_conv2d_split3_f32")
    y = dy.to_numpy((n, oh, ow, oc))
    return (y, int(flag.to_numpy((1,), np.uint32)[0])) if return_flag else y


def conv2d_stem_split3(x, w_oihw, bias=None, stride=(2, 2), padding=(2, 2), act1="none", return_flag=False):
    """si_hip_conv2d_stem_split3_f32: the RGB stem conv on the f32_split arithmetic (fp32 image in, fp32 activations out)"""
    H = _native.hip()
    x, w_oihw = _f32(x), _f32(w_oihw)
    n, ih, iw, ic = x.shape
    oc, _, kh, kw = w_oihw.shape
    oh, ow = conv_out_hw(ih, iw, (kh, kw), stride, padding, (1, 1))
    d = SiConv2dDesc(n, ih, iw, ic, ic, oh, ow, oc, oc, kh, kw, stride[0], stride[1], 1, 1, padding[0], padding[1], 1,
                     1 if bias is not None else 0, ACT[act1], 0, oc, ACT["none"], 0.0)
    if H.si_hip_conv2d_f16_supported(C.byref(d)) != 2:
        raise HipError("si_hip_conv2d_stem_split3_f32: not a stem shape")
    packed = np.zeros(H.si_hip_conv2d_stem_split3_weight_elems(C.byref(d)), np.float16)
    _chk(H.si_hip_conv2d_stem_split3_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack stem split3")
    dx, dw = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(packed)
    db = DeviceBuffer.from_numpy(_f32(bias)) if bias is not None else None
    dy = DeviceBuffer(n * oh * ow * oc * 4)
    flag = _range_flag(d, return_flag)
    _chk(H.si_hip_conv2d_stem_split3_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr if db else None, dy.ptr, None), "si_hip_conv2d_stem_split3_f32")
    y = dy.to_numpy((n, oh, ow, oc))
    return (y, int(flag.to_numpy((1,), np.uint32)[0])) if return_flag else y


def conv2d_wino23_split(x, w_oihw, bias=None, padding=(1, 1), act1="none", residual=None, act2="none", return_flag=False):
    """si_hip_conv2d_wino23_split_f32: fused Winograd F(2,3) with the plane GEMMs on the fp16 matrix cores by operand splitting"""
    H = _native.hip()
    x, w_oihw = _f32(x), _f32(w_oihw)
    n, ih, iw, ic = x.shape
    oc = w_oihw.shape[0]
    oh, ow = conv_out_hw(ih, iw, (3, 3), (1, 1), padding, (1, 1))
    d = SiConv2dDesc(n, ih, iw, ic, ic, oh, ow, oc, oc, 3, 3, 1, 1, 1, 1, padding[0], padding[1], 1,
                     1 if bias is not None else 0, ACT[act1], 1 if residual is not None else 0, oc, ACT[act2], 0.0)
    if not H.si_hip_conv2d_wino23_split_supported(C.byref(d)):
        raise HipError("si_hip_conv2d_wino23_split_f32: unsupported shape")
    packed = np.zeros(H.si_hip_conv2d_wino23_split_weight_elems(C.byref(d)), np.float16)
    _chk(H.si_hip_conv2d_wino23_split_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack wino split")
    dx, dw = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(packed)
    db = DeviceBuffer.from_numpy(_f32(bias)) if bias is not None else None
    dr = DeviceBuffer.from_numpy(_f32(residual)) if residual is not None else None
    dy = DeviceBuffer(n * oh * ow * oc * 4)
    flag = _range_flag(d, return_flag)
    _chk(H.si_hip_conv2d_wino23_split_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr if db else None, dr.ptr if dr else None, dy.ptr, None),
         "si_hip_conv2d_wino23_split_f32")
    y = dy.to_numpy((n, oh, ow, oc))
    return (y, int(flag.to_numpy((1,), np.uint32)[0])) if return_flag else y


def conv2d_upcat(low, skip, w_oihw, bias, scale=(2.0, 2.0), up_first=True, act1="none", split_oc=0, split3=False):
    """si_hip_conv2d_upcat_f32: a 1x1 conv over cat([upsample(low), skip]) (or [skip, upsample(low)]) that reads `low` at the
    source pixel.  Returns y, or (y, y2) for the sibling-split form.  split3: the same on the f32_split arithmetic (si_hip_conv2d_split3_upcat_f32)."""
    H = _native.hip()
    low, skip, w_oihw = _f32(low), _f32(skip), _f32(w_oihw)
    n, oh, ow, cs = skip.shape
    _, lh, lw, cl = low.shape
    ic, oc = cl + cs, w_oihw.shape[0]
    assert w_oihw.shape[1] == ic and w_oihw.shape[2:] == (1, 1)
    d = SiConv2dDesc(n, oh, ow, ic, ic, oh, ow, oc, oc, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1 if bias is not None else 0, ACT[act1], 0, oc, 0, 0.0)
    if split3:
        wp = np.zeros(H.si_hip_conv2d_split3_weight_elems(C.byref(d)), np.float16)
        _chk(H.si_hip_conv2d_split3_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p)), "pack split3")
    else:
        wn = H.si_hip_conv2d_weight_elems(C.byref(d))
        wp = np.empty(wn, np.float32)
        _chk(H.si_hip_conv2d_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p)), "pack")
    fn, fname = (H.si_hip_conv2d_split3_upcat_f32, "si_hip_conv2d_split3_upcat_f32") if split3 else (H.si_hip_conv2d_upcat_f32, "si_hip_conv2d_upcat_f32")
    # the concat buffer: only the skip channels are ever written; the upsampled range is poisoned to prove nobody reads it
    cat = np.full((n, oh, ow, ic), np.nan, np.float32)
    c0 = 0 if up_first else cs
    cat[..., (cl if up_first else 0):(cl if up_first else 0) + cs] = skip
    dcat, dlow, dw = DeviceBuffer.from_numpy(cat), DeviceBuffer.from_numpy(low), DeviceBuffer.from_numpy(wp)
    db = DeviceBuffer.from_numpy(_f32(bias)) if bias is not None else None
    up = _native.SiConv2dUpsampledSource(dlow.ptr, lh, lw, cl, cl, c0, np.float32(1.0) / np.float32(scale[0]), np.float32(1.0) / np.float32(scale[1]))
    if split_oc:
        d.out_ld = split_oc
        dy, dy2 = DeviceBuffer(n * oh * ow * split_oc * 4), DeviceBuffer(n * oh * ow * (oc - split_oc) * 4)
        _chk(fn(C.byref(d), dcat.ptr, C.byref(up), dw.ptr, db.ptr if db else None, dy.ptr, split_oc, dy2.ptr, oc - split_oc, None), fname)
        return dy.to_numpy((n, oh, ow, split_oc)), dy2.to_numpy((n, oh, ow, oc - split_oc))
    dy = DeviceBuffer(n * oh * ow * oc * 4)
    _chk(fn(C.byref(d), dcat.ptr, C.byref(up), dw.ptr, db.ptr if db else None, dy.ptr, 0, None, 0, None), fname)
    return dy.to_numpy((n, oh, ow, oc))


def conv2d_upcat_f16(low, skip, w_oihw, bias, scale=(2.0, 2.0), up_first=True, act1="none", split_oc=0):
    """si_hip_conv2d_upcat_f16: conv2d_upcat with fp16 storage (half tensors in and out, fp32 bias)."""
    H = _native.hip()
    low, skip = np.asarray(low, np.float16), np.asarray(skip, np.float16)
    w_oihw = _f32(w_oihw)
    n, oh, ow, cs = skip.shape
    _, lh, lw, cl = low.shape
    ic, oc = cl + cs, w_oihw.shape[0]
    d = SiConv2dDesc(n, oh, ow, ic, ic, oh, ow, oc, oc, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1 if bias is not None else 0, ACT[act1], 0, oc, 0, 0.0)
    wp = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d)), np.float16)
    _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p)), "pack f16")
    cat = np.full((n, oh, ow, ic), np.nan, np.float16)   # the upsampled range is poison: nobody may read it
    c0 = 0 if up_first else cs
    cat[..., (cl if up_first else 0):(cl if up_first else 0) + cs] = skip
    dcat, dlow, dw = DeviceBuffer.from_numpy(cat), DeviceBuffer.from_numpy(np.ascontiguousarray(low)), DeviceBuffer.from_numpy(wp)
    db = DeviceBuffer.from_numpy(_f32(bias)) if bias is not None else None
    up = _native.SiConv2dUpsampledSource(dlow.ptr, lh, lw, cl, cl, c0, np.float32(1.0) / np.float32(scale[0]), np.float32(1.0) / np.float32(scale[1]))
    if split_oc:
        d.out_ld = split_oc
        dy, dy2 = DeviceBuffer(n * oh * ow * split_oc * 2), DeviceBuffer(n * oh * ow * (oc - split_oc) * 2)
        _chk(H.si_hip_conv2d_upcat_f16(C.byref(d), dcat.ptr, C.byref(up), dw.ptr, db.ptr if db else None, dy.ptr, split_oc, dy2.ptr,
                                       oc - split_oc, None), "si_hip_conv2d_upcat_f16")
        return dy.to_numpy((n, oh, ow, split_oc), np.float16), dy2.to_numpy((n, oh, ow, oc - split_oc), np.float16)
    dy = DeviceBuffer(n * oh * ow * oc * 2)
    _chk(H.si_hip_conv2d_upcat_f16(C.byref(d), dcat.ptr, C.byref(up), dw.ptr, db.ptr if db else None, dy.ptr, 0, None, 0, None),
         "si_hip_conv2d_upcat_f16")
    return dy.to_numpy((n, oh, ow, oc), np.float16)


def conv2d_kernel_name(x_shape, w_shape, stride=(1, 1), padding=(0, 0), groups=1) -> str:
    """The kernel instantiation si_hip_conv2d_f32 picks for this shape with dense, 16-byte aligned tensors."""
    H = _native.hip()
    n, ih, iw, ic = x_shape
    oc, _, kh, kw = w_shape
    oh, ow = conv_out_hw(ih, iw, (kh, kw), stride, padding, (1, 1))
    d = SiConv2dDesc(n, ih, iw, ic, ic, oh, ow, oc, oc, kh, kw, stride[0], stride[1], 1, 1, padding[0], padding[1], groups, 1,
                     ACT["none"], 0, oc, ACT["none"], 0.0)
    return H.si_hip_conv2d_kernel_name(C.byref(d), C.c_void_p(4096)).decode()


def conv2d_winograd(x, w_oihw, bias=None, padding=(1, 1), act1="none", residual=None, act2="none", in_ld=None,
                    out_ld=None, out_c_off=0, tile=2):
    """si_hip_conv2d_wino23_f32 (tile=2, fused Winograd F(2,3)) or si_hip_conv2d_wino43_f32 (tile=4, F(4,3)); raises
    HipError for ineligible shapes."""
    H = _native.hip()
    fam = "wino23" if tile == 2 else "wino43"
    f_elig, f_elems = getattr(H, "si_hip_conv2d_%s_eligible" % fam), getattr(H, "si_hip_conv2d_%s_weight_elems" % fam)
    f_pack, f_run = getattr(H, "si_hip_conv2d_%s_pack_weight_host" % fam), getattr(H, "si_hip_conv2d_%s_f32" % fam)
    x, w_oihw = _f32(x), _f32(w_oihw)
    n, ih, iw, ic = x.shape
    oc = w_oihw.shape[0]
    oh, ow = ih + 2 * padding[0] - 2, iw + 2 * padding[1] - 2
    in_ld = in_ld or ic
    out_ld = out_ld or oc
    d = SiConv2dDesc(n, ih, iw, ic, in_ld, oh, ow, oc, out_ld, 3, 3, 1, 1, 1, 1, padding[0], padding[1], 1,
                     1 if bias is not None else 0, ACT[act1], 1 if residual is not None else 0, oc, ACT[act2], 0.0)
    if not f_elig(C.byref(d)):
        raise HipError("shape not eligible for Winograd")
    u = np.zeros(f_elems(C.byref(d)), np.float32)
    _chk(f_pack(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p)), "wino pack")
    if in_ld != ic:
        xw = np.zeros((n, ih, iw, in_ld), np.float32)
        xw[..., :ic] = x
        x = xw
    dx, du = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(u)
    db = DeviceBuffer.from_numpy(_f32(bias)) if bias is not None else None
    dr = DeviceBuffer.from_numpy(_f32(residual)) if residual is not None else None
    dy = DeviceBuffer(n * oh * ow * out_ld * 4)
    dy.fill(0)
    _chk(f_run(C.byref(d), dx.ptr, du.ptr, db.ptr if db else None, dr.ptr if dr else None,
               dy.ptr + 4 * out_c_off, None), "si_hip_conv2d_%s_f32" % fam)
    y = dy.to_numpy((n, oh, ow, out_ld))
    return y[..., out_c_off:out_c_off + oc].copy() if out_ld != oc else y


def conv2d_split(x, w_a, b_a, w_b, b_b, act1="none", out2_ld=None, out2_c_off=0):
    """si_hip_conv2d_split_f32: two 1x1 convs on the same input in one launch; returns (y_a, y_b)."""
    H = _native.hip()
    x, w_a, w_b = _f32(x), _f32(w_a), _f32(w_b)
    n, ih, iw, ic = x.shape
    oa, ob = w_a.shape[0], w_b.shape[0]
    out2_ld = out2_ld or ob

    def packed(w):
        d = SiConv2dDesc(n, ih, iw, ic, ic, ih, iw, w.shape[0], w.shape[0], 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0.0)
        buf = np.zeros(H.si_hip_conv2d_weight_elems(C.byref(d)), np.float32)
        _chk(H.si_hip_conv2d_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p)), "pack")
        return buf

    wcat = np.concatenate([packed(w_a), packed(w_b)])
    bcat = np.concatenate([_f32(b_a), _f32(b_b)])
    d = SiConv2dDesc(n, ih, iw, ic, ic, ih, iw, oa + ob, oa, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, ACT[act1], 0, 0, 0, 0.0)
    dx, dw, db = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(wcat), DeviceBuffer.from_numpy(bcat)
    dya, dyb = DeviceBuffer(n * ih * iw * oa * 4), DeviceBuffer(n * ih * iw * out2_ld * 4)
    dyb.fill(0)
    _chk(H.si_hip_conv2d_split_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr, dya.ptr, oa, dyb.ptr + 4 * out2_c_off, out2_ld, None),
         "si_hip_conv2d_split_f32")
    yb = dyb.to_numpy((n, ih, iw, out2_ld))
    return dya.to_numpy((n, ih, iw, oa)), yb[..., out2_c_off:out2_c_off + ob].copy()


def conv2d_split3_split(x, w_a, b_a, w_b, b_b, act1="none", out2_ld=None, out2_c_off=0):
    """si_hip_conv2d_split3_split_f32: two sibling 1x1 convs as one launch on the f32_split arithmetic; returns (y_a, y_b)"""
    H = _native.hip()
    x, w_a, w_b = _f32(x), _f32(w_a), _f32(w_b)
    n, ih, iw, ic = x.shape
    oa, ob = w_a.shape[0], w_b.shape[0]
    out2_ld = out2_ld or ob
    wcat = _f32(np.concatenate([w_a, w_b], 0))
    bcat = np.concatenate([_f32(b_a), _f32(b_b)])
    d = SiConv2dDesc(n, ih, iw, ic, ic, ih, iw, oa + ob, oa, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, ACT[act1], 0, 0, 0, 0.0)
    packed = np.zeros(H.si_hip_conv2d_split3_weight_elems(C.byref(d)), np.float16)
    _chk(H.si_hip_conv2d_split3_pack_weight_host(C.byref(d), wcat.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack split3")
    dx, dw, db = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(packed), DeviceBuffer.from_numpy(bcat)
    dya, dyb = DeviceBuffer(n * ih * iw * oa * 4), DeviceBuffer(n * ih * iw * out2_ld * 4)
    dyb.fill(0)
    _chk(H.si_hip_conv2d_split3_split_f32(C.byref(d), dx.ptr, dw.ptr, db.ptr, dya.ptr, oa, dyb.ptr + 4 * out2_c_off, out2_ld, None),
         "si_hip_conv2d_split3_split_f32")
    yb = dyb.to_numpy((n, ih, iw, out2_ld))
    return dya.to_numpy((n, ih, iw, oa)), yb[..., out2_c_off:out2_c_off + ob].copy()


def linear(x, w, b=None):
    H = _native.hip()
    x, w = _f32(x), _f32(w)
    dx, dw = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(w)
    db = DeviceBuffer.from_numpy(_f32(b)) if b is not None else None
    dy = DeviceBuffer(x.shape[0] * w.shape[0] * 4)
    _chk(H.si_hip_linear_f32(dx.ptr, x.shape[0], x.shape[1], dw.ptr, db.ptr if db else None, w.shape[0], dy.ptr, None),
         "si_hip_linear_f32")
    return dy.to_numpy((x.shape[0], w.shape[0]))


def maxpool2d(x, k, s, p, d=(1, 1)):
    H = _native.hip()
    x = _f32(x)
    n, ih, iw, c = x.shape
    oh, ow = conv_out_hw(ih, iw, k, s, p, d)
    desc = SiPool2dDesc(n, ih, iw, c, c, oh, ow, c, k[0], k[1], s[0], s[1], d[0], d[1], p[0], p[1])
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(n * oh * ow * c * 4)
    _chk(H.si_hip_maxpool2d_f32(C.byref(desc), dx.ptr, dy.ptr, None), "si_hip_maxpool2d_f32")
    return dy.to_numpy((n, oh, ow, c))


def maxpool5_chain3(x, half=False, out_ld=None, out_c_off=(0, 0, 0)):
    """si_hip_maxpool5_chain3_{f32,f16}: the three chained 5x5 s1 p2 pools of SPPF; the outputs may be channel slices of
    wider rows (out_ld elements per pixel, starting at out_c_off[k]), as the concat aliasing hands them over."""
    H = _native.hip()
    x = _f16(x) if half else _f32(x)
    n, h, w, c = x.shape
    esz = 2 if half else 4
    ld = out_ld or c
    dx = DeviceBuffer.from_numpy(x)
    outs = [DeviceBuffer(n * h * w * ld * esz) for _ in range(3)]
    for o in outs:
        o.fill(0)
    fn = H.si_hip_maxpool5_chain3_f16 if half else H.si_hip_maxpool5_chain3_f32
    _chk(fn(dx.ptr, n, h, w, c, c, outs[0].ptr + esz * out_c_off[0], ld, outs[1].ptr + esz * out_c_off[1], ld,
            outs[2].ptr + esz * out_c_off[2], ld, None), "si_hip_maxpool5_chain3")
    dt = np.float16 if half else np.float32
    return [o.to_numpy((n, h, w, ld), dt)[..., off:off + c].copy() for o, off in zip(outs, out_c_off)]


def adaptive_avgpool2d(x, out_hw):
    H = _native.hip()
    x = _f32(x)
    n, ih, iw, c = x.shape
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(n * out_hw[0] * out_hw[1] * c * 4)
    _chk(H.si_hip_adaptive_avgpool2d_f32(dx.ptr, n, ih, iw, c, c, dy.ptr, out_hw[0], out_hw[1], c, None),
         "si_hip_adaptive_avgpool2d_f32")
    return dy.to_numpy((n, out_hw[0], out_hw[1], c))


def upsample_nearest(x, scale_h, scale_w, out_hw=None):
    H = _native.hip()
    x = _f32(x)
    n, ih, iw, c = x.shape
    oh, ow = out_hw if out_hw else (int(ih * scale_h), int(iw * scale_w))
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(n * oh * ow * c * 4)
    _chk(H.si_hip_upsample_nearest_f32(dx.ptr, n, ih, iw, c, c, scale_h, scale_w, dy.ptr, oh, ow, c, None),
         "si_hip_upsample_nearest_f32")
    return dy.to_numpy((n, oh, ow, c))


def cat(xs, axis):
    H = _native.hip()
    xs = [_f32(x) for x in xs]
    shp = list(xs[0].shape)
    shp[axis] = sum(x.shape[axis] for x in xs)
    dy = DeviceBuffer(int(np.prod(shp)) * 4)
    off = 0
    for x in xs:
        dx = DeviceBuffer.from_numpy(x)
        if axis == 3:
            _chk(H.si_hip_copy_channels_f32(dx.ptr, x.size // x.shape[3], x.shape[3], x.shape[3], dy.ptr + 4 * off,
                                            shp[3], None), "si_hip_copy_channels_f32")
        else:
            _chk(H.si_hip_cat_axis_f32(dx.ptr, _i4(x.shape), dy.ptr, _i4(shp), axis, off, None), "si_hip_cat_axis_f32")
        off += x.shape[axis]
        sync()
    return dy.to_numpy(shp)


def binary_op(op, a, b, out_shape=None):
    H = _native.hip()
    a, b = _f32(a), _f32(b)
    a4, b4 = pad4(a.shape), pad4(b.shape)
    o4 = pad4(out_shape) if out_shape is not None else [max(x, y) for x, y in zip(a4, b4)]
    da, db_, dy = DeviceBuffer.from_numpy(a), DeviceBuffer.from_numpy(b), DeviceBuffer(int(np.prod(o4)) * 4)
    _chk(H.si_hip_binary_f32(op, da.ptr, _i4(a4), a4[3], db_.ptr, _i4(b4), b4[3], dy.ptr, _i4(o4), o4[3], None),
         "si_hip_binary_f32")
    return dy.to_numpy(o4).reshape(out_shape if out_shape is not None else o4)


def binary_scalar(op, x, scalar):
    """out = x (op) scalar -- BinaryOp's with_scalar form (op codes of include/si_hip.h; 7 / 8 / 9 / 11 put the scalar first)."""
    H = _native.hip()
    x = _f32(x)
    c = x.shape[-1]
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(x.nbytes)
    _chk(H.si_hip_binary_scalar_f32(op, dx.ptr, x.size // c, c, c, float(scalar), dy.ptr, c, None), "si_hip_binary_scalar_f32")
    return dy.to_numpy(x.shape)


def unary_op(op, x, in_ld=None, out_ld=None):
    """out = f(x) -- UnaryOp codes 0..17 (include/si_hip.h); optional pixel strides exercise the strided form."""
    H = _native.hip()
    x = _f32(x)
    c = x.shape[-1]
    pixels = x.size // c
    ild, old = in_ld or c, out_ld or c
    xin = np.full((pixels, ild), np.nan, np.float32)
    xin[:, :c] = x.reshape(pixels, c)
    dx, dy = DeviceBuffer.from_numpy(xin), DeviceBuffer(pixels * old * 4)
    dy.fill(0)
    _chk(H.si_hip_unary_f32(op, dx.ptr, pixels, c, ild, dy.ptr, old, None), "si_hip_unary_f32")
    return dy.to_numpy((pixels, old))[:, :c].reshape(x.shape)


def unary_op_f16(op, x, in_ld=None, out_ld=None):
    """si_hip_unary_f16: UnaryOp on fp16 tensors (the fp32 function on the widened value, one rounding)."""
    H = _native.hip()
    x = _f16(x)
    c = x.shape[-1]
    pixels = x.size // c
    ild, old = in_ld or c, out_ld or c
    xin = np.full((pixels, ild), np.nan, np.float16)
    xin[:, :c] = x.reshape(pixels, c)
    dx, dy = DeviceBuffer.from_numpy(xin), DeviceBuffer(pixels * old * 2)
    dy.fill(0)
    _chk(H.si_hip_unary_f16(op, dx.ptr, pixels, c, ild, dy.ptr, old, None), "si_hip_unary_f16")
    return dy.to_numpy((pixels, old), np.float16)[:, :c].reshape(x.shape)


def activation(kind, x, param=0.0):
    H = _native.hip()
    x = _f32(x)
    c = x.shape[-1]
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(x.nbytes)
    _chk(H.si_hip_activation_f32(ACT[kind], param, dx.ptr, x.size // c, c, c, dy.ptr, c, None), "si_hip_activation_f32")
    return dy.to_numpy(x.shape)


def batchnorm2d(x, mean, var, gamma, beta, eps):
    H = _native.hip()
    x = _f32(x)
    c = x.shape[-1]
    bufs = [DeviceBuffer.from_numpy(_f32(v)) for v in (mean, var, gamma, beta)]
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(x.nbytes)
    _chk(H.si_hip_batchnorm2d_f32(dx.ptr, x.size // c, c, c, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, eps,
                                  dy.ptr, c, None), "si_hip_batchnorm2d_f32")
    return dy.to_numpy(x.shape)


def flatten_nhwc(x):
    H = _native.hip()
    x = _f32(x)
    n, h, w, c = x.shape
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(x.nbytes)
    _chk(H.si_hip_nhwc_to_nchw_f32(dx.ptr, n, h, w, c, c, dy.ptr, None), "si_hip_nhwc_to_nchw_f32")
    return dy.to_numpy((n, c * h * w))


def yolo_detect(feats, weights, biases, grids, anchor_grids, strides, na=3, fused=False):
    """Detect head exactly as the YoloDetect layer runs it: per level 1x1 conv kernel + decode kernel, or (fused)
    si_hip_conv2d_yolo_f32 -- the conv with the decode + concat in its epilogue."""
    H = _native.hip()
    if fused:
        return _yolo_detect_fused(feats, weights, biases, grids, anchor_grids, strides, na)
    feats = [_f32(f) for f in feats]
    n = feats[0].shape[0]
    ne = weights[0].shape[0] // na
    rows_total = sum(f.shape[1] * f.shape[2] * na for f in feats)
    dout = DeviceBuffer(n * rows_total * ne * 4)
    off = 0
    for f, w, b, g, a, s in zip(feats, weights, biases, grids, anchor_grids, strides):
        _, h, wd, cin = f.shape
        conv = conv2d(f, w, b)  # [n,h,w,na*ne]
        g2 = _f32(np.transpose(_f32(g)[0], (1, 2, 0, 3)))   # [na,h,w,2] -> [h,w,na,2] (yolo_detect.cpp:75-79)
        a2 = _f32(np.transpose(_f32(a)[0], (1, 2, 0, 3)))
        dc, dg, da = DeviceBuffer.from_numpy(conv), DeviceBuffer.from_numpy(g2), DeviceBuffer.from_numpy(a2)
        _chk(H.si_hip_yolo_decode_f32(dc.ptr, n, h, wd, na, ne, dg.ptr, da.ptr, float(s), dout.ptr, rows_total, off, None),
             "si_hip_yolo_decode_f32")
        sync()
        off += h * wd * na
    return dout.to_numpy((n, rows_total, ne))


def yolo_detect_split3(feats, weights, biases, grids, anchor_grids, strides, na=3, return_flags=False):
    """si_hip_conv2d_split3_yolo_f32 per level: the Detect head on the f32_split arithmetic (fp32 features, three fp16 MFMA products per
    fp32 product), decode + concat in the epilogue.  return_flags: also the per-level range-guard words."""
    return _yolo_detect_fused(feats, weights, biases, grids, anchor_grids, strides, na, split3=True, return_flags=return_flags)


def _yolo_detect_fused(feats, weights, biases, grids, anchor_grids, strides, na, split3=False, return_flags=False):
    from ._native import SiYoloLevel
    H = _native.hip()
    feats = [_f32(f) for f in feats]
    n = feats[0].shape[0]
    ne = weights[0].shape[0] // na
    rows_total = sum(f.shape[1] * f.shape[2] * na for f in feats)
    dout = DeviceBuffer(n * rows_total * ne * 4)
    dout.fill(0)
    off = 0
    flags = []
    for f, w, b, g, a, s in zip(feats, weights, biases, grids, anchor_grids, strides):
        _, h, wd, cin = f.shape
        w = _f32(w)
        d = SiConv2dDesc(n, h, wd, cin, cin, h, wd, na * ne, na * ne, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, na * ne, 0, 0.0)
        if split3:
            packed = np.zeros(H.si_hip_conv2d_split3_weight_elems(C.byref(d)), np.float16)
            _chk(H.si_hip_conv2d_split3_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack")
        else:
            packed = np.zeros(H.si_hip_conv2d_weight_elems(C.byref(d)), np.float32)
            _chk(H.si_hip_conv2d_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack")
        g2 = _f32(np.transpose(_f32(g)[0], (1, 2, 0, 3)))
        a2 = _f32(np.transpose(_f32(a)[0], (1, 2, 0, 3)))
        bufs = [DeviceBuffer.from_numpy(v) for v in (f, packed, _f32(b), g2, a2)]
        lv = SiYoloLevel(na, ne, rows_total, off, float(s))
        fn = H.si_hip_conv2d_split3_yolo_f32 if split3 else H.si_hip_conv2d_yolo_f32
        flag = _range_flag(d, return_flags and split3)
        _chk(fn(C.byref(d), bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, C.byref(lv), bufs[3].ptr, bufs[4].ptr, dout.ptr, None),
             "si_hip_conv2d_split3_yolo_f32" if split3 else "si_hip_conv2d_yolo_f32")
        sync()
        if flag is not None:
            flags.append(int(flag.to_numpy((1,), np.uint32)[0]))
        off += h * wd * na
    out = dout.to_numpy((n, rows_total, ne))
    return (out, flags) if return_flags else out


def letterbox_geometry(height_origin, width_origin, height_new, width_new):
    """(height_resize, width_resize, scale, padding_t, padding_l) of PreProcess (test_yolo.cpp:194-241)."""
    hr, wr, pt, pl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    sc = C.c_float()
    _native.hip().si_letterbox_geometry(height_origin, width_origin, height_new, width_new, C.byref(hr), C.byref(wr),
                                        C.byref(sc), C.byref(pt), C.byref(pl))
    return hr.value, wr.value, sc.value, pt.value, pl.value


def letterbox(resized_bgr, height_new, width_new, padding_t, padding_l):
    """u8 BGR [hr][wr][3] -> float RGB [height_new][width_new][3], padded with 114, / 255 (test_yolo.cpp:220-259)."""
    H = _native.hip()
    src = np.ascontiguousarray(resized_bgr, dtype=np.uint8)
    hr, wr = int(src.shape[0]), int(src.shape[1])
    dsrc = DeviceBuffer.from_numpy(src)
    dout = DeviceBuffer(height_new * width_new * 3 * 4)
    _chk(H.si_hip_letterbox_u8_f32(dsrc.ptr, hr, wr, dout.ptr, height_new, width_new, padding_t, padding_l, None),
         "si_hip_letterbox_u8_f32")
    return dout.to_numpy((height_new, width_new, 3))


def letterbox_batch(resized_bgr, height_new, width_new, padding_t, padding_l):
    """si_hip_letterbox_batch_u8_f32: u8 BGR [n][hr][wr][3] -> float RGB [n][height_new][width_new][3] in one launch."""
    H = _native.hip()
    src = np.ascontiguousarray(resized_bgr, dtype=np.uint8)
    n, hr, wr = int(src.shape[0]), int(src.shape[1]), int(src.shape[2])
    dsrc = DeviceBuffer.from_numpy(src)
    dout = DeviceBuffer(max(n * height_new * width_new * 3 * 4, 16))
    _chk(H.si_hip_letterbox_batch_u8_f32(dsrc.ptr, n, hr * wr * 3, hr, wr, dout.ptr, height_new, width_new, padding_t, padding_l, None),
         "si_hip_letterbox_batch_u8_f32")
    return dout.to_numpy((n, height_new, width_new, 3))


def resize_bilinear_u8c3(images, dst_h, dst_w):
    """si_hip_resize_bilinear_u8c3: u8 [n][h][w][3] -> u8 [n][dst_h][dst_w][3] (the cv::resize of PreProcess, test_yolo.cpp:213-216)."""
    H = _native.hip()
    src = np.ascontiguousarray(images, dtype=np.uint8)
    n, h, w = int(src.shape[0]), int(src.shape[1]), int(src.shape[2])
    dsrc = DeviceBuffer.from_numpy(src)
    dout = DeviceBuffer(max(n * dst_h * dst_w * 3, 16))
    _chk(H.si_hip_resize_bilinear_u8c3(dsrc.ptr, n, h * w * 3, h, w, dout.ptr, dst_h * dst_w * 3, dst_h, dst_w, None), "si_hip_resize_bilinear_u8c3")
    return dout.to_numpy((n, dst_h, dst_w, 3), np.uint8)


def resize_letterbox_batch(frames_bgr, height_new, width_new):
    """si_hip_resize_letterbox_batch_u8_f32: camera frames u8 BGR [n][h][w][3] -> float RGB [n][height_new][width_new][3]
    (aspect-preserving bilinear resize + pad(114) + / 255: PreProcess whole, test_yolo.cpp:194-259) in one launch."""
    H = _native.hip()
    src = np.ascontiguousarray(frames_bgr, dtype=np.uint8)
    n, h, w = int(src.shape[0]), int(src.shape[1]), int(src.shape[2])
    dsrc = DeviceBuffer.from_numpy(src)
    dout = DeviceBuffer(max(n * height_new * width_new * 3 * 4, 16))
    _chk(H.si_hip_resize_letterbox_batch_u8_f32(dsrc.ptr, n, h * w * 3, h, w, dout.ptr, height_new, width_new, None),
         "si_hip_resize_letterbox_batch_u8_f32")
    return dout.to_numpy((n, height_new, width_new, 3))


def yolo_postprocess(pred, prob_threshold=0.25, nms_threshold=0.45, agnostic=False, adjust=None, max_det=None,
                     pred_dev=None):
    """Device post-processing of test_yolo.cpp:337-428.  pred [n][rows][ne] -> list (one per image) of float arrays
    [k][6] = {x, y, w, h, confidence, label} in picked order.  adjust: None or [n][5]
    {padding_l, padding_t, scale, image cols, image rows}."""
    H = _native.hip()
    pred = _f32(pred)
    n, rows, ne = (int(v) for v in pred.shape)
    if max_det is None:
        max_det = max(rows, 1)
    dpred = pred_dev if pred_dev is not None else DeviceBuffer.from_numpy(pred)
    dadj = DeviceBuffer.from_numpy(_f32(adjust).reshape(n, 5)) if adjust is not None else None
    wsb = H.si_hip_yolo_postprocess_workspace_bytes(n, rows, ne)
    dws = DeviceBuffer(wsb)
    ddets = DeviceBuffer(max(n * max_det * 6 * 4, 16))
    dcnt = DeviceBuffer(max(n * 4, 16))
    _chk(H.si_hip_yolo_postprocess_f32(dpred.ptr, n, rows, ne, float(prob_threshold), float(nms_threshold),
                                       int(bool(agnostic)), dadj.ptr if dadj else None, ddets.ptr, dcnt.ptr, max_det,
                                       dws.ptr, wsb, None), "si_hip_yolo_postprocess_f32")
    if n == 0:
        return [], np.zeros((0,), np.int32)
    cnt = dcnt.to_numpy((n,), np.int32)
    dets = ddets.to_numpy((n, max_det, 6)) if max_det > 0 else np.zeros((n, 0, 6), np.float32)
    return [dets[b, :min(int(cnt[b]), max_det)].copy() for b in range(n)], cnt


# ---------------------------------------------------------------------------
# fp16 storage path (include/si_hip.h "fp16 storage path").  Activations travel as numpy float16 (IEEE binary16, the
# device's _Float16); results come back as float16 unless noted.
# ---------------------------------------------------------------------------
def _f16(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float16)


def conv2d_f16(x, w_oihw, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1, act1="none",
               residual=None, act2="none", act_param=0.0, in_ld=None, out_ld=None, out_c_off=0, out_f32=False, in_fill=0.0):
    """si_hip_conv2d_f16, si_hip_conv2d_stem_f16 when the shape is a stem (fp32 image in, fp16 out), or si_hip_conv2d_depthwise_f16.
    in_fill: what lies between the pixels' channels of a strided (in_ld > ic) input."""
    H = _native.hip()
    w_oihw = _f32(w_oihw)
    n, ih, iw, ic = x.shape
    oc, _, kh, kw = w_oihw.shape
    oh, ow = conv_out_hw(ih, iw, (kh, kw), stride, padding, dilation)
    in_ld = in_ld or ic
    out_ld = out_ld or oc
    d = SiConv2dDesc(n, ih, iw, ic, in_ld, oh, ow, oc, out_ld, kh, kw, stride[0], stride[1], dilation[0], dilation[1],
                     padding[0], padding[1], groups, 1 if bias is not None else 0, ACT[act1],
                     1 if residual is not None else 0, oc, ACT[act2], float(act_param))
    kind = H.si_hip_conv2d_f16_supported(C.byref(d))
    if kind == 0:
        raise HipError("no fp16 conv kernel for this shape")
    db = DeviceBuffer.from_numpy(_f32(bias)) if bias is not None else None
    osz = 4 if out_f32 else 2
    dy = DeviceBuffer(n * oh * ow * out_ld * osz)
    dy.fill(0)
    if kind == 2:
        packed = np.zeros(H.si_hip_conv2d_stem_f16_weight_elems(C.byref(d)), np.float16)
        _chk(H.si_hip_conv2d_stem_f16_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p),
                                                       packed.ctypes.data_as(C.c_void_p)), "pack stem f16")
        xs = _f32(x)
        if in_ld != ic:
            xs = np.zeros((n, ih, iw, in_ld), np.float32)
            xs[..., :ic] = x
        dx, dw = DeviceBuffer.from_numpy(xs), DeviceBuffer.from_numpy(packed)
        _chk(H.si_hip_conv2d_stem_f16(C.byref(d), dx.ptr, dw.ptr, db.ptr if db else None, dy.ptr + 2 * out_c_off, None),
             "si_hip_conv2d_stem_f16")
        y = dy.to_numpy((n, oh, ow, out_ld), np.float16)
        return y[..., out_c_off:out_c_off + oc].copy() if out_ld != oc else y
    if kind == 3:   # depthwise: fp16 activations, the FP32 depthwise weight image
        packed = np.zeros(H.si_hip_conv2d_weight_elems(C.byref(d)), np.float32)
        _chk(H.si_hip_conv2d_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack depthwise")
        if out_f32:
            raise HipError("the fp16 depthwise kernel writes fp16")
        xh = _f16(x)
        if in_ld != ic:
            xw = np.zeros((n, ih, iw, in_ld), np.float16)
            xw[..., :ic] = xh
            xh = xw
        dx, dw = DeviceBuffer.from_numpy(xh), DeviceBuffer.from_numpy(packed)
        dr = DeviceBuffer.from_numpy(_f16(residual)) if residual is not None else None
        _chk(H.si_hip_conv2d_depthwise_f16(C.byref(d), dx.ptr, dw.ptr, db.ptr if db else None, dr.ptr if dr else None,
                                           dy.ptr + 2 * out_c_off, None), "si_hip_conv2d_depthwise_f16")
        y = dy.to_numpy((n, oh, ow, out_ld), np.float16)
        return y[..., out_c_off:out_c_off + oc].copy() if out_ld != oc else y
    packed = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d)), np.float16)
    _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), w_oihw.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack f16")
    x = _f16(x)
    if in_ld != ic:
        xw = np.full((n, ih, iw, in_ld), in_fill, np.float16)
        xw[..., :ic] = x
        x = xw
    dx, dw = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(packed)
    dr = DeviceBuffer.from_numpy(_f16(residual)) if residual is not None else None
    _chk(H.si_hip_conv2d_f16(C.byref(d), dx.ptr, dw.ptr, db.ptr if db else None, dr.ptr if dr else None,
                             dy.ptr + osz * out_c_off, 1 if out_f32 else 0, None), "si_hip_conv2d_f16")
    y = dy.to_numpy((n, oh, ow, out_ld), np.float32 if out_f32 else np.float16)
    return y[..., out_c_off:out_c_off + oc].copy() if out_ld != oc else y


def conv2d_split_f16(x, w_a, b_a, w_b, b_b, act1="none"):
    H = _native.hip()
    x = _f16(x)
    n, h, w, ic = x.shape
    oa, ob = w_a.shape[0], w_b.shape[0]
    wcat = _f32(np.concatenate([w_a, w_b], 0))
    bcat = _f32(np.concatenate([b_a, b_b], 0))
    d = SiConv2dDesc(n, h, w, ic, ic, h, w, oa + ob, oa, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, ACT[act1], 0, oa, 0, 0.0)
    packed = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d)), np.float16)
    _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), wcat.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack f16")
    dx, dw, db = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(packed), DeviceBuffer.from_numpy(bcat)
    dy, dy2 = DeviceBuffer(n * h * w * oa * 2), DeviceBuffer(n * h * w * ob * 2)
    _chk(H.si_hip_conv2d_split_f16(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, oa, dy2.ptr, ob, None), "si_hip_conv2d_split_f16")
    return dy.to_numpy((n, h, w, oa), np.float16), dy2.to_numpy((n, h, w, ob), np.float16)


def maxpool2d_f16(x, k, s, p, d=(1, 1)):
    H = _native.hip()
    x = _f16(x)
    n, ih, iw, c = x.shape
    oh, ow = conv_out_hw(ih, iw, k, s, p, d)
    desc = SiPool2dDesc(n, ih, iw, c, c, oh, ow, c, k[0], k[1], s[0], s[1], d[0], d[1], p[0], p[1])
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(n * oh * ow * c * 2)
    _chk(H.si_hip_maxpool2d_f16(C.byref(desc), dx.ptr, dy.ptr, None), "si_hip_maxpool2d_f16")
    return dy.to_numpy((n, oh, ow, c), np.float16)


def adaptive_avgpool2d_f16(x, out_hw):
    H = _native.hip()
    x = _f16(x)
    n, ih, iw, c = x.shape
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(n * out_hw[0] * out_hw[1] * c * 2)
    _chk(H.si_hip_adaptive_avgpool2d_f16(dx.ptr, n, ih, iw, c, c, dy.ptr, out_hw[0], out_hw[1], c, None),
         "si_hip_adaptive_avgpool2d_f16")
    return dy.to_numpy((n, out_hw[0], out_hw[1], c), np.float16)


def activation_f16(kind, x, param=0.0):
    H = _native.hip()
    x = _f16(x)
    c = x.shape[-1]
    pixels = x.size // c
    dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(x.nbytes)
    _chk(H.si_hip_activation_f16(ACT[kind], float(param), dx.ptr, pixels, c, c, dy.ptr, c, None), "si_hip_activation_f16")
    return dy.to_numpy(x.shape, np.float16)


def binary_same_f16(op, a, b):
    H = _native.hip()
    a, b = _f16(a), _f16(b)
    c = a.shape[-1]
    pixels = a.size // c
    da, db, dy = DeviceBuffer.from_numpy(a), DeviceBuffer.from_numpy(b), DeviceBuffer(a.nbytes)
    _chk(H.si_hip_binary_same_f16({"add": 0, "mul": 2}[op], da.ptr, c, db.ptr, c, dy.ptr, c, pixels, c, None),
         "si_hip_binary_same_f16")
    return dy.to_numpy(a.shape, np.float16)


def binary_bcast_f16(op, a, s):
    """si_hip_binary_bcast_f16: a [n][h][w][c] (op) s [n][c] broadcast over h, w (the squeeze-excite scale)."""
    H = _native.hip()
    a, s = _f16(a), _f16(s)
    n, c = a.shape[0], a.shape[-1]
    ppi = a.size // (n * c)
    da, ds, dy = DeviceBuffer.from_numpy(a), DeviceBuffer.from_numpy(s), DeviceBuffer(a.nbytes)
    _chk(H.si_hip_binary_bcast_f16({"add": 0, "mul": 2}[op], da.ptr, c, ds.ptr, c, dy.ptr, c, n, ppi, c, None), "si_hip_binary_bcast_f16")
    return dy.to_numpy(a.shape, np.float16)


def convert_roundtrip_f16(x):
    """fp32 -> fp16 -> fp32 on the device (si_hip_convert_f32_f16 / si_hip_convert_f16_f32)."""
    H = _native.hip()
    x = _f32(x)
    c = x.shape[-1]
    pixels = x.size // c
    dx, dh, dy = DeviceBuffer.from_numpy(x), DeviceBuffer(x.size * 2), DeviceBuffer(x.nbytes)
    _chk(H.si_hip_convert_f32_f16(dx.ptr, pixels, c, c, dh.ptr, c, None), "si_hip_convert_f32_f16")
    half = dh.to_numpy(x.shape, np.float16)
    _chk(H.si_hip_convert_f16_f32(dh.ptr, pixels, c, c, dy.ptr, c, None), "si_hip_convert_f16_f32")
    return half, dy.to_numpy(x.shape)


def conv_stem_s2c32_f16(x, w0, b0, w1, b1):
    """si_hip_conv2d_stem_s2c32_f16: YOLOv5's first two convs (6x6 s2 p2 3 -> 32 SiLU, 3x3 s2 p1 32 -> oc SiLU) in one launch.
    x fp32 [n][h][w][3]; returns fp16 [n][oh][ow][oc]."""
    H = _native.hip()
    x, w0, w1 = _f32(x), _f32(w0), _f32(w1)
    n, ih, iw, _ = x.shape
    sh_, sw_ = conv_out_hw(ih, iw, (6, 6), (2, 2), (2, 2), (1, 1))
    oh, ow = conv_out_hw(sh_, sw_, (3, 3), (2, 2), (1, 1), (1, 1))
    oc = w1.shape[0]
    d0 = SiConv2dDesc(n, ih, iw, 3, 3, sh_, sw_, 32, 32, 6, 6, 2, 2, 1, 1, 2, 2, 1, 1 if b0 is not None else 0, ACT["silu"], 0, 32, 0, 0.0)
    d1 = SiConv2dDesc(n, sh_, sw_, 32, 32, oh, ow, oc, oc, 3, 3, 2, 2, 1, 1, 1, 1, 1, 1 if b1 is not None else 0, ACT["silu"], 0, oc, 0, 0.0)
    if not H.si_hip_conv2d_stem_s2c32_f16_supported(C.byref(d0), C.byref(d1)):
        raise HipError("si_hip_conv2d_stem_s2c32_f16: unsupported shape")
    p0 = np.zeros(H.si_hip_conv2d_stem_f16_weight_elems(C.byref(d0)), np.float16)
    _chk(H.si_hip_conv2d_stem_f16_pack_weight_host(C.byref(d0), w0.ctypes.data_as(C.c_void_p), p0.ctypes.data_as(C.c_void_p)), "pack stem f16")
    p1 = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d1)), np.float16)
    _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d1), w1.ctypes.data_as(C.c_void_p), p1.ctypes.data_as(C.c_void_p)), "pack f16")
    dx, dp0, dp1 = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(p0), DeviceBuffer.from_numpy(p1)
    db0 = DeviceBuffer.from_numpy(_f32(b0)) if b0 is not None else None
    db1 = DeviceBuffer.from_numpy(_f32(b1)) if b1 is not None else None
    dy = DeviceBuffer(n * oh * ow * oc * 2)
    dy.fill(0)
    _chk(H.si_hip_conv2d_stem_s2c32_f16(C.byref(d0), C.byref(d1), dx.ptr, dp0.ptr, db0.ptr if db0 else None, dp1.ptr,
                                        db1.ptr if db1 else None, dy.ptr, None), "si_hip_conv2d_stem_s2c32_f16")
    sync()
    return dy.to_numpy((n, oh, ow, oc), np.float16)


def conv_stem_s2c32_pw_f16(x, w0, b0, w1, b1, w2, b2, split_oc=32, out2_ld=None, out2_c_off=0):
    """si_hip_conv2d_stem_s2c32_pw_f16: YOLOv5's first two convs AND the 1x1 conv behind them (64 -> 64, SiLU: the first C3's cv1 | cv2 over the
    concatenated filters w2 [64][64][1][1]) in one launch.  x fp32 [n][h][w][3].  split_oc = 32: returns (fp16 [n][oh][ow][32], fp16
    [n][oh][ow][32]) -- the second one written at channel offset out2_c_off of a buffer with pixel stride out2_ld; split_oc = 0: one fp16
    [n][oh][ow][64]."""
    H = _native.hip()
    x, w0, w1, w2 = _f32(x), _f32(w0), _f32(w1), _f32(w2)
    n, ih, iw, _ = x.shape
    sh_, sw_ = conv_out_hw(ih, iw, (6, 6), (2, 2), (2, 2), (1, 1))
    oh, ow = conv_out_hw(sh_, sw_, (3, 3), (2, 2), (1, 1), (1, 1))
    oc = w1.shape[0]
    d0 = SiConv2dDesc(n, ih, iw, 3, 3, sh_, sw_, 32, 32, 6, 6, 2, 2, 1, 1, 2, 2, 1, 1 if b0 is not None else 0, ACT["silu"], 0, 32, 0, 0.0)
    d1 = SiConv2dDesc(n, sh_, sw_, 32, 32, oh, ow, oc, oc, 3, 3, 2, 2, 1, 1, 1, 1, 1, 1 if b1 is not None else 0, ACT["silu"], 0, oc, 0, 0.0)
    ld_a = 32 if split_oc else 64
    d2 = SiConv2dDesc(n, oh, ow, oc, oc, oh, ow, 64, ld_a, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1 if b2 is not None else 0, ACT["silu"], 0, 64, 0, 0.0)
    if not H.si_hip_conv2d_stem_s2c32_pw_f16_supported(C.byref(d0), C.byref(d1), C.byref(d2), split_oc):
        raise HipError("si_hip_conv2d_stem_s2c32_pw_f16: unsupported shape")
    p0 = np.zeros(H.si_hip_conv2d_stem_f16_weight_elems(C.byref(d0)), np.float16)
    _chk(H.si_hip_conv2d_stem_f16_pack_weight_host(C.byref(d0), w0.ctypes.data_as(C.c_void_p), p0.ctypes.data_as(C.c_void_p)), "pack stem f16")
    p1 = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d1)), np.float16)
    _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d1), w1.ctypes.data_as(C.c_void_p), p1.ctypes.data_as(C.c_void_p)), "pack f16")
    p2 = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d2)), np.float16)
    _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d2), w2.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p)), "pack f16")
    bufs = [DeviceBuffer.from_numpy(v) for v in (x, p0, p1, p2)]
    db = [DeviceBuffer.from_numpy(_f32(b)) if b is not None else None for b in (b0, b1, b2)]
    out2_ld = out2_ld or 32
    dy = DeviceBuffer(n * oh * ow * ld_a * 2)
    dy.fill(0)
    dy2 = DeviceBuffer(n * oh * ow * out2_ld * 2)
    dy2.fill(0)
    _chk(H.si_hip_conv2d_stem_s2c32_pw_f16(C.byref(d0), C.byref(d1), C.byref(d2), bufs[0].ptr, bufs[1].ptr, db[0].ptr if db[0] else None, bufs[2].ptr,
                                           db[1].ptr if db[1] else None, bufs[3].ptr, db[2].ptr if db[2] else None, dy.ptr, split_oc,
                                           dy2.ptr + 2 * out2_c_off if split_oc else None, out2_ld, None), "si_hip_conv2d_stem_s2c32_pw_f16")
    sync()
    if not split_oc:
        return dy.to_numpy((n, oh, ow, 64), np.float16)
    y2 = dy2.to_numpy((n, oh, ow, out2_ld), np.float16)
    return dy.to_numpy((n, oh, ow, 32), np.float16), y2[..., out2_c_off:out2_c_off + 32].copy()


def conv_pw_slab_f16(x, w0, b0, w1, b1, residual=None, out_ld=None, out_c_off=0, in_ld=None):
    """si_hip_conv2d_pw_slab_f16: the C3 bottleneck's two convs (1x1 c -> c SiLU, 3x3 s1 p1 c -> oc SiLU, optional shortcut) in one
    launch.  x NHWC fp16; returns NHWC fp16."""
    H = _native.hip()
    x, w0, w1 = _f16(x), _f32(w0), _f32(w1)
    n, ih, iw, c = x.shape
    oc = w1.shape[0]
    in_ld = in_ld or c
    out_ld = out_ld or oc
    d0 = SiConv2dDesc(n, ih, iw, c, in_ld, ih, iw, c, c, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1 if b0 is not None else 0, ACT["silu"], 0, c, 0, 0.0)
    d1 = SiConv2dDesc(n, ih, iw, c, c, ih, iw, oc, out_ld, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1 if b1 is not None else 0, ACT["silu"],
                      1 if residual is not None else 0, oc, 0, 0.0)
    if not H.si_hip_conv2d_pw_slab_f16_supported(C.byref(d0), C.byref(d1)):
        raise HipError("si_hip_conv2d_pw_slab_f16: unsupported shape")
    p0 = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d0)), np.float16)
    _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d0), w0.ctypes.data_as(C.c_void_p), p0.ctypes.data_as(C.c_void_p)), "pack 1x1")
    p1 = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d1)), np.float16)
    _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d1), w1.ctypes.data_as(C.c_void_p), p1.ctypes.data_as(C.c_void_p)), "pack 3x3")
    xs = x
    if in_ld != c:
        xs = np.zeros((n, ih, iw, in_ld), np.float16)
        xs[..., :c] = x
    dx, dp0, dp1 = DeviceBuffer.from_numpy(xs), DeviceBuffer.from_numpy(p0), DeviceBuffer.from_numpy(p1)
    db0 = DeviceBuffer.from_numpy(_f32(b0)) if b0 is not None else None
    db1 = DeviceBuffer.from_numpy(_f32(b1)) if b1 is not None else None
    dr = DeviceBuffer.from_numpy(_f16(residual)) if residual is not None else None
    dy = DeviceBuffer(n * ih * iw * out_ld * 2)
    dy.fill(0)
    _chk(H.si_hip_conv2d_pw_slab_f16(C.byref(d0), C.byref(d1), dx.ptr, dp0.ptr, db0.ptr if db0 else None, dp1.ptr, db1.ptr if db1 else None,
                                     dr.ptr if dr else None, dy.ptr + 2 * out_c_off, None), "si_hip_conv2d_pw_slab_f16")
    sync()
    y = dy.to_numpy((n, ih, iw, out_ld), np.float16)
    return y[..., out_c_off:out_c_off + oc].copy() if out_ld != oc else y


def conv_pw_cv3_f16(x, w0, b0, w1, b1, z, w3, b3, residual=None, z_ld=None, z_c_off=0, out_ld=None, out_c_off=0):
    """si_hip_conv2d_pw_cv3_f16: a C3's last bottleneck pair (1x1 c -> c SiLU, 3x3 c -> c SiLU, optional shortcut) AND the C3's closing 1x1
    conv over cat([pair output, z]) in one launch (c = 64, z 64 channels, w3 [128][128][1][1]).  x, z, residual NHWC fp16; returns NHWC fp16
    [n][h][w][128].  z_ld / z_c_off: z read as a channel slice of a wider buffer."""
    H = _native.hip()
    x, z, w0, w1, w3 = _f16(x), _f16(z), _f32(w0), _f32(w1), _f32(w3)
    n, ih, iw, c = x.shape
    oc3 = w3.shape[0]
    z_ld = z_ld or z.shape[-1]
    out_ld = out_ld or oc3
    d0 = SiConv2dDesc(n, ih, iw, c, c, ih, iw, c, c, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1 if b0 is not None else 0, ACT["silu"], 0, c, 0, 0.0)
    d1 = SiConv2dDesc(n, ih, iw, c, c, ih, iw, c, c, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1 if b1 is not None else 0, ACT["silu"],
                      1 if residual is not None else 0, c, 0, 0.0)
    d2 = SiConv2dDesc(n, ih, iw, 2 * c, 2 * c, ih, iw, oc3, out_ld, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1 if b3 is not None else 0, ACT["silu"], 0, oc3, 0, 0.0)
    if not H.si_hip_conv2d_pw_cv3_f16_supported(C.byref(d0), C.byref(d1), C.byref(d2)):
        raise HipError("si_hip_conv2d_pw_cv3_f16: unsupported shape")
    packs = []
    for d, w in ((d0, w0), (d1, w1), (d2, w3)):
        p = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d)), np.float16)
        _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p)), "pack f16")
        packs.append(DeviceBuffer.from_numpy(p))
    zs = z
    if z_ld != z.shape[-1] or z_c_off:
        zs = np.zeros((n, ih, iw, z_ld), np.float16)
        zs[..., z_c_off:z_c_off + z.shape[-1]] = z
    dx, dz = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(zs)
    db = [DeviceBuffer.from_numpy(_f32(b)) if b is not None else None for b in (b0, b1, b3)]
    dr = DeviceBuffer.from_numpy(_f16(residual)) if residual is not None else None
    dy = DeviceBuffer(n * ih * iw * out_ld * 2)
    dy.fill(0)
    _chk(H.si_hip_conv2d_pw_cv3_f16(C.byref(d0), C.byref(d1), C.byref(d2), dx.ptr, packs[0].ptr, db[0].ptr if db[0] else None, packs[1].ptr,
                                    db[1].ptr if db[1] else None, dr.ptr if dr else None, dz.ptr + 2 * z_c_off, z_ld, packs[2].ptr,
                                    db[2].ptr if db[2] else None, dy.ptr + 2 * out_c_off, None), "si_hip_conv2d_pw_cv3_f16")
    sync()
    y = dy.to_numpy((n, ih, iw, out_ld), np.float16)
    return y[..., out_c_off:out_c_off + oc3].copy() if out_ld != oc3 else y


def yolo_detect_f16(feats, weights, biases, grids, anchor_grids, strides, na=3):
    """si_hip_conv2d_yolo_f16 per level: fp16 features, fp32 [n][rows_total][ne] detections."""
    H = _native.hip()
    feats = [_f16(f) for f in feats]
    n = feats[0].shape[0]
    ne = weights[0].shape[0] // na
    rows_total = sum(f.shape[1] * f.shape[2] * na for f in feats)
    dout = DeviceBuffer(n * rows_total * ne * 4)
    off = 0
    for f, w, b, g, a, s in zip(feats, weights, biases, grids, anchor_grids, strides):
        _, h, wd, cin = f.shape
        w = _f32(w)
        d = SiConv2dDesc(n, h, wd, cin, cin, h, wd, na * ne, na * ne, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, na * ne, 0, 0.0)
        packed = np.zeros(H.si_hip_conv2d_f16_weight_elems(C.byref(d)), np.float16)
        _chk(H.si_hip_conv2d_f16_pack_weight_host(C.byref(d), w.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "pack")
        g2 = _f32(np.transpose(_f32(g)[0], (1, 2, 0, 3)))
        a2 = _f32(np.transpose(_f32(a)[0], (1, 2, 0, 3)))
        bufs = [DeviceBuffer.from_numpy(v) for v in (f, packed, _f32(b), g2, a2)]
        lv = _native.SiYoloLevel(na, ne, rows_total, off, float(s))
        _chk(H.si_hip_conv2d_yolo_f16(C.byref(d), bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, C.byref(lv), bufs[3].ptr, bufs[4].ptr,
                                      dout.ptr, None), "si_hip_conv2d_yolo_f16")
        sync()
        off += h * wd * na
    return dout.to_numpy((n, rows_total, ne))
