"""GPU: the fp16 storage path (BASELINE.json configs[3]).  The reference is fp32 only, so the yardstick is the fp32
oracle evaluated on the SAME fp16-rounded inputs and weights: what is left is fp32 accumulation order plus one rounding
of the result to fp16 (relative 2^-11 = 4.9e-4), hence the 1e-3 bound on max|diff| / max|ref|; copies, pooling maxima
and conversions are exact."""
import numpy as np
import pytest

from util import assert_detect_parity, assert_exact, assert_parity, rng_uniform

pytestmark = pytest.mark.gpu

F16_TOL = 1e-3


@pytest.fixture(scope="module")
def hops(gpu):
    from simpleinfer_amd import hipops
    return hipops


def h(a):
    """round to fp16 and come back: the value the device actually sees"""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


CONV_CASES = [
    # n, ih, iw, ic, oc, k, s, p, d, g
    (2, 20, 20, 64, 64, 1, 1, 0, 1, 1),
    (2, 20, 20, 32, 64, 3, 1, 1, 1, 1),
    (2, 21, 19, 64, 128, 3, 2, 1, 1, 1),
    (1, 12, 12, 256, 96, 3, 1, 1, 1, 1),     # long K, ragged 32-wide column tile
    (3, 9, 9, 64, 48, 3, 1, 2, 2, 1),        # dilation, oc not a multiple of 32
    (2, 10, 10, 64, 64, 3, 1, 1, 1, 2),      # grouped (32 channels per group)
    (5, 4, 4, 128, 256, 1, 1, 0, 1, 1),
    (1, 40, 40, 32, 32, 5, 1, 2, 1, 1),
]


@pytest.mark.parametrize("n,ih,iw,ic,oc,k,s,p,d,g", CONV_CASES)
def test_conv_f16_vs_oracle_on_rounded_operands(hops, orc, n, ih, iw, ic, oc, k, s, p, d, g):
    x = h(rng_uniform(ih * 7 + ic, (n, ih, iw, ic), -1, 1))
    w = h(rng_uniform(ih * 7 + ic + 1, (oc, ic // g, k, k), -0.3, 0.3))
    b = rng_uniform(ih * 7 + ic + 2, (oc,), -0.5, 0.5)
    ref = orc.conv2d(x, w, b, (s, s), (p, p), (d, d), g, path="naive")
    got = hops.conv2d_f16(x, w, b, (s, s), (p, p), (d, d), g)
    assert got.dtype == np.float16
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="fp16 conv")
    # fp32 store of the same accumulators: only the accumulation order is left
    assert_parity(hops.conv2d_f16(x, w, b, (s, s), (p, p), (d, d), g, out_f32=True), ref, 2e-5, what="fp16 conv, fp32 out")


def test_conv_f16_fused_epilogue_strides_and_batch_invariance(hops, orc):
    x = h(rng_uniform(1, (2, 16, 16, 64), -1, 1))
    w = h(rng_uniform(2, (64, 64, 3, 3), -0.2, 0.2))
    b = rng_uniform(3, (64,), -0.5, 0.5)
    r = h(rng_uniform(4, (2, 16, 16, 64), -1, 1))
    y = orc.conv2d(x, w, b, (1, 1), (1, 1), path="naive")
    got = hops.conv2d_f16(x, w, b, (1, 1), (1, 1), act1="silu", residual=r).astype(np.float32)
    assert_parity(got, orc.activation("silu", y) + r, F16_TOL, what="silu + residual")
    got = hops.conv2d_f16(x, w, b, (1, 1), (1, 1), residual=r, act2="relu").astype(np.float32)
    assert_parity(got, orc.activation("relu", y + r), F16_TOL, what="residual + relu")
    got = hops.conv2d_f16(x, w, b, (1, 1), (1, 1), in_ld=96, out_ld=160, out_c_off=32).astype(np.float32)
    assert_parity(got, y, F16_TOL, what="strided tensors")
    full = hops.conv2d_f16(x, w, b, (1, 1), (1, 1), act1="silu")
    assert_exact(hops.conv2d_f16(x[1:2], w, b, (1, 1), (1, 1), act1="silu")[0], full[1], "an image's result does not depend on the batch")
    with pytest.raises(hops.HipError):
        hops.conv2d_f16(h(rng_uniform(5, (1, 8, 8, 12), -1, 1)), h(rng_uniform(6, (32, 12, 3, 3))), None)  # ic % 8 != 0: no 16-byte channel vectors


@pytest.mark.parametrize("ic,oc", [(16, 64), (24, 72), (40, 120), (72, 24), (96, 576), (8, 16), (144, 40)])
def test_conv_f16_pointwise_padded_k(hops, orc, ic, oc):
    """1x1 convs whose channel count is a multiple of 8 but not of 32 (MobileNet's pointwise and squeeze-excite convs): the weights'
    K axis is zero-padded to whole 32-channel blocks and a vector behind the last channel reads zeros -- shown with NaNs between the
    pixels' channels of a wider (concat-slice) input."""
    x = h(rng_uniform(ic, (2, 9, 7, ic), -1, 1))
    w = h(rng_uniform(ic + 1, (oc, ic, 1, 1), -0.5, 0.5))
    b = rng_uniform(ic + 2, (oc,), -0.5, 0.5)
    ref = orc.conv2d(x, w, b, path="naive")
    assert_parity(hops.conv2d_f16(x, w, b).astype(np.float32), ref, F16_TOL, what="dense")
    assert_parity(hops.conv2d_f16(x, w, b, out_f32=True), ref, 2e-5, what="fp32 out")
    assert_parity(hops.conv2d_f16(x, w, b, act1="hardswish").astype(np.float32), orc.activation("hardswish", ref), F16_TOL, what="hardswish")
    got = hops.conv2d_f16(x, w, b, in_ld=ic + 24, in_fill=np.nan)
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="strided input, NaN beyond the channels")


@pytest.mark.parametrize("ic,oc,k,s", [(8, 16, 3, 2), (24, 32, 3, 1), (40, 72, 5, 1), (16, 8, 3, 1), (72, 130, 3, 2)])
def test_conv_f16_spatial_padded_k(hops, orc, ic, oc, k, s):
    """Round 5: ANY ungrouped conv over a multiple of 8 channels has an fp16 kernel (until then the 1x1 ones only: a narrow YOLOv5's
    3x3 convs ran in fp32 between two casts).  Every tap's channels are zero-padded to whole 32-channel blocks in the weights and a
    vector behind the last channel reads zeros -- shown with NaNs between the pixels' channels of a wider input; borders, strides,
    the fused epilogue, and an image's bits do not depend on its batch position (src/layer/conv_2d.cpp:207-283)."""
    p = k // 2
    x = h(rng_uniform(300 + ic, (3, 11, 9, ic), -1, 1))
    w = h(rng_uniform(301 + ic, (oc, ic, k, k), -0.3, 0.3))
    b = rng_uniform(302 + ic, (oc,), -0.5, 0.5)
    ref = orc.conv2d(x, w, b, (s, s), (p, p), path="naive")
    got = hops.conv2d_f16(x, w, b, (s, s), (p, p))
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="dense")
    assert_parity(hops.conv2d_f16(x, w, b, (s, s), (p, p), out_f32=True), ref, 2e-5, what="fp32 out")
    assert_parity(hops.conv2d_f16(x, w, b, (s, s), (p, p), act1="silu").astype(np.float32), orc.activation("silu", ref), F16_TOL, what="silu")
    wide = hops.conv2d_f16(x, w, b, (s, s), (p, p), in_ld=ic + 24, in_fill=np.nan)
    assert_exact(wide, got, "strided input, NaN beyond the channels")
    assert_exact(hops.conv2d_f16(x[2:], w, b, (s, s), (p, p)), got[2:], "batch position")


@pytest.mark.parametrize("n,hw,c,k,s,p", [(2, 14, 16, 3, 1, 1), (2, 15, 72, 5, 2, 2), (3, 7, 96, 3, 2, 1), (1, 9, 240, 5, 1, 2), (2, 6, 8, 3, 1, 1)])
def test_conv_f16_depthwise(hops, orc, n, hw, c, k, s, p):
    """depthwise convolution with fp16 activations (fp32 weights / bias / tap sums): against the fp32 oracle on the same rounded
    input, fused epilogues, strided tensors, batch invariance bit for bit."""
    x = h(rng_uniform(c, (n, hw, hw, c), -1, 1))
    w = rng_uniform(c + 1, (c, 1, k, k), -0.5, 0.5)
    b = rng_uniform(c + 2, (c,), -0.5, 0.5)
    ref = orc.conv2d(x, w, b, (s, s), (p, p), (1, 1), c, path="naive")
    got = hops.conv2d_f16(x, w, b, (s, s), (p, p), (1, 1), c)
    assert got.dtype == np.float16
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="depthwise fp16")
    assert_parity(hops.conv2d_f16(x, w, b, (s, s), (p, p), (1, 1), c, act1="hardswish").astype(np.float32), orc.activation("hardswish", ref), F16_TOL,
                  what="hardswish")
    r = h(rng_uniform(c + 3, ref.shape, -1, 1))
    assert_parity(hops.conv2d_f16(x, w, b, (s, s), (p, p), (1, 1), c, residual=r, act2="relu").astype(np.float32), orc.activation("relu", ref + r), F16_TOL,
                  what="residual + relu")
    assert_parity(hops.conv2d_f16(x, w, b, (s, s), (p, p), (1, 1), c, in_ld=c + 8, out_ld=c + 16, out_c_off=8).astype(np.float32), ref, F16_TOL,
                  what="strided tensors")
    if n > 1:
        assert_exact(hops.conv2d_f16(x[1:2], w, b, (s, s), (p, p), (1, 1), c)[0], got[1], "batch invariance")


def test_binary_broadcast_f16(hops):
    a = h(rng_uniform(1, (3, 5, 7, 72), -2, 2))
    s = h(rng_uniform(2, (3, 72), 0, 1))
    for op, fn in (("mul", np.multiply), ("add", np.add)):
        want = fn(a, s[:, None, None, :]).astype(np.float16)
        assert_exact(hops.binary_bcast_f16(op, a, s), want, "broadcast " + op)


@pytest.mark.parametrize("hw,ic,oc,k,p,res", [(41, 32, 32, 3, 1, False), (41, 64, 64, 1, 0, False), (41, 32, 32, 3, 1, True),
                                             (5, 128, 96, 1, 0, False), (21, 64, 128, 3, 1, True)])
def test_conv_f16_batch_invariance_with_odd_pixel_counts(hops, hw, ic, oc, k, p, res):
    """An odd pixel count per image moves image 1 to other accumulator slots than it has alone.  The fp32 -> fp16 store
    must round the same way in every slot (a compiler-chosen v_fma_mix for some slots rounds once, the others twice)."""
    x = h(rng_uniform(1, (3, hw, hw, ic), -1, 1))
    w = h(rng_uniform(2, (oc, ic, k, k), -0.1, 0.1))
    b = rng_uniform(3, (oc,), -0.05, 0.05)
    r = h(rng_uniform(4, (3, hw, hw, oc), -1, 1)) if res else None
    for act in ("silu", "relu", "none"):
        full = hops.conv2d_f16(x, w, b, (1, 1), (p, p), act1=act, residual=r)
        for i in (1, 2):
            one = hops.conv2d_f16(x[i:i + 1], w, b, (1, 1), (p, p), act1=act, residual=None if r is None else r[i:i + 1])
            assert_exact(one[0], full[i], "image %d, act %s" % (i, act))


STEM_CASES = [
    # k, s, p, oc, n, h, w, act          (6x6 = YOLOv5, 7x7 = ResNet, 3x3 = MobileNet stems)
    (6, 2, 2, 32, 2, 64, 64, "silu"),
    (6, 2, 2, 32, 3, 40, 328, "silu"),      # two 160-pixel column tiles + a ragged one, odd row-block count
    (6, 2, 2, 64, 2, 32, 48, "relu"),       # 64 channels: two channel waves per workgroup
    (6, 2, 1, 32, 2, 30, 36, "none"),       # pad 1: fragments start on odd half indices (funnel-shift path)
    (7, 2, 3, 64, 2, 56, 56, "relu"),
    (7, 2, 3, 32, 1, 33, 47, "silu"),       # odd width: element-wise row loads
    (3, 2, 1, 16, 2, 48, 48, "hardswish"),
    (3, 2, 1, 96, 1, 24, 40, "silu"),       # > 64 channels: channel tiles as items
    (3, 1, 1, 32, 1, 20, 24, "relu"),       # stride 1
]


@pytest.mark.parametrize("k,s,p,oc,n,ih,iw,act", STEM_CASES)
def test_stem_f16_output(hops, orc, k, s, p, oc, n, ih, iw, act):
    """The stem of the fp16 path reads the fp32 image, rounds it to fp16 on the way into LDS and contracts on the fp16
    matrix cores: the yardstick is the oracle on fp16-rounded image and weights (same contract as every other fp16 conv)."""
    x = rng_uniform(10 + k, (n, ih, iw, 3), 0, 1)
    w = rng_uniform(11 + k, (oc, 3, k, k), -0.3, 0.3)
    b = rng_uniform(12 + k, (oc,), -0.5, 0.5)
    fact = (lambda t: t) if act == "none" else (lambda t: orc.activation(act, t))
    ref = fact(orc.conv2d(h(x), h(w), b, (s, s), (p, p), path="naive"))
    got = hops.conv2d_f16(x, w, b, (s, s), (p, p), act1=act)
    assert got.dtype == np.float16 and got.shape == ref.shape
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="stem %dx%d" % (k, k))
    # against the unrounded operands the error is the fp16 rounding of the image and the weights
    full = fact(orc.conv2d(x, w, b, (s, s), (p, p), path="naive"))
    assert_parity(got.astype(np.float32), full, 4 * F16_TOL, what="stem vs fp32 operands")
    if n > 1:
        one = hops.conv2d_f16(x[n - 1:n], w, b, (s, s), (p, p), act1=act)
        assert_exact(one[0], got[n - 1], "an image's result does not depend on the batch")


def test_stem_f16_strided_output_and_two_channel_image(hops, orc):
    x = rng_uniform(31, (2, 32, 32, 2), 0, 1)
    w = rng_uniform(32, (32, 2, 3, 3), -0.3, 0.3)   # kw*c = 6 does not form a supported kernel row: no stem kernel
    with pytest.raises(hops.HipError):
        hops.conv2d_f16(x, w, None, (2, 2), (1, 1))
    x = rng_uniform(33, (2, 32, 32, 3), 0, 1)
    w = rng_uniform(34, (32, 3, 6, 6), -0.3, 0.3)
    ref = orc.conv2d(h(x), h(w), None, (2, 2), (2, 2), path="naive")
    got = hops.conv2d_f16(x, w, None, (2, 2), (2, 2), in_ld=4, out_ld=96, out_c_off=32)   # strided image: element-wise loads
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="strided stem")


def test_maxpool5_chain3_f16_exact(hops, orc):
    x = h(rng_uniform(71, (2, 20, 20, 64), -3, 3))
    cur = x
    got = hops.maxpool5_chain3(x, half=True, out_ld=256, out_c_off=(64, 128, 192))
    for k in range(3):
        cur = orc.maxpool2d(cur, (5, 5), (1, 1), (2, 2))
        assert_exact(got[k].astype(np.float32), cur, "fp16 stage %d" % k)


def test_conv_split_f16(hops, orc):
    x = h(rng_uniform(20, (2, 12, 12, 64), -1, 1))
    wa, wb = h(rng_uniform(21, (32, 64, 1, 1), -0.3, 0.3)), h(rng_uniform(22, (64, 64, 1, 1), -0.3, 0.3))
    ba, bb = rng_uniform(23, (32,), -0.5, 0.5), rng_uniform(24, (64,), -0.5, 0.5)
    ya, yb = hops.conv2d_split_f16(x, wa, ba, wb, bb, act1="silu")
    assert_parity(ya.astype(np.float32), orc.activation("silu", orc.conv2d(x, wa, ba, path="naive")), F16_TOL, what="first sibling")
    assert_parity(yb.astype(np.float32), orc.activation("silu", orc.conv2d(x, wb, bb, path="naive")), F16_TOL, what="second sibling")


@pytest.mark.parametrize("n,levels", [(2, ((8, 32), (4, 64), (2, 128))), (5, ((4, 32), (2, 64), (1, 96)))])
def test_yolo_detect_head_f16(hops, orc, n, levels):
    na, ne = 3, 85
    feats, ws, bs, grids, anchors = [], [], [], [], []
    for i, (hh, c) in enumerate(levels):
        feats.append(h(rng_uniform(50 + i, (n, hh, hh, c), -1, 1)))
        ws.append(h(rng_uniform(60 + i, (na * ne, c, 1, 1), -0.3, 0.3)))
        bs.append(rng_uniform(70 + i, (na * ne,), -0.5, 0.5))
        gy, gx = np.meshgrid(np.arange(hh, dtype=np.float32), np.arange(hh, dtype=np.float32), indexing="ij")
        grids.append(np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, hh, hh, 2)).copy())
        anchors.append(np.broadcast_to(rng_uniform(80 + i, (1, na, 1, 1, 2), 5, 300), (1, na, hh, hh, 2)).copy())
    strides = [8.0, 16.0, 32.0]
    ref = orc.yolo_detect(feats, ws, bs, grids, anchors, strides, na)
    got = hops.yolo_detect_f16(feats, ws, bs, grids, anchors, strides, na)
    assert got.dtype == np.float32
    assert_detect_parity(got, ref, 1e-4, 1e-4, what="fp16 features, fp32 decode")   # output is fp32: the fp32 bar applies, per column group


def test_pool_activation_binary_convert_f16(hops, orc):
    x = h(rng_uniform(30, (2, 20, 20, 64), -2, 2))
    assert_exact(hops.maxpool2d_f16(x, (5, 5), (1, 1), (2, 2)).astype(np.float32), orc.maxpool2d(x, (5, 5), (1, 1), (2, 2)), "maxpool k5")
    x3 = h(rng_uniform(31, (2, 15, 15, 12), -2, 2))                       # c % 8 != 0: scalar path
    assert_exact(hops.maxpool2d_f16(x3, (3, 3), (2, 2), (1, 1)).astype(np.float32), orc.maxpool2d(x3, (3, 3), (2, 2), (1, 1)), "maxpool k3s2")
    xa = h(rng_uniform(32, (2, 7, 7, 64), -1, 1))
    assert_parity(hops.adaptive_avgpool2d_f16(xa, (1, 1)).astype(np.float32), orc.adaptive_avgpool2d(xa, (1, 1)), F16_TOL, what="avgpool")
    for kind in ("relu", "silu", "sigmoid", "hardswish", "hardsigmoid"):
        assert_parity(hops.activation_f16(kind, x).astype(np.float32), orc.activation(kind, x), F16_TOL, what=kind)
    y = h(rng_uniform(33, (2, 20, 20, 64), -2, 2))
    assert_exact(hops.binary_same_f16("add", x, y), (x + y).astype(np.float16), "add rounds once")
    assert_exact(hops.binary_same_f16("mul", x, y), (x * y).astype(np.float16), "mul rounds once")
    z = rng_uniform(34, (3, 5, 5, 10), -70000, 70000)
    half, back = hops.convert_roundtrip_f16(z)
    with np.errstate(over="ignore"):
        assert_exact(half, z.astype(np.float16), "fp32 -> fp16 is round-to-nearest-even (overflow to inf)")
    assert_exact(back, half.astype(np.float32), "fp16 -> fp32 is exact")


# ---- tile variants of the fp16 implicit GEMM (round 4): the one-stage kernel (0-2) and the kernels that fetch the weights
# straight from L2 in MFMA lane order (3, 7, 9, 10, 11; seven more were measured and retired) must agree BIT FOR BIT -- the same k order through the same 16-deep MFMA steps --
# so that the tile policy may follow the launch size without touching the batch-invariance contract.
F16_TILES = [0, 1, 2, 3, 7, 9, 10, 11]


@pytest.fixture()
def tile16(gpu):
    from simpleinfer_amd import hipops

    def set_variant(v):
        hipops.set_plan(f16_tile=int(v))   # (SiConvPlan::f16_tile in every descriptor hipops builds from here on)
    yield set_variant
    hipops.set_plan()


F16_TILE_SHAPES = [
    # shape NHWC, oc, k, s, p, d, groups
    ((2, 21, 19, 64), 96, 3, 2, 1, 1, 1),     # 3x3 s2, ragged 32-wide column block, M not a multiple of any tile
    ((1, 20, 20, 256), 160, 3, 2, 1, 1, 1),   # K = 2304: 36 K-tiles (even), ring of two
    ((1, 11, 13, 192), 128, 3, 1, 1, 1, 1),   # K = 1728: 27 K-tiles (odd)
    ((3, 13, 17, 128), 64, 1, 1, 0, 1, 1),    # pointwise, two K-tiles
    ((2, 9, 9, 64), 32, 1, 1, 0, 1, 1),       # ONE K-tile, 32 output channels
    ((2, 12, 12, 64), 255, 1, 1, 0, 1, 1),    # ragged oc (Detect's channel count)
    ((2, 11, 11, 64), 64, 3, 1, 1, 1, 2),     # grouped, 32 channels per group (32-wide K blocks)
    ((2, 10, 10, 40), 72, 1, 1, 0, 1, 1),     # zero-padded K
    ((2, 12, 12, 32), 64, 3, 1, 2, 2, 1),     # 32-wide K blocks, dilation
    ((1, 40, 40, 32), 32, 5, 1, 2, 1, 1),     # 25 taps
]


@pytest.mark.parametrize("shape,oc,k,s,p,d,g", F16_TILE_SHAPES)
def test_every_f16_tile_variant_same_bits(hops, orc, tile16, shape, oc, k, s, p, d, g):
    seed = (shape[1] * 131 + shape[3] * 7 + oc) % 100000
    x = h(rng_uniform(seed, shape, -1, 1))
    w = h(rng_uniform(seed + 1, (oc, shape[3] // g, k, k), -0.3, 0.3))
    b = rng_uniform(seed + 2, (oc,), -0.5, 0.5)
    tile16(0)
    base32 = hops.conv2d_f16(x, w, b, (s, s), (p, p), (d, d), g, out_f32=True)
    base16 = hops.conv2d_f16(x, w, b, (s, s), (p, p), (d, d), g, act1="silu")
    ref = orc.conv2d(x, w, b, (s, s), (p, p), (d, d), g, path="naive")
    assert_parity(base32, ref, 2e-5, what="one-stage kernel, fp32 out")
    for v in F16_TILES[1:]:
        tile16(v)
        got32 = hops.conv2d_f16(x, w, b, (s, s), (p, p), (d, d), g, out_f32=True)
        bad = int((got32.view(np.uint32) != base32.view(np.uint32)).sum())
        assert bad == 0, "fp16 tile variant %d: %d of %d fp32 outputs differ from variant 0 (max abs %.3e)" % (
            v, bad, base32.size, float(np.abs(got32 - base32).max()))
        got16 = hops.conv2d_f16(x, w, b, (s, s), (p, p), (d, d), g, act1="silu")
        assert_exact(got16, base16, "fp16 tile variant %d, silu epilogue" % v)


def test_every_f16_tile_variant_same_bits_through_epilogues_split_and_detect(hops, tile16):
    x = h(rng_uniform(11, (2, 15, 13, 64), -1, 1))
    w = h(rng_uniform(12, (96, 64, 3, 3), -0.3, 0.3))
    b = rng_uniform(13, (96,), -0.5, 0.5)
    r = h(rng_uniform(14, (2, 15, 13, 96), -1, 1))
    xs = h(rng_uniform(20, (2, 12, 12, 64), -1, 1))
    wa, wb = h(rng_uniform(21, (32, 64, 1, 1), -0.3, 0.3)), h(rng_uniform(22, (64, 64, 1, 1), -0.3, 0.3))
    ba, bb = rng_uniform(23, (32,), -0.5, 0.5), rng_uniform(24, (64,), -0.5, 0.5)
    na, ne, n = 3, 85, 3
    feats, ws, bs, grids, anchors = [], [], [], [], []
    for i, (hh, c) in enumerate(((8, 64), (4, 128), (2, 256))):
        feats.append(h(rng_uniform(50 + i, (n, hh, hh, c), -1, 1)))
        ws.append(h(rng_uniform(60 + i, (na * ne, c, 1, 1), -0.3, 0.3)))
        bs.append(rng_uniform(70 + i, (na * ne,), -0.5, 0.5))
        gy, gx = np.meshgrid(np.arange(hh, dtype=np.float32), np.arange(hh, dtype=np.float32), indexing="ij")
        grids.append(np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, hh, hh, 2)).copy())
        anchors.append(np.broadcast_to(rng_uniform(80 + i, (1, na, 1, 1, 2), 5, 300), (1, na, hh, hh, 2)).copy())
    outs = {}
    for v in F16_TILES:
        tile16(v)
        outs[v] = (
            hops.conv2d_f16(x, w, b, (1, 1), (1, 1), act1="silu", residual=r, out_ld=128, out_c_off=16),
            hops.conv2d_f16(x, w, b, (1, 1), (1, 1), residual=r, act2="relu"),
            hops.conv2d_f16(x, w, b, (1, 1), (1, 1), act1="hardswish"),
        ) + tuple(hops.conv2d_split_f16(xs, wa, ba, wb, bb, act1="silu")) + (
            hops.yolo_detect_f16(feats, ws, bs, grids, anchors, [8.0, 16.0, 32.0], na),)
    for v in F16_TILES[1:]:
        for i, (got, want) in enumerate(zip(outs[v], outs[0])):
            assert_exact(got, want, "fp16 tile variant %d, output %d" % (v, i))


@pytest.mark.parametrize("n,ih,iw,oc,act,stride,res", [
    (2, 64, 64, 64, "silu", 2, False),       # whole 4 x 16 tiles
    (3, 38, 50, 64, "silu", 2, False),       # ragged tiles both ways (19 x 25 outputs), odd-sized borders
    (1, 7, 9, 32, "none", 2, False),         # one partial tile, a single 32-channel column block (8-row tiles)
    (5, 16, 32, 64, "none", 2, True),
    (2, 32, 32, 32, "silu", 1, True),        # stride 1: the C3 bottleneck's 3x3 with its shortcut (YOLOv5s conv_4)
    (3, 19, 25, 32, "silu", 1, True),        # ragged
    (1, 8, 16, 64, "silu", 1, False),
    (2, 21, 40, 64, "none", 1, True),
    (2, 20, 32, 64, "silu", -1, True),       # stride -1: 64 INPUT channels, stride 1 (the 80x80 C3 bottleneck convs; whole tiles only)
    (3, 12, 16, 64, "silu", -1, False),
    (2, 16, 48, 64, "relu", -1, False),      # ResNet's basic block: ReLU, then (second conv) shortcut + ReLU
    (2, 16, 48, 64, "res+relu", -1, True),
    (2, 14, 18, 32, "res+relu", 1, True),    # ... ragged, on the 32-channel form
    (2, 16, 64, 128, "silu", -2, False),     # stride -2: 64 input channels, stride 2, 128 outputs (YOLOv5s conv_7)
    (3, 12, 32, 128, "none", -2, True),
])
def test_s2c32_kernel_same_bits_as_generic_tiles(hops, orc, gpu, n, ih, iw, oc, act, stride, res):
    """Round 4: a 3x3 stride-2 pad-1 conv over 32 channels (YOLOv5's second conv) runs as the persistent spatial-tile kernel
    conv_s2c32_f16_kernel.  Same k order, same MFMA steps, same epilogue expressions as the generic tiles: BIT identical; and the
    fp16 bar against the oracle (src/layer/conv_2d.cpp:207-283) holds."""
    ic = 64 if stride < 0 else 32
    stride = 2 if stride == -2 else abs(stride)
    x = h(rng_uniform(700, (n, ih, iw, ic), -1, 1))
    w = h(rng_uniform(701, (oc, ic, 3, 3), -0.3, 0.3))
    b = rng_uniform(702, (oc,), -0.5, 0.5)
    kw = {} if act == "none" else ({"act2": "relu"} if act == "res+relu" else {"act1": act})
    st = (stride, stride)
    oh, ow = (ih + 2 - 3) // stride + 1, (iw + 2 - 3) // stride + 1
    if res:
        kw["residual"] = h(rng_uniform(703, (n, oh, ow, oc), -1, 1))
    with hops.plan(f16_s2c32=0):
        base = hops.conv2d_f16(x, w, b, st, (1, 1), **kw)
    got = hops.conv2d_f16(x, w, b, st, (1, 1), **kw)
    wide = hops.conv2d_f16(x, w, b, st, (1, 1), out_ld=oc + 32, out_c_off=16, **kw)   # into a slice of a wider tensor
    assert_exact(got, base, "c32 patch kernel vs generic tiles")
    assert_exact(wide, base, "c32 patch kernel, strided output")
    ref = orc.conv2d(x, w, b, st, (1, 1), path="naive")
    ref = ref if act in ("none", "res+relu") else orc.activation(act, ref)
    if res:
        ref = ref + kw["residual"].astype(np.float32)
    if act == "res+relu":
        ref = orc.activation("relu", ref)
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="c32 patch kernel")


@pytest.mark.parametrize("n,hh,ww,ic,oc,act,res", [
    (32, 40, 40, 128, 128, "silu", True),     # YOLOv5s' 40x40 bottleneck conv at the headline batch: 5-row slabs, 7 pixel blocks, two waves per SIMD
    (32, 20, 20, 256, 256, "silu", False),    # ... the 20x20 one: 4 pixel blocks, two output-channel groups per slab
    (2, 40, 40, 128, 128, "silu", False),     # a small grid takes finer slabs
    (3, 20, 20, 256, 256, "silu", True),      # ... 20x20 with the shortcut: two output-channel groups per slab, TM = 4
    (2, 13, 17, 128, 256, "none", True),      # ragged: the last slab of an image is short, the last 32-pixel block part empty
    (1, 7, 9, 256, 128, "relu", False),       # one slab per image
    (5, 28, 28, 128, 128, "res+relu", True),  # ResNet's basic block form
    (2, 3, 50, 128, 128, "silu", False),      # wide and flat
    (4, 14, 14, 256, 256, "relu", True),
    (2, 10, 10, 256, 128, "hardswish", False),  # not one of the kernel's activation pairs: the generic tiles serve it
])
def test_slab_kernel_same_bits_as_generic_tiles(hops, orc, gpu, n, hh, ww, ic, oc, act, res):
    """Round 5: 3x3 stride-1 pad-1 convs over 128 / 256 channels run as one-shot row slabs (conv_slab_f16.hip): the input rows of
    a slab staged once for all channel blocks, one barrier, the whole K loop from LDS with streamed lane-order weights.  Same k order,
    same MFMA steps, same epilogue expressions as the generic tiles: BIT identical, at any batch position, into strided tensors; and
    the fp16 bar against the oracle (src/layer/conv_2d.cpp:207-283) holds."""
    x = h(rng_uniform(900, (n, hh, ww, ic), -1, 1))
    w = h(rng_uniform(901, (oc, ic, 3, 3), -0.1, 0.1))
    b = rng_uniform(902, (oc,), -0.5, 0.5)
    kw = {} if act == "none" else ({"act2": "relu"} if act == "res+relu" else {"act1": act})
    if res:
        kw["residual"] = h(rng_uniform(903, (n, hh, ww, oc), -1, 1))
    with hops.plan(f16_slab=0):
        base = hops.conv2d_f16(x, w, b, (1, 1), (1, 1), **kw)
    got = hops.conv2d_f16(x, w, b, (1, 1), (1, 1), **kw)
    with hops.plan(f16_slab_w2=0):   # the one-wave-per-SIMD form of the 128-channel slab kernel (round 6: every W2 / NBLK / residual combination)
        assert_exact(hops.conv2d_f16(x, w, b, (1, 1), (1, 1), **kw), base, "slab kernel, one wave per SIMD")
    wide = hops.conv2d_f16(x, w, b, (1, 1), (1, 1), out_ld=oc + 32, out_c_off=16, **kw)   # into a slice of a wider tensor
    kw1 = dict(kw)
    if res:
        kw1["residual"] = kw["residual"][n - 1:]
    last = hops.conv2d_f16(x[n - 1:], w, b, (1, 1), (1, 1), **kw1)                          # the last image alone: another grid
    assert_exact(got, base, "slab kernel vs generic tiles")
    assert_exact(wide, base, "slab kernel, strided output")
    assert_exact(last, got[n - 1:], "slab kernel, batch position")
    if n > 8:
        return   # (the full-batch cases are GPU-against-GPU: the oracle's scalar loops are for the small ones)
    ref = orc.conv2d(x, w, b, (1, 1), (1, 1), path="naive")
    ref = ref if act in ("none", "res+relu") else orc.activation(act, ref)
    if res:
        ref = ref + kw["residual"].astype(np.float32)
    if act == "res+relu":
        ref = orc.activation("relu", ref)
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="slab kernel")


@pytest.mark.parametrize("n,hh,ww,c,res", [
    (2, 40, 40, 128, True),       # YOLOv5s' 40x40 bottleneck: 5-row slabs, x = 7 x 40 pixels in nine 32-pixel blocks
    (3, 20, 20, 256, True),       # ... 20x20: two output-channel groups per slab, each computes the whole 1x1
    (2, 40, 40, 128, False),      # the head's bottlenecks have no shortcut
    (2, 13, 17, 128, True),       # ragged: short last slab, image borders at every slab
    (1, 7, 9, 256, False),        # one slab per image: both zero rows in one patch
    (2, 16, 32, 64, True),        # 64 channels: the persistent patch form (conv_pw_patch_f16.hip), 4 x 16-pixel tiles, more items than ...
    (1, 80, 80, 64, True),        # ... YOLOv5s' 80x80 bottleneck
    (3, 8, 16, 64, False),        # two tiles per image, every patch touches three image borders
    (2, 16, 48, 32, True),        # 32 channels (YOLOv5s' first C3): 8 x 16-pixel tiles, 180-pixel patches in six pixel blocks
    (1, 8, 16, 32, False),
])
def test_bottleneck_pair_in_one_launch_same_bits(hops, orc, gpu, n, hh, ww, c, res):
    """Round 5: the C3 bottleneck's 1x1 conv + SiLU computed inside the slab kernel of the 3x3 conv that follows it
    (si_hip_conv2d_pw_slab_f16; computation of src/layer/conv_2d.cpp:207-283 twice): the same MFMA steps in the same k order, the
    same epilogue expressions, the intermediate rounded to fp16 exactly as the separate launch stores it, zero padding applied to the
    INTERMEDIATE (not to x) -- bit-identical to the two launches, with the shortcut, into strided tensors, at any batch position."""
    x = h(rng_uniform(910, (n, hh, ww, c), -1, 1))
    w0 = h(rng_uniform(911, (c, c, 1, 1), -0.15, 0.15))
    b0 = rng_uniform(912, (c,), -0.5, 0.5)
    w1 = h(rng_uniform(913, (c, c, 3, 3), -0.1, 0.1))
    b1 = rng_uniform(914, (c,), -0.5, 0.5)
    r = x if res else None
    mid = hops.conv2d_f16(x, w0, b0, (1, 1), (0, 0), act1="silu")
    want = hops.conv2d_f16(mid, w1, b1, (1, 1), (1, 1), act1="silu", residual=r)
    got = hops.conv_pw_slab_f16(x, w0, b0, w1, b1, residual=r)
    assert_exact(got, want, "fused bottleneck pair vs two launches")
    wide = hops.conv_pw_slab_f16(x, w0, b0, w1, b1, residual=r, out_ld=c + 32, out_c_off=16, in_ld=c + 8)
    assert_exact(wide, want, "fused pair, strided tensors")
    last = hops.conv_pw_slab_f16(x[n - 1:], w0, b0, w1, b1, residual=None if r is None else r[n - 1:])
    assert_exact(last, got[n - 1:], "fused pair, batch position")
    ref = orc.activation("silu", orc.conv2d(mid, w1, b1, (1, 1), (1, 1), path="naive"))
    if res:
        ref = ref + x.astype(np.float32)
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="fused pair (3x3 stage vs oracle on the rounded intermediate)")


@pytest.mark.parametrize("n,hh,ww,res", [
    (2, 16, 32, True),        # more items than workgroups may get, the shortcut
    (1, 80, 80, True),        # YOLOv5s' 80x80 C3 (backbone: with the shortcut)
    (2, 80, 80, False),       # ... the head's C3 (no shortcut)
    (3, 8, 16, False),        # two tiles per image, every patch touches three image borders
])
def test_c3_tail_pair_concat_cv3_in_one_launch_same_bits(hops, orc, gpu, n, hh, ww, res):
    """Round 6: a C3's last bottleneck pair AND its closing 1x1 conv over cat([pair output, z]) in one launch (si_hip_conv2d_pw_cv3_f16; the
    64-channel pair form): the pair's output tile is multiplied from LDS beside z's pixels, neither it nor the concat buffer is written.  The
    same MFMA steps in the same k order (y's 64-channel block, then z's), the same epilogue expressions, y rounded to fp16 exactly as the
    separate launch stores it: bit-identical to pair launch -> concat -> conv launch, with z read as a channel slice of a wider buffer and
    the output written into one, at any batch position; and the fp16 bar against the oracle holds (src/layer/conv_2d.cpp:207-283)."""
    c = 64
    x = h(rng_uniform(930, (n, hh, ww, c), -1, 1))
    z = h(rng_uniform(931, (n, hh, ww, c), -1, 1))
    w0, b0 = h(rng_uniform(932, (c, c, 1, 1), -0.15, 0.15)), rng_uniform(933, (c,), -0.5, 0.5)
    w1, b1 = h(rng_uniform(934, (c, c, 3, 3), -0.1, 0.1)), rng_uniform(935, (c,), -0.5, 0.5)
    w3, b3 = h(rng_uniform(936, (2 * c, 2 * c, 1, 1), -0.15, 0.15)), rng_uniform(937, (2 * c,), -0.5, 0.5)
    r = x if res else None
    y = hops.conv_pw_slab_f16(x, w0, b0, w1, b1, residual=r)
    want = hops.conv2d_f16(np.concatenate([y, z], -1), w3, b3, (1, 1), (0, 0), act1="silu")
    got = hops.conv_pw_cv3_f16(x, w0, b0, w1, b1, z, w3, b3, residual=r)
    assert_exact(got, want, "pair + concat + cv3 in one launch vs the launches it replaces")
    wide = hops.conv_pw_cv3_f16(x, w0, b0, w1, b1, z, w3, b3, residual=r, z_ld=160, z_c_off=64, out_ld=192, out_c_off=32)
    assert_exact(wide, want, "... z and the output as channel slices of wider buffers")
    last = hops.conv_pw_cv3_f16(x[n - 1:], w0, b0, w1, b1, z[n - 1:], w3, b3, residual=None if r is None else r[n - 1:])
    assert_exact(last, got[n - 1:], "... batch position")
    ref = orc.activation("silu", orc.conv2d(np.concatenate([y, z], -1).astype(np.float32), w3, b3, (1, 1), (0, 0), path="naive"))
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="pair + concat + cv3 (cv3 stage vs the oracle on the rounded intermediate)")


@pytest.mark.parametrize("n,ih,iw,oc", [
    (2, 128, 128, 64),     # whole tiles (32 x 32 outputs)
    (3, 76, 100, 64),      # ragged tiles both ways (19 x 25 outputs)
    (1, 20, 24, 32),       # one partial tile, a single 32-channel column block
    (5, 64, 192, 64),
])
def test_stem_s2c32_fused_same_bits(hops, orc, gpu, n, ih, iw, oc):
    """Round 4: YOLOv5's first two convs in one persistent kernel (si_hip_conv2d_stem_s2c32_f16): the 32-channel intermediate is
    computed tile by tile into LDS and never written.  Same MFMA steps, same epilogues, same fp16 rounding of the intermediate as
    si_hip_conv2d_stem_f16 followed by si_hip_conv2d_f16: BIT identical; and the fp16 bar against the oracle's two convs holds
    (src/layer/conv_2d.cpp:207-283)."""
    x = rng_uniform(800, (n, ih, iw, 3), 0, 1)
    w0, b0 = rng_uniform(801, (32, 3, 6, 6), -0.3, 0.3), rng_uniform(802, (32,), -0.5, 0.5)
    w1, b1 = h(rng_uniform(803, (oc, 32, 3, 3), -0.3, 0.3)), rng_uniform(804, (oc,), -0.5, 0.5)
    mid = hops.conv2d_f16(x, w0, b0, (2, 2), (2, 2), act1="silu")
    want = hops.conv2d_f16(mid, w1, b1, (2, 2), (1, 1), act1="silu")
    got = hops.conv_stem_s2c32_f16(x, w0, b0, w1, b1)
    assert_exact(got, want, "fused stem + conv vs the two launches")
    ref0 = orc.activation("silu", orc.conv2d(x, h(w0), b0, (2, 2), (2, 2), path="naive"))
    ref = orc.activation("silu", orc.conv2d(ref0, w1, b1, (2, 2), (1, 1), path="naive"))
    assert_parity(got.astype(np.float32), ref, F16_TOL, what="fused stem + conv")


@pytest.mark.parametrize("n,ih,iw,split", [
    (2, 128, 128, 32),     # whole tiles, the C3's split destination
    (3, 76, 100, 32),      # ragged tiles both ways
    (1, 20, 24, 0),        # one partial tile, all 64 columns to one tensor
    (4, 64, 192, 32),
])
def test_stem_pair_plus_pointwise_fused_same_bits(hops, orc, gpu, n, ih, iw, split):
    """Round 6: the 1x1 conv behind YOLOv5's first two convs (the first C3's cv1 | cv2 as ONE 64 -> 64 conv with a split destination) computed
    from the tile while it is in the CU (si_hip_conv2d_stem_s2c32_pw_f16): conv_1's 64-channel output is never written.  Same MFMA steps in the
    same k order, same epilogues and the same fp16 rounding of both intermediates as the three launches it replaces: BIT identical (also into a
    slice of a wider buffer); and the fp16 bar against the oracle's three convs holds (src/layer/conv_2d.cpp:207-283)."""
    x = rng_uniform(820, (n, ih, iw, 3), 0, 1)
    w0, b0 = rng_uniform(821, (32, 3, 6, 6), -0.3, 0.3), rng_uniform(822, (32,), -0.5, 0.5)
    w1, b1 = h(rng_uniform(823, (64, 32, 3, 3), -0.3, 0.3)), rng_uniform(824, (64,), -0.5, 0.5)
    w2, b2 = h(rng_uniform(825, (64, 64, 1, 1), -0.3, 0.3)), rng_uniform(826, (64,), -0.5, 0.5)
    mid = hops.conv_stem_s2c32_f16(x, w0, b0, w1, b1)
    want = hops.conv2d_f16(mid, w2, b2, act1="silu")
    if split:
        ya, yb = hops.conv_stem_s2c32_pw_f16(x, w0, b0, w1, b1, w2, b2, split_oc=32, out2_ld=64, out2_c_off=32)
        got = np.concatenate([ya, yb], -1)
        sa, sb = hops.conv2d_split_f16(mid, w2[:32], b2[:32], w2[32:], b2[32:], act1="silu")
        assert_exact(got, np.concatenate([sa, sb], -1), "fused stem + conv + 1x1 vs the sibling-fused launch")
    else:
        got = hops.conv_stem_s2c32_pw_f16(x, w0, b0, w1, b1, w2, b2, split_oc=0)
    assert_exact(got, want, "fused stem + conv + 1x1 vs the three launches")
    ref0 = orc.activation("silu", orc.conv2d(x, h(w0), b0, (2, 2), (2, 2), path="naive"))
    ref1 = orc.activation("silu", orc.conv2d(ref0, w1, b1, (2, 2), (1, 1), path="naive"))
    ref = orc.activation("silu", orc.conv2d(ref1, w2, b2, (1, 1), (0, 0), path="naive"))
    assert_parity(got.astype(np.float32), ref, 2 * F16_TOL, what="fused stem + conv + 1x1")


def test_fused_stem_kernels_full_size_same_bits(hops, gpu):
    """BASELINE's size (YOLOv5s 640 x 640, batch 32: 12 800 items, every workgroup walks ~25 of them): the stem pair and the stem triple against the
    launches they replace, bit for bit, over the WHOLE tensors -- the persistent kernels' buffer stores (out-of-range cases as offsets, the pixel's
    distance in the scalar offset) write the same values to the same addresses as the pointer stores of the generic tiles on every image, not only
    on the small cases above.  GPU against GPU: no oracle at this size."""
    n, sz = 32, 640
    x = rng_uniform(830, (n, sz, sz, 3), 0, 1)
    w0, b0 = rng_uniform(831, (32, 3, 6, 6), -0.3, 0.3), rng_uniform(832, (32,), -0.5, 0.5)
    w1, b1 = h(rng_uniform(833, (64, 32, 3, 3), -0.3, 0.3)), rng_uniform(834, (64,), -0.5, 0.5)
    w2, b2 = h(rng_uniform(835, (64, 64, 1, 1), -0.3, 0.3)), rng_uniform(836, (64,), -0.5, 0.5)
    stem = hops.conv2d_f16(x, w0, b0, (2, 2), (2, 2), act1="silu")
    with hops.plan(f16_s2c32=0):
        want1 = hops.conv2d_f16(stem, w1, b1, (2, 2), (1, 1), act1="silu")       # generic tiles: pointer stores
    pair = hops.conv_stem_s2c32_f16(x, w0, b0, w1, b1)
    assert_exact(pair, want1, "stem pair, 32 x 640 x 640")
    assert_exact(hops.conv2d_f16(stem, w1, b1, (2, 2), (1, 1), act1="silu"), want1, "3x3 s2 patch kernel, 32 x 320 x 320 x 32")
    want2 = hops.conv2d_f16(want1, w2, b2, act1="silu")
    ya, yb = hops.conv_stem_s2c32_pw_f16(x, w0, b0, w1, b1, w2, b2, split_oc=32, out2_ld=64, out2_c_off=32)
    assert_exact(np.concatenate([ya, yb], -1), want2, "stem triple, 32 x 640 x 640")


@pytest.mark.parametrize("op", list(range(18)))
def test_unary_ops_with_fp16_storage(hops, orc, op):
    """si_hip_unary_f16 (round 5; UnaryOp of expand_expression.cpp:123-165 on fp16 tensors): the fp32 function of si_hip_unary_f32 on the
    widened value with ONE rounding on the way out -- so the result is the oracle's value for the same (half) input, rounded to fp16,
    up to the few-ulp freedom of the library functions (one fp16 ulp here); dense and strided."""
    from test_oracle import unary_input
    x = unary_input(op, (2, 7, 9, 24)).astype(np.float16)
    ref = orc.unary_op(op, x.astype(np.float32))
    for kw in ({}, {"in_ld": 40, "out_ld": 32}, {"in_ld": 27, "out_ld": 25}):
        got = hops.unary_op_f16(op, x, **kw).astype(np.float32)
        fin = np.isfinite(ref) & (np.abs(ref) < 6.0e4)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), "unary %d: NaN pattern" % op
        want = ref.astype(np.float16).astype(np.float32)
        ulp = np.maximum(np.abs(want) * 2.0 ** -10, 2.0 ** -24)
        assert np.all(np.abs(got[fin] - want[fin]) <= ulp[fin] + 1e-30), ("unary %d %s" % (op, kw), np.abs(got[fin] - want[fin]).max())


@pytest.mark.parametrize("n,levels,ne", [
    (3, ((20, 128), (10, 256), (5, 512)), 85),     # 400 / 100 / 25 pixels per image: whole tiles, a 16-pixel tail, a tile of 25
    (2, ((9, 256), (3, 128), (1, 512)), 85),       # 91 rows x 3: image bases only 4-byte aligned -> the dword form of the run
    (1, ((16, 512), (8, 128), (4, 256)), 85),
    (2, ((16, 128), (8, 256), (4, 512)), 25),      # another head (20 classes): whole tiles through the general decode
])
def test_detect_tile_kernel_same_bits_as_generic_tiles(hops, orc, gpu, n, levels, ne):
    """Round 4: a Detect level over 128 / 256 / 512 channels runs as detect_f16_tile_kernel (64 consecutive pixels x all 255 columns
    per workgroup, decoded rows staged through LDS, one contiguous run out).  Same MFMA sequence per element and the same decode
    expressions as the generic tiles, so the two forms agree BIT for bit; and the oracle's bar holds (src/layer/yolo_detect.cpp:223-266).
    Round 5: whole tiles of the 3 x 85 head take a decode of their own (sigmoid + one LDS write per element, the box arithmetic only in
    the three column blocks that hold box columns); every other shape the general one."""
    from simpleinfer_amd import _native
    H = _native.hip()
    na = 3
    feats, ws, bs, grids, anchors = [], [], [], [], []
    for i, (hh, c) in enumerate(levels):
        feats.append(h(rng_uniform(150 + i, (n, hh, hh, c), -1, 1)))
        ws.append(h(rng_uniform(160 + i, (na * ne, c, 1, 1), -0.3, 0.3)))
        bs.append(rng_uniform(170 + i, (na * ne,), -0.5, 0.5))
        gy, gx = np.meshgrid(np.arange(hh, dtype=np.float32), np.arange(hh, dtype=np.float32), indexing="ij")
        grids.append(np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, hh, hh, 2)).copy())
        anchors.append(np.broadcast_to(rng_uniform(180 + i, (1, na, 1, 1, 2), 5, 300), (1, na, hh, hh, 2)).copy())
    strides = [8.0, 16.0, 32.0]
    import ctypes as C
    d = _native.SiConv2dDesc(n, 20, 20, 128, 128, 20, 20, na * ne, na * ne, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, na * ne, 0, 0.0)
    lv = _native.SiYoloLevel(na, ne, 1200, 0, 8.0)
    off = _native.SiConvPlan(f16_detect_tile=0)
    d.plan = C.pointer(off)
    assert H.si_hip_conv2d_yolo_f16_tile(C.byref(d), C.byref(lv)) == 0
    with hops.plan(f16_detect_tile=0):
        base = hops.yolo_detect_f16(feats, ws, bs, grids, anchors, strides, na)
    d.plan = None
    assert H.si_hip_conv2d_yolo_f16_tile(C.byref(d), C.byref(lv)) == 1
    got = hops.yolo_detect_f16(feats, ws, bs, grids, anchors, strides, na)
    assert_exact(got, base, "Detect tile kernel vs generic tiles")
    assert_detect_parity(got, orc.yolo_detect(feats, ws, bs, grids, anchors, strides, na), 1e-4, 1e-4, what="Detect tile kernel")


@pytest.mark.parametrize("n,lh,lw,cl,cs,oc,scale,up_first", [
    (2, 10, 10, 128, 128, 128, (2.0, 2.0), True),    # the YOLOv5 PAN form, 64-wide K blocks
    (3, 5, 7, 64, 192, 96, (2.0, 2.0), False),       # upsampled tensor second, ragged column block
    (1, 4, 6, 32, 64, 64, (3.0, 2.0), True),         # 96 channels: 32-wide K blocks; non-square scale
    (2, 10, 10, 256, 256, 256, (2.0, 2.0), True),    # YOLOv5s conv_34's form (K = 512, 256 columns)
    (1, 7, 9, 128, 384, 160, (2.0, 2.0), False),     # ... upsampled tensor second, 160 columns, ragged M
])
def test_conv_f16_reads_upsampled_source(hops, orc, n, lh, lw, cl, cs, oc, scale, up_first):
    """si_hip_conv2d_upcat_f16 (round 4): the 1x1 conv behind cat(upsample(x), skip) reads x at the source pixel with the reference's
    index rule (src/layer/upsample.cpp:85-92) -- BIT exact versus si_hip_conv2d_f16 on the materialised concat, also in the
    sibling-split form."""
    oh, ow = int(lh * scale[0]), int(lw * scale[1])
    low, skip = h(rng_uniform(400, (n, lh, lw, cl), -1, 1)), h(rng_uniform(401, (n, oh, ow, cs), -1, 1))
    w, b = h(rng_uniform(402, (oc, cl + cs, 1, 1), -0.3, 0.3)), rng_uniform(403, (oc,), -0.5, 0.5)
    upo = orc.upsample_nearest(low, scale[0], scale[1], (oh, ow))
    cat = np.concatenate([upo, skip] if up_first else [skip, upo], axis=-1)
    got = hops.conv2d_upcat_f16(low, skip, w, b, scale, up_first, act1="silu")
    assert got.dtype == np.float16
    assert_exact(got, hops.conv2d_f16(cat, w, b, act1="silu"), "dual-source fp16 conv == conv over the materialised concat")
    assert_parity(got.astype(np.float32), orc.activation("silu", orc.conv2d(cat, w, b)), F16_TOL, what="vs the oracle's upsample + cat + conv")
    if oc >= 64:
        ya, yb = hops.conv2d_upcat_f16(low, skip, w, b, scale, up_first, act1="silu", split_oc=32)
        assert_exact(np.concatenate([ya, yb], -1), got, "sibling-split form")
    with hops.plan(f16_tile=0):   # (a forced tile and the policy agree: the dual-source form lives in the one-stage 64 x 64 kernel)
        assert_exact(hops.conv2d_upcat_f16(low, skip, w, b, scale, up_first, act1="silu"), got, "dual-source form: forced tile vs the policy's")
