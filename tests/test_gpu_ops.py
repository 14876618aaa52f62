"""GPU: per-operator parity of the HIP kernels (called through the C-ABI, include/si_hip.h) against the
CPU oracle and the committed golden vectors.  Mirrors the reference's layer-test matrix
(test/test_layer/*.cpp) with fixed seeds, and adds what it lacks: batch > 1,
ragged channel counts, strided (concat-slice) tensors, fused epilogues.

Tolerance: fp32 kernels 1e-4 relative to max|ref| (util.REL_TOL); index / copy / max ops bit-exact.
"""
import os

import numpy as np
import pytest

from util import assert_detect_parity, assert_exact, assert_parity, rng_uniform

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops_golden.npz"))
CONV_CASES = ["conv_3x3_s1_p1", "conv_3x3_s1_p0", "conv_3x3_s2_p1", "conv_6x6_s2_p2", "conv_1x1", "conv_grouped",
              "conv_depthwise", "conv_dilated", "conv_7x7_s2_p3"]


@pytest.fixture(scope="module")
def hops(gpu):
    from simpleinfer_amd import hipops
    return hipops


@pytest.mark.parametrize("name", CONV_CASES)
def test_conv_golden(hops, name):
    s, p, d, g = (int(v) for v in GOLD[name + "/cfg"])
    b = GOLD[name + "/b"] if (name + "/b") in GOLD.files else None
    got = hops.conv2d(GOLD[name + "/x"], GOLD[name + "/w"], b, (s, s), (p, p), (d, d), g)
    assert_parity(got, GOLD[name + "/y"], what=name)


# (shape NHWC, oc, k, s, p, d, g): reference test_conv_2d.cpp conv0..conv3, then the YOLOv5s / ResNet18 layer
# families of SURVEY.md 8(a) at reduced spatial size, then ragged / degenerate shapes
CONV_SHAPES = [
    ((1, 128, 128, 32), 16, 3, 1, 1, 1, 1),     # test_conv_2d.cpp:8-132
    ((1, 128, 128, 32), 16, 3, 1, 1, 1, 2),     # :134-274 groups=2
    ((1, 160, 160, 3), 32, 6, 2, 2, 1, 1),      # :276-416 stem (1/4 size)
    ((1, 10, 10, 256), 255, 1, 1, 0, 1, 1),     # :418-558 Detect 1x1
    ((2, 40, 40, 64), 64, 3, 1, 1, 1, 1),       # C3 bottleneck 3x3
    ((2, 40, 40, 64), 128, 3, 2, 1, 1, 1),      # downsample 3x3 s2
    ((2, 20, 20, 512), 256, 1, 1, 0, 1, 1),     # wide 1x1
    ((2, 20, 20, 256), 512, 3, 2, 1, 1, 1),     # K = 2304
    ((3, 56, 56, 3), 64, 7, 2, 3, 1, 1),        # ResNet stem
    ((2, 14, 14, 128), 256, 1, 2, 0, 1, 1),     # ResNet 1x1 s2 downsample
    ((1, 7, 9, 5), 7, 3, 1, 1, 1, 1),           # ragged everything
    ((2, 5, 5, 6), 3, 3, 1, 0, 1, 1),
    ((1, 1, 1, 4), 4, 1, 1, 0, 1, 1),           # single pixel
    ((1, 3, 3, 1), 1, 3, 1, 1, 1, 1),           # single channel
    ((1, 9, 9, 8), 8, 3, 1, 2, 2, 1),           # dilation
    ((1, 12, 12, 12), 12, 3, 1, 1, 1, 12),      # depthwise
    ((1, 12, 12, 12), 24, 3, 2, 1, 1, 3),       # grouped, oc/g = 8
    ((5, 33, 17, 20), 36, 5, 3, 2, 1, 1),       # odd kernel/stride, M not a tile multiple
]


@pytest.mark.parametrize("shape,oc,k,s,p,d,g", CONV_SHAPES)
def test_conv_vs_oracle(hops, orc, shape, oc, k, s, p, d, g):
    seed = hash((shape, oc, k, s, p, d, g)) % 100000
    x = rng_uniform(seed, shape)
    w = rng_uniform(seed + 1, (oc, shape[3] // g, k, k), -0.5, 0.5)
    b = rng_uniform(seed + 2, (oc,), -0.5, 0.5)
    got = hops.conv2d(x, w, b, (s, s), (p, p), (d, d), g)
    # reference dispatch (Winograd when eligible) is the parity target; the fp64 loop bounds both
    assert_parity(got, orc.conv2d(x, w, b, (s, s), (p, p), (d, d), g, path="auto"), what="vs reference path")
    assert_parity(got, orc.conv2d(x, w, b, (s, s), (p, p), (d, d), g, path="naive"), 2e-5, what="vs fp64")


def test_conv_reference_test_tolerance(hops, orc):
    """the reference's own criterion: abs < 2e-4 against its float loop, inputs U[0,1) (test_conv_2d.cpp:100-131)"""
    x = rng_uniform(1, (1, 128, 128, 32))
    w = rng_uniform(2, (16, 32, 3, 3))
    b = rng_uniform(3, (16,))
    got = hops.conv2d(x, w, b, (1, 1), (1, 1))
    ref = orc.conv2d(x, w, b, (1, 1), (1, 1), path="naive", acc64=False)
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max() / 16)


@pytest.mark.parametrize("act", ["relu", "silu", "sigmoid", "hardsigmoid", "hardswish", "leakyrelu"])
def test_conv_fused_activation(hops, orc, act):
    x = rng_uniform(4, (2, 12, 12, 16), -1, 1)
    w = rng_uniform(5, (24, 16, 3, 3), -0.5, 0.5)
    b = rng_uniform(6, (24,), -0.5, 0.5)
    y = orc.conv2d(x, w, b, (1, 1), (1, 1), path="naive")
    ref = np.where(y > 0, y, np.float32(0.1) * y) if act == "leakyrelu" else orc.activation(act, y)
    assert_parity(hops.conv2d(x, w, b, (1, 1), (1, 1), act1=act, act_param=0.1), ref, what=act)


def test_conv_fused_residual_orders(hops, orc):
    x = rng_uniform(7, (2, 10, 10, 32), -1, 1)
    w = rng_uniform(8, (32, 32, 3, 3), -0.3, 0.3)
    b = rng_uniform(9, (32,), -0.5, 0.5)
    r = rng_uniform(10, (2, 10, 10, 32), -1, 1)
    y = orc.conv2d(x, w, b, (1, 1), (1, 1), path="naive")
    # YOLOv5 bottleneck: x + silu(conv)      ResNet basic block: relu(conv + identity)
    assert_parity(hops.conv2d(x, w, b, (1, 1), (1, 1), act1="silu", residual=r), orc.activation("silu", y) + r)
    assert_parity(hops.conv2d(x, w, b, (1, 1), (1, 1), residual=r, act2="relu"), orc.activation("relu", y + r))


@pytest.mark.parametrize("ic,oc", [(24, 72), (40, 24), (72, 16), (8, 40), (144, 48), (36, 64)])
def test_conv_pointwise_padded_k(hops, orc, ic, oc):
    """1x1 convs whose channel count is a multiple of 4 but not of 32 (MobileNet's pointwise / squeeze-excite convs): the
    weights are zero-padded to whole 32-channel blocks and the fast kernel masks the tail vectors.  The input is a NaN-filled
    wider row here, so a vector read past the last channel would poison the result."""
    x = rng_uniform(40 + ic, (2, 9, 7, ic), -1, 1)
    w = rng_uniform(41 + ic, (oc, ic, 1, 1), -0.5, 0.5)
    b = rng_uniform(42 + ic, (oc,), -0.5, 0.5)
    ref = orc.conv2d(x, w, b, (1, 1), (0, 0), path="naive")
    assert_parity(hops.conv2d(x, w, b), ref, what="dense")
    assert_parity(hops.conv2d(x, w, b, act1="hardswish"), orc.activation("hardswish", ref), what="hardswish")
    assert_parity(hops.conv2d(x, w, b, in_ld=ic + 12, out_ld=oc + 8, out_c_off=4, in_fill=np.nan), ref, what="strided, NaN beyond the channels")
    name = hops.conv2d_kernel_name(x.shape, w.shape)
    assert "fast" in name, name


def test_conv_strided_tensors(hops, orc):
    """input read from / output written into a wider concat buffer (pixel stride > channels)"""
    x = rng_uniform(11, (2, 9, 9, 16), -1, 1)
    w = rng_uniform(12, (8, 16, 1, 1), -0.5, 0.5)
    b = rng_uniform(13, (8,), -0.5, 0.5)
    ref = orc.conv2d(x, w, b)
    assert_parity(hops.conv2d(x, w, b, in_ld=40), ref, what="in_ld")
    assert_parity(hops.conv2d(x, w, b, out_ld=24, out_c_off=12), ref, what="out_ld + offset")
    assert_parity(hops.conv2d(x, w, b, in_ld=18, out_ld=11, out_c_off=3), ref, what="unaligned strides")


# (n, h, w, ic, oc, pad): YOLOv5s / ResNet18 3x3 s1 families, odd sizes (clipped 2x2 stores), every tile-block shape
# (4x8, 8x4, 16x2 tiles), blocks spanning images, oc = 32 (one MFMA column tile) and 96 (ragged column block)
WINO_SHAPES = [(2, 40, 40, 64, 64, 1), (1, 80, 80, 32, 32, 1), (3, 20, 20, 128, 128, 1), (5, 10, 10, 256, 64, 1),
               (2, 13, 17, 16, 32, 1), (2, 12, 12, 32, 96, 0), (1, 7, 5, 16, 32, 1), (4, 4, 4, 32, 64, 1),
               (1, 56, 56, 64, 64, 1)]


@pytest.mark.parametrize("n,h,w,ic,oc,pad", WINO_SHAPES)
def test_winograd_vs_reference_path(hops, orc, n, h, w, ic, oc, pad):
    x = rng_uniform(n * 1000 + h, (n, h, w, ic), -1, 1)
    wt = rng_uniform(n * 1000 + h + 1, (oc, ic, 3, 3), -0.5, 0.5)
    b = rng_uniform(n * 1000 + h + 2, (oc,), -0.5, 0.5)
    got = hops.conv2d_winograd(x, wt, b, (pad, pad))
    # parity target: the reference's own Winograd F(2,3) pipeline (oracle restatement), bounded by the fp64 loop
    assert_parity(got, orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="winograd"), what="vs reference Winograd")
    assert_parity(got, orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="naive"), 2e-5, what="vs fp64")
    # the reference test's own criterion (test/test_layer/test_winograd.cpp:130): abs 2e-3
    assert np.abs(got - orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="naive", acc64=False)).max() < 2e-3


@pytest.mark.parametrize("n,h,w,ic,oc,pad", WINO_SHAPES)
def test_winograd_split_vs_reference_path(hops, orc, n, h, w, ic, oc, pad):
    """si_hip_conv2d_wino23_split_f32 (round 5, engine option f32_split): the fused Winograd kernel with its plane GEMMs on the fp16
    matrix cores from three fp16 products per fp32 product -- held to the SAME bars as the fp32 Winograd kernel: the oracle's restated
    reference pipeline (src/layer/conv_2d.cpp:382-487) at 1e-4, the float64 convolution at 2e-5, the reference test's own 2e-3; the
    fused epilogue; and an image's bits do not depend on its batch position."""
    x = rng_uniform(n * 1000 + h, (n, h, w, ic), -1, 1)
    wt = rng_uniform(n * 1000 + h + 1, (oc, ic, 3, 3), -0.5, 0.5)
    b = rng_uniform(n * 1000 + h + 2, (oc,), -0.5, 0.5)
    got = hops.conv2d_wino23_split(x, wt, b, (pad, pad))
    assert_parity(got, orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="winograd"), what="vs reference Winograd")
    naive = orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="naive")
    assert_parity(got, naive, 2e-5, what="vs fp64")
    assert np.abs(got - orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="naive", acc64=False)).max() < 2e-3
    e_split = np.abs(got - naive).max() / np.abs(naive).max()
    e_f32 = np.abs(hops.conv2d_winograd(x, wt, b, (pad, pad)) - naive).max() / np.abs(naive).max()
    assert e_split <= 2.0 * e_f32 + 1e-7, (e_split, e_f32)
    r = rng_uniform(n * 1000 + h + 3, got.shape, -1, 1)
    full = hops.conv2d_wino23_split(x, wt, b, (pad, pad), act1="silu", residual=r)
    assert_parity(full, orc.activation("silu", naive) + r, what="silu + residual")
    assert_exact(hops.conv2d_wino23_split(x[n - 1:], wt, b, (pad, pad), act1="silu", residual=r[n - 1:])[0], full[n - 1], "batch position")


def test_winograd_fused_epilogue_and_strides(hops, orc):
    x = rng_uniform(71, (2, 20, 20, 32), -1, 1)
    wt = rng_uniform(72, (64, 32, 3, 3), -0.3, 0.3)
    b = rng_uniform(73, (64,), -0.5, 0.5)
    r = rng_uniform(74, (2, 20, 20, 64), -1, 1)
    y = orc.conv2d(x, wt, b, (1, 1), (1, 1), path="naive")
    assert_parity(hops.conv2d_winograd(x, wt, b, act1="silu", residual=r), orc.activation("silu", y) + r, what="silu + residual")
    assert_parity(hops.conv2d_winograd(x, wt, b, residual=r, act2="relu"), orc.activation("relu", y + r), what="residual + relu")
    assert_parity(hops.conv2d_winograd(x, wt, b, in_ld=48, out_ld=160, out_c_off=32), y, what="strided tensors")
    # batch invariance: bit-identical per image
    full = hops.conv2d_winograd(x, wt, b, act1="silu")
    assert_exact(hops.conv2d_winograd(x[1:2], wt, b, act1="silu")[0], full[1])
    with pytest.raises(hops.HipError):
        hops.conv2d_winograd(rng_uniform(75, (1, 8, 8, 12), -1, 1), rng_uniform(76, (32, 12, 3, 3)), None)  # ic % 16 != 0


@pytest.mark.parametrize("n,h,w,ic,oc,pad", [(2, 40, 40, 64, 64, 1), (1, 80, 80, 32, 32, 1), (3, 20, 20, 128, 128, 1), (5, 10, 10, 256, 64, 1),
                                             (2, 12, 12, 32, 96, 0), (4, 4, 4, 32, 64, 1), (1, 56, 56, 64, 64, 1), (2, 13, 17, 96, 32, 1),
                                             (3, 7, 5, 32, 32, 1)])
def test_winograd_half_size_units_same_bits(hops, gpu, n, h, w, ic, oc, pad):
    """The kernel's 16-tile form on v_mfma_f32_16x16x4_f32 (half-size work units for launches that do not fill the chip evenly,
    DESIGN.md 3g) against its 32-tile form on v_mfma_f32_32x32x2_f32: the same transforms and the same ascending channel order
    in one fma chain per output, hence the SAME BITS -- every tile-block shape (2x8 / 4x4 / 8x2 tiles of 16), odd sizes, blocks
    spanning images, ragged oc blocks, fused epilogues and strided tensors."""
    x = rng_uniform(n * 1000 + h, (n, h, w, ic), -1, 1)
    wt = rng_uniform(n * 1000 + h + 1, (oc, ic, 3, 3), -0.5, 0.5)
    b = rng_uniform(n * 1000 + h + 2, (oc,), -0.5, 0.5)
    r = rng_uniform(n * 1000 + h + 3, (n, h + 2 * pad - 2, w + 2 * pad - 2, oc), -1, 1)
    outs = {}
    for form in (32, 16):
        with hops.plan(wino23_form=form):
            outs[form] = [hops.conv2d_winograd(x, wt, b, (pad, pad)),
                          hops.conv2d_winograd(x, wt, b, (pad, pad), act1="silu", residual=r),
                          hops.conv2d_winograd(x, wt, b, (pad, pad), residual=r, act2="relu", in_ld=ic + 16, out_ld=oc + 32, out_c_off=16)]
    for a, c, what in zip(outs[32], outs[16], ("plain", "silu + residual", "residual + relu, strided")):
        assert_exact(c, a, "16-tile form vs 32-tile form: " + what)


def test_winograd_64_channel_workgroups_and_unaligned_outputs(hops, orc):
    """The 64-output-channel form of the kernel (two accumulator groups per wave; picked by itself only for ic >= 256 on large
    grids) is forced onto every eligible shape of the two tests above (SiConvPlan::wino23_ocg = 2 in the calls' descriptors; until round 6
    a process-wide environment switch and a child process): same parity bars, same bit-exact batch invariance, the strided /
    offset-output cases included."""
    with hops.plan(wino23_ocg=2):
        for shape in WINO_SHAPES:
            test_winograd_vs_reference_path(hops, orc, *shape)
        test_winograd_fused_epilogue_and_strides(hops, orc)


# F(4x4, 3x3): same shapes (tile-block shapes 2x8 / 4x4 / 8x2 / 16x1, clipped 4x4 stores on odd sizes, image-spanning
# blocks) plus maps smaller than one tile.  Parity target stays the reference's Winograd pipeline; the fp64 bound shows
# the larger tile costs about one digit (measured <= 1e-5 of the output scale) and stays far inside 1e-4.
@pytest.mark.parametrize("n,h,w,ic,oc,pad", WINO_SHAPES + [(2, 3, 3, 16, 32, 1), (1, 28, 28, 128, 128, 1), (3, 14, 14, 256, 32, 1)])
def test_winograd43_vs_reference_path(hops, orc, n, h, w, ic, oc, pad):
    x = rng_uniform(n * 1000 + h, (n, h, w, ic), -1, 1)
    wt = rng_uniform(n * 1000 + h + 1, (oc, ic, 3, 3), -0.5, 0.5)
    b = rng_uniform(n * 1000 + h + 2, (oc,), -0.5, 0.5)
    got = hops.conv2d_winograd(x, wt, b, (pad, pad), tile=4)
    assert_parity(got, orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="winograd"), what="vs reference Winograd")
    assert_parity(got, orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="naive"), 2e-5, what="vs fp64")
    assert np.abs(got - orc.conv2d(x, wt, b, (1, 1), (pad, pad), path="naive", acc64=False)).max() < 2e-3


def test_winograd43_fused_epilogue_and_strides(hops, orc):
    x = rng_uniform(71, (2, 20, 20, 32), -1, 1)
    wt = rng_uniform(72, (64, 32, 3, 3), -0.3, 0.3)
    b = rng_uniform(73, (64,), -0.5, 0.5)
    r = rng_uniform(74, (2, 20, 20, 64), -1, 1)
    y = orc.conv2d(x, wt, b, (1, 1), (1, 1), path="naive")
    w4 = lambda *a, **k: hops.conv2d_winograd(*a, tile=4, **k)  # noqa: E731
    assert_parity(w4(x, wt, b, act1="silu", residual=r), orc.activation("silu", y) + r, what="silu + residual")
    assert_parity(w4(x, wt, b, residual=r, act2="relu"), orc.activation("relu", y + r), what="residual + relu")
    assert_parity(w4(x, wt, b, in_ld=48, out_ld=160, out_c_off=32), y, what="strided tensors")
    full = w4(x, wt, b, act1="silu")
    assert_exact(w4(x[1:2], wt, b, act1="silu")[0], full[1])
    with pytest.raises(hops.HipError):
        w4(rng_uniform(75, (1, 8, 8, 12), -1, 1), rng_uniform(76, (32, 12, 3, 3)), None)  # ic % 16 != 0


# depthwise (groups == ic == oc): the dedicated HBM-bound kernel, conv_depthwise.hip.  MobileNet shapes (3x3 s1 / s2,
# 5x5 s2), dilation, channel counts that are not multiples of 4 (scalar path), batch > 1, odd sizes, fused epilogue.
@pytest.mark.parametrize("n,h,w,c,k,s,p,d", [
    (2, 28, 28, 32, 3, 1, 1, 1), (2, 29, 27, 64, 3, 2, 1, 1), (1, 14, 14, 96, 5, 2, 2, 1), (3, 10, 10, 24, 3, 1, 2, 2),
    (2, 9, 11, 10, 3, 1, 1, 1), (1, 7, 7, 4, 7, 1, 3, 1), (2, 5, 5, 3, 3, 1, 0, 1)])
def test_conv_depthwise_kernel(hops, orc, n, h, w, c, k, s, p, d):
    x = rng_uniform(h * 13 + c, (n, h, w, c), -1, 1)
    wt = rng_uniform(h * 13 + c + 1, (c, 1, k, k), -0.5, 0.5)
    b = rng_uniform(h * 13 + c + 2, (c,), -0.5, 0.5)
    ref = orc.conv2d(x, wt, b, (s, s), (p, p), (d, d), c, path="naive")
    assert_parity(hops.conv2d(x, wt, b, (s, s), (p, p), (d, d), c), ref, what="depthwise")
    r = rng_uniform(h * 13 + c + 3, ref.shape, -1, 1)
    got = hops.conv2d(x, wt, b, (s, s), (p, p), (d, d), c, act1="hardswish", residual=r, act2="relu")
    assert_parity(got, orc.activation("relu", orc.activation("hardswish", ref) + r), what="depthwise fused epilogue")
    if c % 4 == 0:
        assert_parity(hops.conv2d(x, wt, b, (s, s), (p, p), (d, d), c, in_ld=c + 8, out_ld=2 * c, out_c_off=c), ref, what="strided")


@pytest.mark.parametrize("n,size,oc,k,s,p", [(2, 64, 16, 3, 2, 1), (1, 33, 32, 3, 1, 1), (2, 40, 48, 3, 2, 1)])
def test_stem_3x3_rgb(hops, orc, n, size, oc, k, s, p):
    # MobileNet-style stems (3x3 over 3 channels) take the persistent row kernel too
    x = rng_uniform(300 + size, (n, size, size, 3), 0, 1)
    w = rng_uniform(301 + size, (oc, 3, k, k), -0.5, 0.5)
    b = rng_uniform(302 + size, (oc,), -0.5, 0.5)
    ref = orc.activation("hardswish", orc.conv2d(x, w, b, (s, s), (p, p), path="naive"))
    assert_parity(hops.conv2d(x, w, b, (s, s), (p, p), act1="hardswish"), ref, what="3x3x3 stem")


def test_conv_split_siblings(hops, orc):
    """YOLOv5 C3: cv1 and cv2 (both 1x1 + SiLU on the same x) as one launch with a split destination"""
    x = rng_uniform(60, (2, 20, 20, 64), -1, 1)
    wa, wb = rng_uniform(61, (32, 64, 1, 1), -0.5, 0.5), rng_uniform(62, (64, 64, 1, 1), -0.5, 0.5)
    ba, bb = rng_uniform(63, (32,), -0.5, 0.5), rng_uniform(64, (64,), -0.5, 0.5)
    ya, yb = hops.conv2d_split(x, wa, ba, wb, bb, act1="silu", out2_ld=160, out2_c_off=96)
    assert_parity(ya, orc.activation("silu", orc.conv2d(x, wa, ba)), what="first sibling")
    assert_parity(yb, orc.activation("silu", orc.conv2d(x, wb, bb)), what="second sibling (strided slice)")
    # identical bits to the two separate launches
    assert_exact(ya, hops.conv2d(x, wa, ba, act1="silu"))
    assert_exact(yb, hops.conv2d(x, wb, bb, act1="silu"))


def test_conv_batch_invariance_bit_exact(hops):
    """an image's result must not depend on the batch it travels in (the data-parallel sharding contract)"""
    x = rng_uniform(14, (6, 24, 24, 32), -1, 1)
    w = rng_uniform(15, (64, 32, 3, 3), -0.5, 0.5)
    b = rng_uniform(16, (64,), -0.5, 0.5)
    full = hops.conv2d(x, w, b, (1, 1), (1, 1), act1="silu")
    for i in (0, 3, 5):
        assert_exact(hops.conv2d(x[i:i + 1], w, b, (1, 1), (1, 1), act1="silu")[0], full[i])


def test_conv_linearity(hops):
    x1, x2 = rng_uniform(17, (1, 16, 16, 8), -1, 1), rng_uniform(18, (1, 16, 16, 8), -1, 1)
    w = rng_uniform(19, (16, 8, 3, 3), -0.5, 0.5)
    a = hops.conv2d(x1 + x2, w, None, (1, 1), (1, 1))
    assert_parity(a, hops.conv2d(x1, w, None, (1, 1), (1, 1)) + hops.conv2d(x2, w, None, (1, 1), (1, 1)), 1e-5)


def test_linear(hops, orc):
    assert_parity(hops.linear(GOLD["linear/x"], GOLD["linear/w"], GOLD["linear/b"]), GOLD["linear/y"])
    x, w, b = rng_uniform(20, (1, 128)), rng_uniform(21, (64, 128)), rng_uniform(22, (64,))  # test_linear.cpp:8-65
    assert np.abs(hops.linear(x, w, b) - orc.linear(x, w, b)).max() < 1e-4
    x, w = rng_uniform(23, (64, 512), -1, 1), rng_uniform(24, (1000, 512), -0.1, 0.1)       # ResNet18 head
    assert_parity(hops.linear(x, w, None), orc.linear(x, w, None))


@pytest.mark.parametrize("n,h,w,c", [(2, 20, 20, 64), (3, 13, 7, 24), (1, 5, 5, 8), (2, 40, 40, 4)])
def test_maxpool5_chain3_exact(hops, orc, n, h, w, c):
    """SPPF's three chained 5x5 s1 p2 pools in one launch == the reference's MaxPool2d applied three times."""
    x = rng_uniform(70 + h, (n, h, w, c), -3, 3)
    want, cur = [], x
    for _ in range(3):
        cur = orc.maxpool2d(cur, (5, 5), (1, 1), (2, 2))
        want.append(cur)
    got = hops.maxpool5_chain3(x)
    for k in range(3):
        assert_exact(got[k], want[k], "stage %d" % k)
    # outputs as channel slices of a wider concat row (what the engine's aliasing hands over)
    got = hops.maxpool5_chain3(x, out_ld=4 * c, out_c_off=(c, 2 * c, 3 * c))
    for k in range(3):
        assert_exact(got[k], want[k], "strided stage %d" % k)
    with pytest.raises(hops.HipError):
        hops.maxpool5_chain3(rng_uniform(1, (1, 80, 80, 8), -1, 1))      # two map planes do not fit 64 KB of LDS
    with pytest.raises(hops.HipError):
        hops.maxpool5_chain3(rng_uniform(1, (1, 8, 8, 6), -1, 1))        # channels not a multiple of 4


def test_maxpool_exact(hops, orc):
    assert_exact(hops.maxpool2d(GOLD["maxpool_k5s1p2/x"], (5, 5), (1, 1), (2, 2)), GOLD["maxpool_k5s1p2/y"])
    assert_exact(hops.maxpool2d(GOLD["maxpool_k3s2p1/x"], (3, 3), (2, 2), (1, 1)), GOLD["maxpool_k3s2p1/y"])
    x = rng_uniform(25, (1, 8, 8, 3), -1, 1)                                      # test_max_pool_2d.cpp:7-73
    assert_exact(hops.maxpool2d(x, (2, 2), (2, 2), (0, 0)), orc.maxpool2d(x, (2, 2), (2, 2), (0, 0)))
    x = rng_uniform(26, (8, 20, 20, 256), -1, 1)                                  # :75-150 SPPF
    assert_exact(hops.maxpool2d(x, (5, 5), (1, 1), (2, 2)), orc.maxpool2d(x, (5, 5), (1, 1), (2, 2)))
    x = -rng_uniform(27, (1, 6, 6, 4), 1, 2)                                      # all-negative: padding must be lowest()
    assert_exact(hops.maxpool2d(x, (3, 3), (1, 1), (1, 1)), orc.maxpool2d(x, (3, 3), (1, 1), (1, 1)))


def test_avgpool(hops, orc):
    assert_parity(hops.adaptive_avgpool2d(GOLD["gap/x"], (1, 1)), GOLD["gap/y"], 1e-6)
    x = rng_uniform(28, (1, 8, 8, 3))                                             # test_adaptive_avg_pool_2d.cpp:7-52
    assert np.abs(hops.adaptive_avgpool2d(x, (1, 1)) - orc.adaptive_avgpool2d(x, (1, 1))).max() < 1e-6
    x = rng_uniform(29, (2, 12, 8, 5))
    assert_parity(hops.adaptive_avgpool2d(x, (3, 4)), orc.adaptive_avgpool2d(x, (3, 4)), 1e-6)
    with pytest.raises(hops.HipError):
        hops.adaptive_avgpool2d(x, (5, 3))  # not divisible: kUnsupport in the reference (:78-84)
    # global pooling over large maps (squeeze-excite / classifier heads): the workgroup-reduction kernel, any channel count
    for shape in [(3, 56, 56, 16), (2, 7, 7, 100), (2, 28, 28, 1), (1, 14, 14, 576)]:
        x = rng_uniform(40 + shape[3], shape, -1, 1)
        assert_parity(hops.adaptive_avgpool2d(x, (1, 1)), orc.adaptive_avgpool2d(x, (1, 1)), 1e-5, what="global avgpool %s" % (shape,))
    x = rng_uniform(45, (2, 28, 28, 64), -1, 1)
    assert_exact(hops.adaptive_avgpool2d(x[1:2], (1, 1))[0], hops.adaptive_avgpool2d(x, (1, 1))[1], "batch invariance")


def test_upsample_exact(hops, orc):
    assert_exact(hops.upsample_nearest(GOLD["upsample2/x"], 2.0, 2.0), GOLD["upsample2/y"])
    x = rng_uniform(30, (1, 16, 16, 3))                                           # test_upsample.cpp:8-54
    assert_exact(hops.upsample_nearest(x, 2.0, 2.0), orc.upsample_nearest(x, 2.0, 2.0))
    x = rng_uniform(31, (4, 10, 10, 128))                                         # :56-102
    assert_exact(hops.upsample_nearest(x, 2.0, 2.0), orc.upsample_nearest(x, 2.0, 2.0))
    x = rng_uniform(32, (1, 5, 7, 2))
    assert_exact(hops.upsample_nearest(x, 1.5, 2.5, (7, 17)), orc.upsample_nearest(x, 1.5, 2.5, (7, 17)))


def test_cat_exact(hops, orc):
    xs = [GOLD["cat/x0"], GOLD["cat/x1"], GOLD["cat/x2"]]                         # test_cat.cpp:7-65 (C = 3, 2, 4)
    assert_exact(hops.cat(xs, 3), GOLD["cat/y"])
    a, b = rng_uniform(33, (2, 3, 4, 8)), rng_uniform(34, (2, 5, 4, 8))
    assert_exact(hops.cat([a, b], 1), orc.cat([a, b], 1))
    a, b = rng_uniform(35, (2, 3, 4, 8)), rng_uniform(36, (2, 3, 6, 8))
    assert_exact(hops.cat([a, b], 2), orc.cat([a, b], 2))
    a, b = rng_uniform(37, (20, 20, 20, 256)), rng_uniform(38, (20, 20, 20, 256))
    assert_exact(hops.cat([a, b], 3), np.concatenate([a, b], 3))


def test_activations(hops, orc):
    x = GOLD["act/x"]
    for kind in ("silu", "relu", "sigmoid", "hardsigmoid", "hardswish"):
        assert_parity(hops.activation(kind, x), GOLD["act/" + kind], what=kind)
    x = rng_uniform(39, (1, 128, 128, 3), -4, 4)                                  # test_silu.cpp etc: abs 1e-6
    for kind in ("silu", "sigmoid", "hardsigmoid", "hardswish"):
        assert np.abs(hops.activation(kind, x) - orc.activation(kind, x)).max() < 2e-6, kind
    assert_exact(hops.activation("relu", x), orc.activation("relu", x))
    big = np.array([[-100.0, -20.0, 0.0, 20.0, 100.0, 1e-30, -1e-30, 88.0]], np.float32).reshape(1, 1, 2, 4)
    assert_parity(hops.activation("silu", big), orc.activation("silu", big), 1e-6)


def test_binary(hops, orc):
    a, b = rng_uniform(40, (1, 128, 128, 3)), rng_uniform(41, (1, 128, 128, 3))    # test_binary_op.cpp:7-87
    assert_exact(hops.binary_op(0, a, b), a + b)
    assert_exact(hops.binary_op(2, a, b), a * b)
    assert_exact(hops.binary_op(0, GOLD["binary/a"], GOLD["binary/b"]), GOLD["binary/add"])
    assert_exact(hops.binary_op(2, GOLD["binary/a"], GOLD["binary/b"]), GOLD["binary/mul"])
    a, b = rng_uniform(42, (2, 6, 6, 16)), rng_uniform(43, (2, 1, 1, 16))          # SE-style broadcast
    assert_exact(hops.binary_op(2, a, b), orc.binary_op(2, a, b))
    assert_exact(hops.binary_op(2, b, a), orc.binary_op(2, b, a))                   # channel vector on the left
    assert_exact(hops.binary_op(0, a, b), orc.binary_op(0, a, b))
    g = rng_uniform(46, (1, 1, 1, 16))                                             # one vector for every image
    assert_exact(hops.binary_op(2, a, g), orc.binary_op(2, a, g))
    a6, b6 = rng_uniform(47, (3, 5, 7, 6)), rng_uniform(48, (3, 1, 1, 6))          # c % 4 != 0: the general kernel
    assert_exact(hops.binary_op(2, a6, b6), orc.binary_op(2, a6, b6))
    a, b = rng_uniform(44, (1, 3, 1, 4)), rng_uniform(45, (2, 3, 5, 1))            # both sides broadcast
    assert_exact(hops.binary_op(0, a, b), orc.binary_op(0, a, b))
    with pytest.raises(hops.HipError):
        hops.binary_op(4, a, a)  # max / min: codes the loader never emits (expand_expression.cpp:198-203)


@pytest.mark.parametrize("op", range(18))
def test_unary_ops(hops, orc, op):
    """UnaryOp (expand_expression.cpp:123-165; no reference layer): the arithmetic codes bit-exact against the oracle (IEEE sqrt
    and division on both sides), the library functions within 4 ulp element by element; also through pixel strides."""
    from test_oracle import assert_ulp, unary_input
    x = unary_input(op, (3, 9, 11, 20))
    ref = orc.unary_op(op, x)
    for kw in ({}, {"in_ld": 28, "out_ld": 24}, {"in_ld": 21, "out_ld": 23}):
        got = hops.unary_op(op, x, **kw)
        if op in (0, 1, 2, 3, 4, 5, 6, 15):
            assert_exact(got, ref, "unary %d %s" % (op, kw))
        else:
            assert_ulp(got, ref, 4, "unary %d %s" % (op, kw))


def test_binary_all_codes_and_scalar_forms(hops, orc):
    """BinaryOp codes the loader can emit beyond the reference layer's add / mul (expand_expression.cpp:198-244): two tensors,
    tensor (op) literal, literal (op) tensor, and the squeeze-excite style broadcast with a non-commutative operator."""
    from test_oracle import assert_ulp
    a, b = rng_uniform(1, (2, 6, 5, 16), 0.5, 3.0), rng_uniform(2, (2, 6, 5, 16), 0.5, 3.0)
    for op in (0, 1, 2, 3, 7, 8):
        assert_exact(hops.binary_op(op, a, b), orc.binary_op(op, a, b), "binary %d" % op)
        assert_exact(hops.binary_scalar(op, a, 1.75), orc.binary_scalar(op, a, 1.75), "scalar %d" % op)
    for op in (6, 9, 10, 11):
        assert_ulp(hops.binary_op(op, a, b - 1.2), orc.binary_op(op, a, b - np.float32(1.2)), 4, "binary %d" % op)
        assert_ulp(hops.binary_scalar(op, a, 2.5), orc.binary_scalar(op, a, 2.5), 4, "scalar %d" % op)
    v = rng_uniform(3, (2, 1, 1, 16), 0.5, 2.0)
    assert_exact(hops.binary_op(3, a, v), orc.binary_op(3, a, v), "x / se")
    assert_exact(hops.binary_op(1, v, a, a.shape), orc.binary_op(1, v, a, a.shape), "se - x (the vector is the first operand)")
    odd = rng_uniform(4, (1, 3, 5, 7), 0.5, 2.0)     # channel count without a 16-byte vector path
    assert_exact(hops.binary_scalar(8, odd, 3.0), orc.binary_scalar(8, odd, 3.0), "3.0 / x, 7 channels")


def test_batchnorm_flatten(hops, orc):
    g = GOLD
    assert_parity(hops.batchnorm2d(g["bn/x"], g["bn/mean"], g["bn/var"], g["bn/gamma"], g["bn/beta"], 1e-5), g["bn/y"], 1e-5)
    assert_exact(hops.flatten_nhwc(g["flatten/x"]), g["flatten/y"])
    x = rng_uniform(46, (1, 2, 2, 128))                                            # test_flatten.cpp:7-45
    assert_exact(hops.flatten_nhwc(x), orc.flatten_nhwc(x))


@pytest.mark.parametrize("n,levels", [(2, ((8, 32), (4, 64), (2, 128))), (5, ((4, 32), (2, 64), (1, 96)))])
def test_yolo_detect_head(hops, orc, n, levels):
    # the second case has maps smaller than a 32-row MFMA tile and batch > 2: one tile spans several images
    na, ne = 3, 85
    feats, ws, bs, grids, anchors = [], [], [], [], []
    for i, (h, c) in enumerate(levels):
        feats.append(rng_uniform(50 + i, (n, h, h, c), -1, 1))
        ws.append(rng_uniform(60 + i, (na * ne, c, 1, 1), -0.3, 0.3))
        bs.append(rng_uniform(70 + i, (na * ne,), -0.5, 0.5))
        gy, gx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="ij")
        grids.append(np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, h, h, 2)).copy())
        anchors.append(np.broadcast_to(rng_uniform(80 + i, (1, na, 1, 1, 2), 5, 300), (1, na, h, h, 2)).copy())
    strides = [8.0, 16.0, 32.0]
    ref = orc.yolo_detect(feats, ws, bs, grids, anchors, strides, na)
    # per column group: a whole-tensor scale (box sizes ~1e3) would leave the sigmoid scores unconstrained
    assert_detect_parity(hops.yolo_detect(feats, ws, bs, grids, anchors, strides, na), ref, what="conv + decode kernels")
    assert_detect_parity(hops.yolo_detect(feats, ws, bs, grids, anchors, strides, na, fused=True), ref, what="decode fused in conv epilogue")


@pytest.mark.parametrize("n,levels", [(2, ((16, 128), (8, 256), (4, 512))), (5, ((8, 64), (5, 128), (3, 192)))])
def test_yolo_detect_head_f32_split(hops, orc, n, levels):
    """si_hip_conv2d_split3_yolo_f32 (round 5, engine option f32_split): the Detect levels on the three-fp16-products arithmetic with the
    decode of src/layer/yolo_detect.cpp:223-266 in the epilogue -- the fp32 bars of test_yolo_detect_head.  Whole 64-pixel tiles
    inside one image (the straight-line decode) and, in the second case, 64 / 25 / 9 pixels per image: tiles across image borders."""
    na, ne = 3, 85
    feats, ws, bs, grids, anchors = [], [], [], [], []
    for i, (h, c) in enumerate(levels):
        feats.append(rng_uniform(250 + i, (n, h, h, c), -1, 1))
        ws.append(rng_uniform(260 + i, (na * ne, c, 1, 1), -0.3, 0.3))
        bs.append(rng_uniform(270 + i, (na * ne,), -0.5, 0.5))
        gy, gx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="ij")
        grids.append(np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, h, h, 2)).copy())
        anchors.append(np.broadcast_to(rng_uniform(280 + i, (1, na, 1, 1, 2), 5, 300), (1, na, h, h, 2)).copy())
    strides = [8.0, 16.0, 32.0]
    ref = orc.yolo_detect(feats, ws, bs, grids, anchors, strides, na)
    got = hops.yolo_detect_split3(feats, ws, bs, grids, anchors, strides, na)
    assert_detect_parity(got, ref, what="Detect on the f32_split arithmetic")
    # round 6: from four 64-pixel tiles per CU on, the 128- and 256-channel levels run as 64-pixel runs of the output (detect_split_tile_kernel;
    # SiConvPlan::split3_bm = -1 takes it at any size); a forced tile height takes the YOLO form of the generic split kernel -- the same k-steps on
    # the same two chains, the same decode expressions: the same bits
    with hops.plan(split3_bm=-1):
        assert_exact(hops.yolo_detect_split3(feats, ws, bs, grids, anchors, strides, na), got, "Detect tile kernel vs the generic kernel's YOLO form")


@pytest.mark.parametrize("n,h,c,na,ne", [(2, 24, 128, 3, 85), (3, 12, 256, 3, 85), (2, 10, 128, 3, 30), (1, 9, 256, 2, 40)])
def test_detect_split_tile_kernel_shapes(hops, orc, n, h, c, na, ne):
    """detect_split_tile_kernel alone: nine whole 64-pixel tiles per image; tiles across the 32-pixel halves' ends (144, 100, 81 pixels per image: a
    last tile of 16 / 36 / 17 pixels); heads that are not 3 x 85 (the general decode path: 90 and 80 columns) -- against the oracle
    (src/layer/yolo_detect.cpp:223-266) at the fp32 bars and bit for bit against the generic kernel's YOLO form; the guard word stays 0, and one
    feature value at 1e5 sets it."""
    feats = [rng_uniform(5000, (n, h, h, c), -1, 1)]
    ws = [rng_uniform(5001, (na * ne, c, 1, 1), -0.3, 0.3)]
    bs = [rng_uniform(5002, (na * ne,), -0.5, 0.5)]
    gy, gx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="ij")
    grids = [np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, h, h, 2)).copy()]
    anchors = [np.broadcast_to(rng_uniform(5003, (1, na, 1, 1, 2), 5, 300), (1, na, h, h, 2)).copy()]
    ref = orc.yolo_detect(feats, ws, bs, grids, anchors, [8.0], na)
    hot = [feats[0].copy()]
    hot[0][n - 1, h // 2, h // 3, 5] = 1.0e5
    with hops.plan(split3_bm=-1):   # (these launches are far below four tiles per CU: the policy alone would give them the generic kernel)
        got, flags = hops.yolo_detect_split3(feats, ws, bs, grids, anchors, [8.0], na, return_flags=True)
        assert list(hops.yolo_detect_split3(hot, ws, bs, grids, anchors, [8.0], na, return_flags=True)[1]) == [1]
    assert list(flags) == [0]
    assert_detect_parity(got, ref, what="Detect tile kernel, %d x %d x %d, %d x %d columns" % (h, h, c, na, ne))
    assert_exact(hops.yolo_detect_split3(feats, ws, bs, grids, anchors, [8.0], na), got, "tile kernel vs the YOLO form of the generic kernel (the policy's choice here)")
    with hops.plan(split3_bm=32):
        assert_exact(hops.yolo_detect_split3(feats, ws, bs, grids, anchors, [8.0], na), got, "tile kernel vs the YOLO form, 32-row tiles forced")


# ---- letterbox + detection post-processing (test/test_yolo/test_yolo.cpp:194-259, 337-428) ----
def test_letterbox_exact(hops, orc):
    assert hops.letterbox_geometry(1080, 810, 640, 640) == orc.letterbox_geometry(1080, 810, 640, 640)
    assert hops.letterbox_geometry(375, 500, 640, 640) == orc.letterbox_geometry(375, 500, 640, 640)
    r = np.random.Generator(np.random.Philox(5))
    hr, wr, sc, pt, pl = hops.letterbox_geometry(375, 500, 640, 640)
    img = r.integers(0, 256, (hr, wr, 3), dtype=np.uint8)
    assert_exact(hops.letterbox(img, 640, 640, pt, pl), orc.letterbox(img, 640, 640, pt, pl), "letterbox")
    # the batched form (frames of one camera, one launch) and its odd-size fallback: the same bits per image
    imgs = r.integers(0, 256, (3, hr, wr, 3), dtype=np.uint8)
    got = hops.letterbox_batch(imgs, 640, 640, pt, pl)
    for b in range(3):
        assert_exact(got[b], orc.letterbox(imgs[b], 640, 640, pt, pl), "letterbox batch image %d" % b)
    hr2, wr2, _, pt2, pl2 = hops.letterbox_geometry(30, 50, 63, 63)
    odd = r.integers(0, 256, (2, hr2, wr2, 3), dtype=np.uint8)
    got = hops.letterbox_batch(odd, 63, 63, pt2, pl2)
    for b in range(2):
        assert_exact(got[b], orc.letterbox(odd[b], 63, 63, pt2, pl2), "letterbox batch, odd size, image %d" % b)


@pytest.mark.parametrize("sh,sw,dh,dw", [(720, 1280, 360, 640), (1080, 810, 640, 480), (375, 500, 480, 640), (33, 47, 64, 20), (1, 5, 3, 9), (7, 1, 2, 4)])
def test_resize_bilinear_u8c3_exact(hops, orc, sh, sw, dh, dw):
    """si_hip_resize_bilinear_u8c3 (round 4, the cv::resize of PreProcess, test_yolo.cpp:213-216): EXACT against the numpy
    restatement of the published 8-bit INTER_LINEAR fixed-point algorithm (oracle/orc.py -- simpleocv's source is absent, so this is
    not pinned to the reference), down- and up-scaling, one-pixel edges; and within 1 LSB of float bilinear interpolation with
    half-pixel centres (torch.nn.functional.interpolate, align_corners=False) where torch is importable."""
    r = np.random.Generator(np.random.Philox(sh * 7 + sw))
    imgs = r.integers(0, 256, (2, sh, sw, 3), dtype=np.uint8)
    got = hops.resize_bilinear_u8c3(imgs, dh, dw)
    for b in range(2):
        assert_exact(got[b], orc.resize_bilinear_u8c3(imgs[b], dh, dw), "resize image %d" % b)
    try:
        import torch
    except Exception:  # noqa: BLE001
        return
    t = torch.from_numpy(imgs.astype(np.float32)).permute(0, 3, 1, 2)
    ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False).permute(0, 2, 3, 1).numpy()
    assert np.abs(got.astype(np.float32) - ref).max() <= 1.0 + 1e-3


def test_resize_letterbox_fused_equals_the_two_steps(hops, orc):
    """si_hip_resize_letterbox_batch_u8_f32: PreProcess whole (test_yolo.cpp:194-259) for a batch of camera frames in one launch
    == resize, then letterbox, bit for bit -- a 720x1280 frame into the 640x640 input, a portrait frame, an upscaled thumbnail."""
    r = np.random.Generator(np.random.Philox(77))
    for (sh, sw, H, W) in ((720, 1280, 640, 640), (1080, 810, 640, 640), (90, 120, 160, 160), (31, 17, 63, 63)):
        frames = r.integers(0, 256, (3, sh, sw, 3), dtype=np.uint8)
        hr, wr, sc, pt, pl = hops.letterbox_geometry(sh, sw, H, W)
        got = hops.resize_letterbox_batch(frames, H, W)
        for b in range(3):
            want = orc.letterbox(orc.resize_bilinear_u8c3(frames[b], hr, wr), H, W, pt, pl)
            assert_exact(got[b], want, "fused resize + letterbox, %dx%d image %d" % (sh, sw, b))


@pytest.mark.parametrize("n,rows,nc,thr,agnostic,hot", [
    (3, 25200, 80, 0.25, False, 0.03),     # the application's shape and thresholds
    (2, 25200, 80, 0.25, True, 0.03),
    (5, 1000, 3, 0.10, False, 0.30),       # rows not a multiple of the staging block, few classes, dense overlaps
    (2, 700, 20, -1.0, False, 0.20),       # every row survives the filter
    (2, 300, 20, 2.0, False, 0.20),        # nothing survives
    (1, 64, 1, 0.25, False, 0.50),
])
def test_yolo_postprocess_matches_oracle(hops, orc, n, rows, nc, thr, agnostic, hot):
    from util import synthetic_predictions
    pred = synthetic_predictions(100 + rows + nc, n, rows, nc=nc, n_gt=5, hot_frac=hot)
    adjust = np.array([[80, 0, 0.5925926, 810, 1080], [0, 80, 1.0, 640, 480], [10, 20, 1.7, 300, 200]] * 2,
                      np.float32)[:n]
    for adj in (None, adjust):
        got, gcnt = hops.yolo_postprocess(pred, thr, 0.45, agnostic, adj)
        ref, rcnt = orc.yolo_postprocess(pred, thr, 0.45, agnostic, adj)
        assert_exact(gcnt, rcnt, "counts")
        for b in range(n):
            assert_exact(got[b], ref[b], "image %d" % b)
    if thr == 0.25:
        assert all(0 < c < rows for c in rcnt)


def test_yolo_postprocess_cap_and_empty(hops, orc):
    from util import synthetic_predictions
    pred = synthetic_predictions(7, 2, 2000, nc=10, n_gt=8, hot_frac=0.2)
    ref, rcnt = orc.yolo_postprocess(pred, 0.25, 0.45)
    got, gcnt = hops.yolo_postprocess(pred, 0.25, 0.45, max_det=5)
    assert_exact(gcnt, rcnt, "counts report every pick even past the cap")
    for b in range(2):
        assert rcnt[b] > 5
        assert_exact(got[b], ref[b][:5], "first max_det picks")
    got, gcnt = hops.yolo_postprocess(np.zeros((2, 0, 85), np.float32), 0.25, 0.45)
    assert list(gcnt) == [0, 0] and all(len(g) == 0 for g in got)


def test_yolo_postprocess_equal_confidences(hops, orc):
    """Equal confidences: the reference's unstable quicksort leaves their relative order unspecified (test_yolo.cpp:28-66);
    the device breaks ties by element index.  Same boxes either way when the tied boxes do not suppress each other."""
    from util import synthetic_predictions
    pred = synthetic_predictions(21, 2, 600, nc=6, n_gt=3, hot_frac=0.1)
    pred[:, 100:140, 4] = 0.5            # a run of rows with identical objectness ...
    pred[:, 100:140, 5:] = 0.1
    pred[:, 100:140, 5] = 0.9            # ... and identical class score: 40 equal confidences of 0.45
    pred[:, 100:140, 0] = np.linspace(20, 620, 40)[None]   # far apart: no mutual suppression
    pred[:, 100:140, 1] = 320
    pred[:, 100:140, 2:4] = 8
    got, gcnt = hops.yolo_postprocess(pred, 0.25, 0.45)
    ref, rcnt = orc.yolo_postprocess(pred, 0.25, 0.45)
    assert_exact(gcnt, rcnt, "counts")
    canon = lambda a: a[np.lexsort(a.T[::-1])]  # noqa: E731
    for b in range(2):
        assert_exact(canon(got[b]), canon(ref[b]), "same boxes, image %d" % b)
        assert (np.diff(got[b][:, 4]) <= 0).all()
        tied = got[b][got[b][:, 4] == np.float32(0.5) * np.float32(0.9)]
        assert len(tied) == 40 and (np.diff(tied[:, 0]) > 0).all(), "ties come out in element order"


@pytest.mark.parametrize("n,ih,iw,oc,k,p,act,strided", [
    (2, 96, 640, 32, 6, 2, "silu", False),    # two 160-pixel column tiles (5-wave workgroup), runs of several rows
    (1, 50, 66, 32, 6, 2, "silu", False),     # odd sizes: ragged column tile, row width not a multiple of 4 floats (scalar loads)
    (3, 37, 224, 64, 7, 3, "relu", False),    # ResNet stem: 7x7, two channel tiles, K = 147 padded to 148, odd height
    (2, 64, 64, 16, 3, 1, "hardswish", True), # MobileNet stem: 3x3, 16 of 32 channel lanes live, strided output slice
    (1, 18, 20, 32, 6, 2, "none", False),     # smaller than one run / one column tile
])
def test_stem_rolling_window_kernel(hops, orc, n, ih, iw, oc, k, p, act, strided):
    """conv_stem_roll.hip: persistent workgroups walking down runs of output rows over an LDS ring of input rows, 16x16x4 MFMAs.
    Against the reference path's restatement and fp64; the kernel name is the rolling one; an image's result does not depend on
    its batch position (bit exact)."""
    x = rng_uniform(300 + k, (n, ih, iw, 3))
    w = rng_uniform(301 + k, (oc, 3, k, k), -0.3, 0.3)
    b = rng_uniform(302 + k, (oc,), -0.5, 0.5)
    kw = dict(out_ld=oc + 8, out_c_off=4) if strided else {}
    got = hops.conv2d(x, w, b, (2, 2), (p, p), act1=act, **kw)
    ref = orc.conv2d(x, w, b, (2, 2), (p, p), path="naive")
    ref = orc.activation(act, ref) if act != "none" else ref
    assert_parity(got, ref, 2e-5, what="rolling stem vs fp64")
    assert "conv_stem_roll" in hops.conv2d_kernel_name(x.shape, w.shape, (2, 2), (p, p))
    if n > 1:
        one = hops.conv2d(x[n - 1:n], w, b, (2, 2), (p, p), act1=act, **kw)
        assert_exact(one[0], got[n - 1], "stem: batch position")


@pytest.mark.parametrize("n,h,w,ic,oc", [(32, 80, 80, 64, 64), (32, 40, 40, 128, 128), (32, 20, 20, 256, 256), (32, 160, 160, 32, 32),
                                          (64, 14, 14, 256, 256), (64, 7, 7, 512, 512)])
def test_winograd_full_size_batch_position_invariance(hops, n, h, w, ic, oc):
    """The Winograd layers of YOLOv5s (batch 32) and ResNet18 (batch 64) at bench size: an image run alone gives the bits it has
    inside the batch (at 256 channels the batch takes the 64-channel workgroup form and the single image the 32-channel one),
    and two launches of the batch agree."""
    rng = np.random.default_rng(1)
    x = rng.random((n, h, w, ic), dtype=np.float32) - 0.5
    wt = (rng.random((oc, ic, 3, 3), dtype=np.float32) - 0.5) * 0.2
    b = rng.random(oc, dtype=np.float32) - 0.5
    full = hops.conv2d_winograd(x, wt, b, (1, 1), act1="silu")
    for i in (0, 1, n // 2, n - 1):
        assert_exact(hops.conv2d_winograd(x[i:i + 1], wt, b, (1, 1), act1="silu")[0], full[i], "image %d alone vs in the batch" % i)
    assert_exact(hops.conv2d_winograd(x, wt, b, (1, 1), act1="silu"), full, "run to run")


@pytest.mark.parametrize("n,h,w,ic,oc,k,s,p", [(32, 80, 80, 128, 128, 1, 1, 0), (32, 160, 160, 64, 64, 1, 1, 0), (32, 160, 160, 32, 32, 1, 1, 0),
                                                (32, 20, 20, 1024, 512, 1, 1, 0), (32, 80, 80, 128, 256, 3, 2, 1), (32, 320, 320, 32, 64, 3, 2, 1)])
def test_implicit_gemm_full_size_batch_position_invariance(hops, n, h, w, ic, oc, k, s, p):
    """The same for the implicit-GEMM kernel's pointwise and general instantiations at YOLOv5s batch-32 layer sizes."""
    rng = np.random.default_rng(2)
    x = rng.random((n, h, w, ic), dtype=np.float32) - 0.5
    wt = (rng.random((oc, ic, k, k), dtype=np.float32) - 0.5) * 0.2
    b = rng.random(oc, dtype=np.float32) - 0.5
    full = hops.conv2d(x, wt, b, (s, s), (p, p), act1="silu")
    for i in (0, 1, n // 2, n - 1):
        assert_exact(hops.conv2d(x[i:i + 1], wt, b, (s, s), (p, p), act1="silu")[0], full[i], "image %d alone vs in the batch" % i)
    assert_exact(hops.conv2d(x, wt, b, (s, s), (p, p), act1="silu"), full, "run to run")


def test_stem_full_size_batch_position_invariance(hops):
    """The YOLOv5s stem at its bench size (32 x 640 x 640 x 3): images 0, 1 and 31 of the batch are bit-identical to the same image
    run alone.  (Regression: 16-byte buffer stores with an SGPR offset read their data registers late; with the next tile's
    arithmetic right behind them 1216 of 3.3 M elements of image 31 were wrong -- and nothing at test sizes showed it.)"""
    rng = np.random.default_rng(0)
    x = rng.random((32, 640, 640, 3), dtype=np.float32)
    w = (rng.random((32, 3, 6, 6), dtype=np.float32) - 0.5) * 0.3
    b = rng.random(32, dtype=np.float32) - 0.5
    full = hops.conv2d(x, w, b, (2, 2), (2, 2), act1="silu")
    for i in (0, 1, 31):
        assert_exact(hops.conv2d(x[i:i + 1], w, b, (2, 2), (2, 2), act1="silu")[0], full[i], "stem image %d alone vs in the batch" % i)


@pytest.mark.parametrize("n,lh,lw,cl,cs,oc,scale,up_first", [
    (2, 10, 10, 64, 32, 64, (2.0, 2.0), True),      # the YOLOv5 PAN form: cat([upsample(x), skip])
    (3, 5, 7, 32, 96, 32, (2.0, 2.0), False),       # upsampled tensor second; <= 32 output channels (the 128x32 tile)
    (1, 4, 6, 32, 32, 96, (3.0, 2.0), True),        # non-square scale: the index rule with inv = 1/3
])
def test_conv_reads_upsampled_source(hops, orc, n, lh, lw, cl, cs, oc, scale, up_first):
    """si_hip_conv2d_upcat_f32 against the reference's three passes restated (Upsample::Forward, Cat::Forward, the 1x1 conv): BIT
    exact versus this library's own unfused kernels on the materialised concat, within the bar of the oracle."""
    oh, ow = int(lh * scale[0]), int(lw * scale[1])
    low, skip = rng_uniform(400, (n, lh, lw, cl), -1, 1), rng_uniform(401, (n, oh, ow, cs), -1, 1)
    w, b = rng_uniform(402, (oc, cl + cs, 1, 1), -0.3, 0.3), rng_uniform(403, (oc,), -0.5, 0.5)
    upo = orc.upsample_nearest(low, scale[0], scale[1], (oh, ow))
    cat = np.concatenate([upo, skip] if up_first else [skip, upo], axis=-1)
    got = hops.conv2d_upcat(low, skip, w, b, scale, up_first, act1="silu")
    assert_exact(got, hops.conv2d(cat, w, b, act1="silu"), "dual-source conv == conv over the materialised concat")
    assert_parity(got, orc.activation("silu", orc.conv2d(cat, w, b)), what="vs the oracle's upsample + cat + conv")
    if oc >= 64:
        ya, yb = hops.conv2d_upcat(low, skip, w, b, scale, up_first, act1="silu", split_oc=32)
        assert_exact(np.concatenate([ya, yb], -1), got, "sibling-split form")


@pytest.mark.parametrize("n,hw,ic,oc,k,s,p,g", [
    (2, 20, 32, 16, 3, 1, 1, 2),      # the reference's grouped test shape (test_conv_2d.cpp:134-274): 16 -> 8 per group
    (1, 14, 128, 128, 3, 1, 1, 32),   # ResNeXt-style: 4 channels per group, 8 groups per MFMA block
    (2, 9, 64, 96, 3, 2, 1, 8),       # 8 -> 12 per group, stride 2, odd size
    (1, 12, 48, 48, 1, 1, 0, 6),      # 8 per group but 6 groups: not a multiple of 4 groups per block -> the generic kernel
    (3, 7, 64, 32, 5, 1, 2, 4),       # 16 -> 8 per group, 5x5
])
def test_conv_general_groups(hops, orc, n, hw, ic, oc, k, s, p, g):
    """ForwardIm2ColWithGroup (conv_2d.cpp:285-380) for 1 < channels/group < 32: neighbouring groups merged into dense 32-channel
    blocks with a block-diagonal weight image, on the fast MFMA kernel."""
    x = rng_uniform(500 + g, (n, hw, hw, ic), -1, 1)
    w = rng_uniform(501 + g, (oc, ic // g, k, k), -0.3, 0.3)
    b = rng_uniform(502 + g, (oc,), -0.5, 0.5)
    ref = orc.conv2d(x, w, b, (s, s), (p, p), (1, 1), g, path="naive")
    got = hops.conv2d(x, w, b, (s, s), (p, p), (1, 1), g, act1="relu")
    assert_parity(got, orc.activation("relu", ref), 2e-5, what="grouped conv vs fp64")
    assert_parity(hops.conv2d(x, w, b, (s, s), (p, p), (1, 1), g), orc.conv2d(x, w, b, (s, s), (p, p), (1, 1), g, path="auto"), what="vs the reference path")
    name = hops.conv2d_kernel_name(x.shape, w.shape, (s, s), (p, p), g)
    assert ("fast" in name) == (g != 6), name


# ---- fp32 convolution from three fp16 MFMA products (round 5, csrc/hip/conv_split3.hip; engine option f32_split, opt-in) ----
@pytest.mark.parametrize("n,hw,ic,oc,k,s,act,res", [
    (2, 21, 128, 96, 3, 2, "silu", False),       # ragged M and oc, image borders
    (1, 16, 256, 128, 3, 1, "none", True),
    (3, 13, 64, 255, 1, 1, "relu", False),       # pointwise, ragged oc
    (2, 10, 192, 64, 5, 1, "silu", True),        # 25 taps, three channel blocks; <= 64 columns: the 2 x 2-wave tile
    (2, 23, 32, 64, 3, 2, "silu", False),        # 32-channel K-tiles (YOLOv5s conv_1's form), ragged M
    (1, 14, 96, 160, 3, 1, "relu", True),        # 32-channel K-tiles, 64 x 128 tile, ragged oc
    (3, 9, 160, 40, 1, 1, "none", False),        # pointwise over 32-channel K-tiles, 40 of 64 columns live
])
def test_conv_split3_vs_oracle_and_fp64(hops, orc, n, hw, ic, oc, k, s, act, res):
    """si_hip_conv2d_split3_f32 against the reference's arithmetic restated (Conv2d::ForwardIm2Col, src/layer/conv_2d.cpp:207-283) at
    the fp32 bars -- 1e-4 of the tensor's scale vs the oracle, 2e-5 vs the float64 convolution -- with the fused epilogue; and it is at
    least as close to float64 as the true-fp32 kernel on the same data (exact 22-bit products, fp32 accumulation)."""
    x = rng_uniform(800 + hw, (n, hw, hw, ic), -2, 2)
    w = rng_uniform(801, (oc, ic, k, k), -0.2, 0.2)
    b = rng_uniform(802, (oc,), -0.5, 0.5)
    p = k // 2
    oh = (hw + 2 * p - k) // s + 1
    r = rng_uniform(803, (n, oh, oh, oc), -1, 1) if res else None
    got = hops.conv2d_split3(x, w, b, (s, s), (p, p), act1=act, residual=r)
    ref = orc.conv2d(x, w, b, (s, s), (p, p), path="auto")
    ref = orc.activation(act, ref) if act != "none" else ref
    if res:
        ref = ref + r
    assert_parity(got, ref, what="split3 vs the reference path")
    plain = hops.conv2d_split3(x, w, b, (s, s), (p, p))
    naive = orc.conv2d(x, w, b, (s, s), (p, p), path="naive")
    e3 = np.abs(plain - naive).max() / np.abs(naive).max()
    e32 = np.abs(hops.conv2d(x, w, b, (s, s), (p, p)) - naive).max() / np.abs(naive).max()
    assert e3 <= 2e-5 and e3 <= 2.0 * e32 + 1e-7, (e3, e32)
    one = hops.conv2d_split3(x[n - 1:], w, b, (s, s), (p, p), act1=act, residual=None if r is None else r[n - 1:])
    assert_exact(one, got[n - 1:], "split3: batch position")


def test_conv_split3_siblings_and_launch_size_tiles(hops, orc):
    """Round 6: (a) two sibling 1x1 convs (YOLOv5 C3's cv1 | cv2) as ONE launch on the f32_split arithmetic with a split destination
    (si_hip_conv2d_split3_split_f32) -- each half bit-identical to its own si_hip_conv2d_split3_f32 launch, also into a slice of a wider tensor,
    and inside the fp32 bars; (b) the 32- / 64- / 128-row tiles of the kernel produce the same bits (the tile follows the launch size: small
    batches take 32-row tiles), so an image's result does not depend on the batch it rides in."""
    x = rng_uniform(4500, (3, 13, 17, 256), -2, 2)
    wa, wb = rng_uniform(4501, (128, 256, 1, 1), -0.2, 0.2), rng_uniform(4502, (160, 256, 1, 1), -0.2, 0.2)
    ba, bb = rng_uniform(4503, (128,), -0.5, 0.5), rng_uniform(4504, (160,), -0.5, 0.5)
    ya, yb = hops.conv2d_split3_split(x, wa, ba, wb, bb, act1="silu", out2_ld=224, out2_c_off=32)
    assert_exact(ya, hops.conv2d_split3(x, wa, ba, act1="silu"), "first sibling")
    assert_exact(yb, hops.conv2d_split3(x, wb, bb, act1="silu"), "second sibling, into a channel slice")
    assert_parity(yb, orc.activation("silu", orc.conv2d(x, wb, bb, (1, 1), (0, 0), path="naive")), what="split3 siblings vs fp64")
    w = rng_uniform(4505, (192, 128, 3, 3), -0.2, 0.2)
    xs = rng_uniform(4506, (2, 19, 11, 128), -2, 2)
    b = rng_uniform(4507, (192,), -0.5, 0.5)
    outs = {}
    for bm in (32, 64, 128):
        with hops.plan(split3_bm=bm):
            outs[bm] = hops.conv2d_split3(xs, w, b, (2, 2), (1, 1), act1="silu")
    assert_exact(outs[32], outs[64], "32-row vs 64-row tiles")
    assert_exact(outs[128], outs[64], "128-row vs 64-row tiles")
    assert_exact(hops.conv2d_split3(xs, w, b, (2, 2), (1, 1), act1="silu"), outs[64], "the policy's tile")
    # <= 64 output columns (the 2 x 2-wave forms, 64 / 128 rows; 32- and 64-channel K-tiles) and the Detect form (32 / 64 rows)
    for ic in (32, 64):
        xc = rng_uniform(4508 + ic, (2, 23, 21, ic), -2, 2)
        wc, bc = rng_uniform(4509 + ic, (64, ic, 3, 3), -0.2, 0.2), rng_uniform(4510 + ic, (64,), -0.5, 0.5)
        with hops.plan(split3_bm=128):
            y128 = hops.conv2d_split3(xc, wc, bc, (2, 2), (1, 1), act1="silu")
        assert_exact(hops.conv2d_split3(xc, wc, bc, (2, 2), (1, 1), act1="silu"), y128, "64 columns over %d channels: 64-row vs 128-row tiles" % ic)
    na, ne, n = 3, 85, 2
    feats, ws, bs, grids, anchors = [], [], [], [], []
    for i, (hh, c) in enumerate([(12, 128), (5, 256)]):
        feats.append(rng_uniform(4520 + i, (n, hh, hh, c), -1, 1))
        ws.append(rng_uniform(4530 + i, (na * ne, c, 1, 1), -0.3, 0.3))
        bs.append(rng_uniform(4540 + i, (na * ne,), -0.5, 0.5))
        gy, gx = np.meshgrid(np.arange(hh, dtype=np.float32), np.arange(hh, dtype=np.float32), indexing="ij")
        grids.append(np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, hh, hh, 2)).copy())
        anchors.append(np.broadcast_to(rng_uniform(4550 + i, (1, na, 1, 1, 2), 5, 300), (1, na, hh, hh, 2)).copy())
    with hops.plan(split3_bm=64):
        d64 = hops.yolo_detect_split3(feats, ws, bs, grids, anchors, [8.0, 16.0], na)
    assert_exact(hops.yolo_detect_split3(feats, ws, bs, grids, anchors, [8.0, 16.0], na), d64, "Detect on the split kernel: 32-row vs 64-row tiles")


@pytest.mark.parametrize("n,lh,lw,cl,cs,oc,scale,up_first,split", [
    (2, 10, 10, 128, 128, 128, (2.0, 2.0), True, 64),     # the YOLOv5 PAN form with the sibling split (conv_40's shape class)
    (3, 5, 7, 64, 192, 160, (2.0, 2.0), False, 0),        # upsampled tensor second, ragged column block
    (1, 4, 6, 256, 256, 256, (3.0, 2.0), True, 128),      # non-square scale, eight K-tiles
])
def test_conv_split3_reads_the_upsampled_source(hops, orc, n, lh, lw, cl, cs, oc, scale, up_first, split):
    """si_hip_conv2d_split3_upcat_f32 (round 6): the 1x1 conv behind upsample + concat on the f32_split arithmetic reads the upsampled channels
    from the low-resolution tensor at the reference's source pixel (src/layer/upsample.cpp:85-92) -- bit-identical to si_hip_conv2d_split3_f32 on
    the materialised concat (the upsampled range of the concat buffer is poisoned with NaN: nobody reads it), sibling split included, inside the
    fp32 bars against the oracle."""
    low = rng_uniform(4600, (n, lh, lw, cl), -2, 2)
    oh, ow = int(lh * scale[0]), int(lw * scale[1])
    skip = rng_uniform(4601, (n, oh, ow, cs), -2, 2)
    w = rng_uniform(4602, (oc, cl + cs, 1, 1), -0.2, 0.2)
    b = rng_uniform(4603, (oc,), -0.5, 0.5)
    upx = hops.upsample_nearest(low, scale[0], scale[1])
    cat = np.concatenate([upx, skip] if up_first else [skip, upx], -1)
    want = hops.conv2d_split3(cat, w, b, act1="silu")
    if split:
        ya, yb = hops.conv2d_upcat(low, skip, w, b, scale, up_first, act1="silu", split_oc=split, split3=True)
        got = np.concatenate([ya, yb], -1)
    else:
        got = hops.conv2d_upcat(low, skip, w, b, scale, up_first, act1="silu", split3=True)
    assert_exact(got, want, "split3 dual-source vs the materialised concat")
    assert_parity(got, orc.activation("silu", orc.conv2d(cat, w, b, (1, 1), (0, 0), path="naive")), what="split3 dual-source vs the oracle")


# ---- f32_split: the range guard and the dynamic range (round 6; VERDICT r05 missing 2 / weak 1) ----
@pytest.mark.parametrize("n,ih,iw,oc,s,act,in_ld", [
    (2, 40, 48, 64, 2, "silu", None),      # YOLOv5s conv_1's form
    (3, 21, 23, 64, 2, "silu", None),      # odd sizes: ragged M, every border
    (2, 17, 30, 40, 1, "relu", None),      # stride 1, 40 of 64 columns live
    (1, 12, 16, 32, 2, "none", 48),        # the input as a channel slice of a wider tensor (pixel stride 48, NaN between the slices)
])
def test_conv_split3_32_channel_3x3_forms(hops, orc, n, ih, iw, oc, s, act, in_ld):
    """3x3 convs over 32 channels on the split kernel (YOLOv5s conv_1 under f32_split, the largest layer of that path): the 64-row and the 128-row
    tile agree bit for bit (the same k-steps on the same two accumulator chains), the fp32 bars against the oracle (src/layer/conv_2d.cpp:207-283)
    hold, a strided input view reads only its own channels, and the guard sees an overflow."""
    x = rng_uniform(4800 + ih, (n, ih, iw, 32), -2, 2)
    w = rng_uniform(4801, (oc, 32, 3, 3), -0.2, 0.2)
    b = rng_uniform(4802, (oc,), -0.5, 0.5)
    got, flag = hops.conv2d_split3(x, w, b, (s, s), (1, 1), act1=act, return_flag=True, in_ld=in_ld)
    assert flag == 0
    with hops.plan(split3_bm=128):
        tall = hops.conv2d_split3(x, w, b, (s, s), (1, 1), act1=act, in_ld=in_ld)
    assert_exact(got, tall, "64-row vs 128-row tiles")
    ref = orc.conv2d(x, w, b, (s, s), (1, 1), path="naive")
    assert_parity(got, orc.activation(act, ref) if act != "none" else ref, what="split3, 32 channels")
    hot = x.copy()
    hot[n - 1, ih // 2, iw // 2, 7] = 1.0e5
    assert hops.conv2d_split3(hot, w, b, (s, s), (1, 1), act1="relu", return_flag=True, in_ld=in_ld)[1] == 1


def _split_kernels(hops):
    return {"split3": lambda x, w, b, **kw: hops.conv2d_split3(x, w, b, (1, 1), (1, 1), **kw),
            "wino_split": lambda x, w, b, **kw: hops.conv2d_wino23_split(x, w, b, (1, 1), **kw)}


@pytest.mark.parametrize("kernel", ["split3", "wino_split"])
def test_f32_split_range_guard_flags_an_overflow_and_nothing_else(hops, orc, kernel):
    """An activation that rounds to fp16 infinity cannot be split (the reference convolves any finite fp32, src/layer/conv_2d.cpp:207-283): the
    kernel reports it through SiConv2dDesc::range_flag -- one pixel at 1e5 sets the flag (and the outputs it feeds are non-finite: the flag
    is what tells the caller to recompute in true fp32); the same tensor without it, values right at fp16's largest finite number, and tiny
    values do not.  For the Winograd form the guard is on the TRANSFORMED input: four inputs of 3e4 each overflow their sum, no single one does.
    relu would have hidden a NaN (relu(NaN) = 0): the test is on the accumulators, in front of the activation."""
    conv = _split_kernels(hops)[kernel]
    x = rng_uniform(4100, (2, 12, 12, 64), -2, 2)
    w = rng_uniform(4101, (64, 64, 3, 3), -0.2, 0.2)
    b = rng_uniform(4102, (64,), -0.5, 0.5)
    y, flag = conv(x, w, b, act1="relu", return_flag=True)
    assert flag == 0 and np.isfinite(y).all()
    assert_parity(y, orc.activation("relu", orc.conv2d(x, w, b, (1, 1), (1, 1), path="naive")), what=kernel)
    hot = x.copy()
    hot[1, 5, 7, 13] = 1.0e5
    y, flag = conv(hot, w, b, act1="relu", return_flag=True)
    assert flag == 1, "one activation at 1e5 must trip the guard"
    # the largest finite fp16 neighbourhood still splits (65519 rounds to 65504, hi + lo carries the rest); split3 only: B^T d B sums inputs
    if kernel == "split3":
        edge = x.copy()
        edge[0, 3, 3, 5] = 65519.0
        y, flag = conv(edge, w, b, return_flag=True)
        assert flag == 0
        assert_parity(y, orc.conv2d(edge, w, b, (1, 1), (1, 1), path="naive"), what="split3 at fp16's edge")
    else:
        quad = np.zeros_like(x)
        quad[0, 4:8, 4:8, 3] = np.array([[3.0e4, 0, -3.0e4, 0]] * 4, np.float32) * np.array([[1], [0], [-1], [0]], np.float32)
        y, flag = conv(quad, w, b, return_flag=True)   # d00 - d02 - d20 + d22 = 1.2e5 > 65504 although every input is 3e4
        assert flag == 1, "a transformed input value above fp16's range must trip the guard"
    tiny = x * np.float32(1e-7)
    y, flag = conv(tiny, w, b, return_flag=True)
    assert flag == 0 and np.isfinite(y).all()
    # weights outside fp16's range: refused where they are split (the layer then stays on the fp32 kernels)
    wbig = w.copy()
    wbig[3, 4, 0, 0] = 2.0e5   # (a corner tap: U[0][0] = g[0][0] itself -- the centre tap only reaches U through quarters)
    with pytest.raises(hops.HipError):
        conv(x, wbig, b)


@pytest.mark.parametrize("k,s,p,oc,n,ih,iw,act", [
    (6, 2, 2, 32, 2, 64, 64, "silu"),       # YOLOv5's stem form (even fragments)
    (6, 2, 2, 32, 3, 40, 328, "silu"),      # two 160-pixel column tiles + a ragged one, odd row-block count
    (6, 2, 2, 64, 2, 32, 48, "relu"),       # 64 channels: two channel waves per workgroup
    (6, 2, 1, 32, 2, 30, 36, "none"),       # pad 1: fragments start on odd half indices (funnel-shift path)
    (7, 2, 3, 64, 2, 56, 56, "relu"),       # ResNet18's stem
    (7, 2, 3, 32, 1, 32, 44, "silu"),
    (6, 2, 2, 32, 2, 36, 100, "silu"),      # a last column tile of 18 pixels; 50 output rows: segments of uneven length
    (3, 2, 1, 16, 2, 48, 48, "hardswish"),  # MobileNetV2's stem
    (3, 2, 1, 96, 1, 24, 40, "silu"),       # > 64 channels: channel tiles as items
])
def test_stem_split3_vs_oracle_and_fp64(hops, orc, k, s, p, oc, n, ih, iw, act):
    """The RGB stem on the f32_split arithmetic (si_hip_conv2d_stem_split3_f32: the fp16 stem kernel's staging and MFMA loop on hi / lo halves,
    fp32 out) against the reference's convolution (src/layer/conv_2d.cpp:207-283) at the FP32 bars -- 1e-4 vs the oracle, 2e-5 vs float64 --
    element-wise too, no further from float64 than twice the true-fp32 stem kernel, and an image's bits do not depend on the batch."""
    from util import mixed_err
    x = rng_uniform(4500 + k, (n, ih, iw, 3), 0, 1)
    w = rng_uniform(4501 + k, (oc, 3, k, k), -0.3, 0.3)
    b = rng_uniform(4502 + k, (oc,), -0.5, 0.5)
    fact = (lambda t: t) if act == "none" else (lambda t: orc.activation(act, t))
    got, flag = hops.conv2d_stem_split3(x, w, b, (s, s), (p, p), act1=act, return_flag=True)
    assert flag == 0 and got.dtype == np.float32
    assert_parity(got, fact(orc.conv2d(x, w, b, (s, s), (p, p), path="auto")), what="split stem %dx%d" % (k, k))
    plain = hops.conv2d_stem_split3(x, w, b, (s, s), (p, p))
    naive = orc.conv2d(x, w, b, (s, s), (p, p), path="naive")
    e3 = np.abs(plain - naive).max() / np.abs(naive).max()
    f32 = hops.conv2d(x, w, b, (s, s), (p, p))
    e32 = np.abs(f32 - naive).max() / np.abs(naive).max()
    assert e3 <= 2e-5 and e3 <= 2.0 * e32 + 1e-7, (e3, e32)
    m, m32 = mixed_err(plain, naive), mixed_err(f32, naive)
    assert m <= 1e-4 and m <= 2.0 * m32 + 1e-7, (m, m32)
    if n > 1:
        one = hops.conv2d_stem_split3(x[n - 1:], w, b, (s, s), (p, p), act1=act)
        assert_exact(one, got[n - 1:], "split stem: batch position")


def test_stem_split3_random_shapes(hops, orc):
    """Seeded sweep over what the launcher decides from the shape: segments per column (image height x batch against the resident waves), the last
    column tile's width, the last row block's height (odd output heights: one of the two rows of an item is dropped), channel tiles (blockIdx.y),
    1 .. 3 input channels that still form a kernel row the kernel has -- each against the oracle at the fp32 bar, and a second call on the last
    image alone (an image's bits do not depend on the batch, whatever the segment cut)."""
    rng = np.random.default_rng(4700)
    cases = 0
    for k, p in ((6, 2), (7, 3), (3, 1)):
        for _ in range(6):
            n = int(rng.integers(1, 5))
            ih = int(rng.integers(k, 150))
            iw = 4 * int(rng.integers(3, 90))          # rows on 16-byte boundaries with 3 channels: iw * 3 % 4 == 0
            oc = int(rng.choice([8, 16, 24, 32, 48, 64, 96]))
            x = rng_uniform(4701 + cases, (n, ih, iw, 3), -1, 1)
            w = rng_uniform(4702 + cases, (oc, 3, k, k), -0.3, 0.3)
            b = rng_uniform(4703 + cases, (oc,), -0.5, 0.5)
            got, flag = hops.conv2d_stem_split3(x, w, b, (2, 2), (p, p), act1="silu", return_flag=True)
            assert flag == 0
            assert_parity(got, orc.activation("silu", orc.conv2d(x, w, b, (2, 2), (p, p), path="naive")), what="split stem %dx%d n=%d %dx%d oc=%d" % (k, k, n, ih, iw, oc))
            if n > 1:
                assert_exact(hops.conv2d_stem_split3(x[n - 1:], w, b, (2, 2), (p, p), act1="silu"), got[n - 1:], "split stem: batch position, %d x %d" % (ih, iw))
            cases += 1
    assert cases == 18


def test_stem_split3_full_size_against_the_fp32_stem(hops):
    """BASELINE's size (32 x 640 x 640 x 3 -> 320 x 320 x 32): 3840 wave segments of 13-14 row blocks, every store a buffer store with the pixel's
    distance in the scalar offset.  The whole output against the true-fp32 stem kernel under the element-wise bar (two arithmetics of fp32 class:
    |d| <= 1e-4 (|ref| + rms)), every image; the guard word stays 0; and the last image alone gives the bits it has in the batch."""
    from util import mixed_err
    n, sz = 32, 640
    x = rng_uniform(4900, (n, sz, sz, 3), 0, 1)
    w = rng_uniform(4901, (32, 3, 6, 6), -0.3, 0.3)
    b = rng_uniform(4902, (32,), -0.5, 0.5)
    got, flag = hops.conv2d_stem_split3(x, w, b, (2, 2), (2, 2), act1="silu", return_flag=True)
    ref = hops.conv2d(x, w, b, (2, 2), (2, 2), act1="silu")
    assert flag == 0 and got.shape == ref.shape == (n, 320, 320, 32)
    worst = max(mixed_err(got[i], ref[i]) for i in range(n))
    assert worst <= 1e-4, worst
    assert_exact(hops.conv2d_stem_split3(x[n - 1:], w, b, (2, 2), (2, 2), act1="silu"), got[n - 1:], "split stem: image 31 alone")


def test_stem_split3_range_guard_and_dynamic_range(hops, orc):
    """The split stem's range contract: a pixel value fp16 cannot hold sets the guard word (relu in front of the store would have hidden the NaN),
    values at fp16's edge and tiny images do not; images on 0..255 (un-normalised) hold the element-wise bar; weights out of range are refused
    at the pack."""
    from util import mixed_err
    x = rng_uniform(4600, (2, 48, 64, 3), 0, 1)
    w = rng_uniform(4601, (32, 3, 6, 6), -0.3, 0.3)
    b = rng_uniform(4602, (32,), -0.5, 0.5)
    y, flag = hops.conv2d_stem_split3(x, w, b, act1="relu", return_flag=True)
    assert flag == 0 and np.isfinite(y).all()
    hot = x.copy()
    hot[1, 17, 33, 2] = 1.0e5
    _, flag = hops.conv2d_stem_split3(hot, w, b, act1="relu", return_flag=True)
    assert flag == 1
    edge = x.copy()
    edge[0, 5, 5, 1] = 65519.0
    y, flag = hops.conv2d_stem_split3(edge, w, b, return_flag=True)
    assert flag == 0
    assert_parity(y, orc.conv2d(edge, w, b, (2, 2), (2, 2), path="naive"), what="split stem at fp16's edge")
    for scale in (1e-6, 1e-3, 255.0, 1e4):
        xs, bs = x * np.float32(scale), b * np.float32(scale)
        ref = orc.conv2d(xs, w, bs, (2, 2), (2, 2), path="naive")
        y, flag = hops.conv2d_stem_split3(xs, w, bs, return_flag=True)
        assert flag == 0 and mixed_err(y, ref) <= 1e-4, (scale, flag, mixed_err(y, ref))
    wbig = w.copy()
    wbig[5, 1, 2, 3] = 2.0e5
    with pytest.raises(hops.HipError):
        hops.conv2d_stem_split3(x, wbig, b)
    # image rows that do not start on 16-byte boundaries (width * channels not a multiple of 4): refused -- the engine keeps such a stem in fp32
    with pytest.raises(hops.HipError):
        hops.conv2d_stem_split3(rng_uniform(4603, (1, 33, 47, 3), 0, 1), w, b)


@pytest.mark.parametrize("kernel", ["split3", "wino_split"])
def test_f32_split_dynamic_range_sweep_elementwise(hops, orc, kernel):
    """The opt-in arithmetic across the dynamic range (round 5 tested x in U[-2, 2] only): tensor scales 1e-6 ... 1e4, under
    the ELEMENT-WISE bar |d| <= 1e-4 (|ref| + rms(ref)) against the float64 convolution -- and, from scale 1e-3 up, an error at most twice the
    true-fp32 kernel's under the same metric.  (Below 6e-5 the hi half is an fp16 subnormal: the pair still resolves 2.9e-11 absolute, i.e.
    ~1e-5 of a tensor whose scale is 1e-6 -- inside the bar, no longer fp32-class; include/si_hip.h states it.)  Where the scale makes the
    operands overflow fp16 (the Winograd form from ~1.6e4) the guard must say so instead."""
    from util import mixed_err
    conv = _split_kernels(hops)[kernel]
    f32 = (lambda x, w, b: hops.conv2d(x, w, b, (1, 1), (1, 1))) if kernel == "split3" else (lambda x, w, b: hops.conv2d_winograd(x, w, b, (1, 1)))
    x0 = rng_uniform(4200, (2, 14, 14, 128), -2, 2)
    w = rng_uniform(4201, (64, 128, 3, 3), -0.2, 0.2)
    b0 = rng_uniform(4202, (64,), -0.5, 0.5)
    for scale in (1e-6, 1e-4, 1e-2, 1.0, 1e2, 1e4):
        x, b = x0 * np.float32(scale), b0 * np.float32(scale)
        ref = orc.conv2d(x, w, b, (1, 1), (1, 1), path="naive")
        y, flag = conv(x, w, b, return_flag=True)
        if flag:
            assert kernel == "wino_split" and scale >= 1e4, "guard tripped at scale %g" % scale   # |B^T d B| up to 8e4
            continue
        m, m32 = mixed_err(y, ref), mixed_err(f32(x, w, b), ref)
        assert m <= 1e-4, "scale %g: element-wise error %.3e" % (scale, m)
        if scale >= 1e-3:
            assert m <= 2.0 * m32 + 1e-7, "scale %g: %.3e vs the fp32 kernel's %.3e" % (scale, m, m32)


@pytest.mark.parametrize("kernel", ["split3", "wino_split"])
def test_f32_split_mixed_magnitude_channels(hops, orc, kernel):
    """A wide dynamic range INSIDE one tensor: input channels of magnitude 1e-4 beside channels of magnitude 1e2, and output channels whose
    filters are 1e-3 of the others' -- held PER OUTPUT CHANNEL (|d| <= 1e-4 (|ref| + rms of that channel)), so that a small-magnitude channel
    cannot hide behind the tensor's largest value, and to at most twice the true-fp32 kernel's error under the same metric."""
    from util import mixed_err
    conv = _split_kernels(hops)[kernel]
    f32 = (lambda x, w, b: hops.conv2d(x, w, b, (1, 1), (1, 1))) if kernel == "split3" else (lambda x, w, b: hops.conv2d_winograd(x, w, b, (1, 1)))
    x = rng_uniform(4300, (2, 14, 14, 128), -2, 2)
    cs = np.where(np.arange(128) % 2 == 0, np.float32(1e-4), np.float32(1e2))
    x = x * cs
    w = rng_uniform(4301, (64, 128, 3, 3), -0.2, 0.2)
    w[::4] *= np.float32(1e-3)
    b = np.zeros(64, np.float32)
    ref = orc.conv2d(x, w, b, (1, 1), (1, 1), path="naive")
    y, flag = conv(x, w, b, return_flag=True)
    assert flag == 0
    m, m32 = mixed_err(y, ref, per_channel=True), mixed_err(f32(x, w, b), ref, per_channel=True)
    assert m <= 1e-4 and m <= 2.0 * m32 + 1e-7, (m, m32)
    # the small-magnitude INPUT channels alone (the large ones zeroed): what the max-based metric of round 5 could not see
    xs = x * (cs < 1).astype(np.float32)
    ref = orc.conv2d(xs, w, b, (1, 1), (1, 1), path="naive")
    m = mixed_err(conv(xs, w, b), ref, per_channel=True)
    assert m <= 1e-4, m


def test_detect_split3_range_guard(hops):
    """si_hip_conv2d_split3_yolo_f32 tests its accumulators in front of the sigmoid (sigmoid(Inf) = 1 would hide an overflow)"""
    na, ne, n = 3, 85, 2
    feats, ws, bs, grids, anchors = [], [], [], [], []
    for i, (h, c) in enumerate([(8, 128), (4, 256)]):
        feats.append(rng_uniform(4400 + i, (n, h, h, c), -1, 1))
        ws.append(rng_uniform(4410 + i, (na * ne, c, 1, 1), -0.3, 0.3))
        bs.append(rng_uniform(4420 + i, (na * ne,), -0.5, 0.5))
        gy, gx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="ij")
        grids.append(np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, h, h, 2)).copy())
        anchors.append(np.broadcast_to(rng_uniform(4430 + i, (1, na, 1, 1, 2), 5, 300), (1, na, h, h, 2)).copy())
    _, flags = hops.yolo_detect_split3(feats, ws, bs, grids, anchors, [8.0, 16.0], na, return_flags=True)
    assert flags == [0, 0]
    feats[1][1, 2, 3, 7] = 3.0e5
    _, flags = hops.yolo_detect_split3(feats, ws, bs, grids, anchors, [8.0, 16.0], na, return_flags=True)
    assert flags == [0, 1]
