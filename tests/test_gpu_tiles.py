"""GPU: every workgroup tile / MFMA shape of the fp32 implicit-GEMM convolution produces the SAME bits, and those bits are
the ones a scalar fmaf chain in the kernel's documented k order produces on the CPU (oracle path "chain").

This is what lets the tile policy follow the batch size (32-row tiles on v_mfma_f32_16x16x4_f32 for small launches, 64-row
tiles on v_mfma_f32_32x32x2_f32 for large ones) without touching the engine's contract that an image's result does not depend
on the batch it rides in (reference: Engine::Forward is per-image arithmetic, src/layer/conv_2d.cpp:207-283).
"""
import numpy as np
import pytest

from util import rng_uniform

pytestmark = pytest.mark.gpu

# variant ids of conv_igemm.hip (SiConvPlan::f32_tile: the plan travels inside each call's descriptor)
ALL_TILES = [4, 2, 0, 1, 5, 6, 10, 3, 11, 12, 13, 14, 16, 17, 19, 22]
SMALL_TILES = [11, 12, 13, 14, 16, 17, 19, 22]


@pytest.fixture(scope="module")
def hops(gpu):
    from simpleinfer_amd import hipops
    return hipops


@pytest.fixture()
def tile(gpu):
    from simpleinfer_amd import hipops

    def set_variant(v):
        hipops.set_plan(f32_tile=int(v))
    yield set_variant
    hipops.set_plan()


# (shape NHWC, oc, k, s, p, groups): fast-path shapes (ic/groups % 32 == 0) of every kind the YOLOv5s / ResNet18 graphs hold,
# sizes where M is not a multiple of any tile and oc is not a multiple of 64; then generic-path shapes (ragged channels)
SHAPES = [
    ((2, 21, 19, 64), 96, 3, 2, 1, 1),     # 3x3 s2, cb-major K = 576
    ((1, 20, 20, 256), 160, 3, 2, 1, 1),   # K = 2304
    ((3, 13, 17, 128), 64, 1, 1, 0, 1),    # pointwise
    ((2, 9, 9, 32), 32, 1, 1, 0, 1),       # oc <= 32 (128x32 default tile)
    ((2, 14, 14, 64), 128, 1, 2, 0, 1),    # 1x1 stride 2 (general instantiation, not pointwise)
    ((2, 12, 12, 64), 255, 1, 1, 0, 1),    # ragged oc
    ((2, 11, 11, 64), 64, 3, 1, 1, 2),     # grouped, 32 channels per group
    ((2, 10, 10, 40), 72, 1, 1, 0, 1),     # zero-padded K (PADK instantiation)
    ((2, 9, 11, 20), 24, 3, 1, 1, 1),      # generic kernel (channels not a multiple of 32)
]


@pytest.mark.parametrize("shape,oc,k,s,p,g", SHAPES)
def test_every_tile_matches_the_fma_chain_bit_for_bit(hops, orc, tile, shape, oc, k, s, p, g):
    seed = hash((shape, oc, k, s)) % 100000
    x = rng_uniform(seed, shape, -1, 1)
    w = rng_uniform(seed + 1, (oc, shape[3] // g, k, k), -0.5, 0.5)
    b = rng_uniform(seed + 2, (oc,), -0.5, 0.5)
    ref = orc.conv2d(x, w, b, (s, s), (p, p), (1, 1), g, path="chain")
    for v in ALL_TILES:
        tile(v)
        got = hops.conv2d(x, w, b, (s, s), (p, p), (1, 1), g)
        bad = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
        assert bad == 0, "tile variant %d: %d of %d elements differ from the fma chain (max abs %.3e)" % (
            v, bad, ref.size, float(np.abs(got - ref).max()))
    # and the chain is the convolution (the reference's arithmetic within its tolerance)
    naive = orc.conv2d(x, w, b, (s, s), (p, p), (1, 1), g, path="naive")
    assert np.abs(ref - naive).max() <= 2e-5 * np.abs(naive).max()


@pytest.mark.parametrize("act1,res,act2", [("silu", False, "none"), ("silu", True, "none"), ("none", True, "relu"), ("hardswish", False, "none")])
def test_every_tile_same_bits_through_the_fused_epilogues(hops, tile, act1, res, act2):
    x = rng_uniform(11, (2, 15, 13, 64), -1, 1)
    w = rng_uniform(12, (96, 64, 3, 3), -0.3, 0.3)
    b = rng_uniform(13, (96,), -0.5, 0.5)
    r = rng_uniform(14, (2, 15, 13, 96), -1, 1) if res else None
    outs = {}
    for v in [4, 0, 10] + SMALL_TILES:
        tile(v)
        outs[v] = hops.conv2d(x, w, b, (1, 1), (1, 1), act1=act1, residual=r, act2=act2, out_ld=128, out_c_off=16)
    for v, y in outs.items():
        assert np.array_equal(y.view(np.uint32), outs[4].view(np.uint32)), "tile %d differs from the 64x64 tile" % v


def test_every_tile_same_bits_sibling_split_and_upsampled_source(hops, tile):
    x = rng_uniform(21, (2, 10, 14, 64), -1, 1)
    wa, wb = rng_uniform(22, (32, 64, 1, 1), -0.5, 0.5), rng_uniform(23, (64, 64, 1, 1), -0.5, 0.5)
    ba, bb = rng_uniform(24, (32,)), rng_uniform(25, (64,))
    low, skip = rng_uniform(26, (2, 5, 7, 64), -1, 1), rng_uniform(27, (2, 10, 14, 32), -1, 1)
    wu, bu = rng_uniform(28, (96, 96, 1, 1), -0.5, 0.5), rng_uniform(29, (96,))
    base = None
    for v in [4] + SMALL_TILES:
        tile(v)
        ya, yb = hops.conv2d_split(x, wa, ba, wb, bb, act1="silu")
        yu = hops.conv2d_upcat(low, skip, wu, bu, act1="silu")
        y1, y2 = hops.conv2d_upcat(low, skip, wu, bu, act1="silu", split_oc=32)
        cur = [ya, yb, yu, y1, y2]
        if base is None:
            base = cur
        for a, c in zip(base, cur):
            assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), "tile %d" % v


def test_every_tile_same_bits_in_the_detect_epilogue(hops, tile):
    """YOLOv5 Detect (sigmoid + grid / anchor decode + concat in the conv epilogue, reference src/layer/yolo_detect.cpp:223-266) on the
    64x64 tile and on every 16x16-MFMA tile: the same bits, including maps smaller than a tile (rows spanning images)."""
    na, ne, n = 3, 85, 3
    feats, ws, bs, grids, anchors = [], [], [], [], []
    for i, (h, c) in enumerate([(12, 64), (6, 128), (3, 256)]):
        feats.append(rng_uniform(50 + i, (n, h, h, c), -1, 1))
        ws.append(rng_uniform(60 + i, (na * ne, c, 1, 1), -0.3, 0.3))
        bs.append(rng_uniform(70 + i, (na * ne,), -0.5, 0.5))
        gy, gx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="ij")
        grids.append(np.broadcast_to(np.stack([gx - 0.5, gy - 0.5], -1)[None, None], (1, na, h, h, 2)).copy())
        anchors.append(np.broadcast_to(rng_uniform(80 + i, (1, na, 1, 1, 2), 5, 300), (1, na, h, h, 2)).copy())
    base = None
    for v in [4] + SMALL_TILES:
        tile(v)
        y = hops.yolo_detect(feats, ws, bs, grids, anchors, [8.0, 16.0, 32.0], na, fused=True)
        if base is None:
            base = y
        assert np.array_equal(y.view(np.uint32), base.view(np.uint32)), "tile %d" % v


def test_policy_follows_the_launch_size(hops):
    """the default policy (conv_igemm.hip conv_variant): a launch with thousands of 64x64 tiles keeps that tile, a mid-sized one
    gets half tiles, a small one 32x32 tiles -- the same layer at batch 32 / 8 / 1"""
    layer = ((160, 160, 64), (128, 64, 3, 3), (2, 2), (1, 1))
    names = {b: hops.conv2d_kernel_name((b,) + layer[0], *layer[1:]) for b in (32, 8, 1)}
    assert "<64, 64," in names[32], names
    assert "<32, 64," in names[8], names
    assert "<32, 32," in names[1], names
