"""GPU: graph-level parity of the Engine (C++ host + HIP kernels, driven through include/si_engine.h)
against the whole-graph CPU oracle (oracle/orc.py run_graph), plus the engine-only properties: fusion and
concat aliasing change nothing, hipGraph replay changes nothing, batch sharding is bit-exact."""
import numpy as np
import pytest

from util import assert_detect_parity, assert_exact, assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def si(gpu):
    import simpleinfer_amd
    return simpleinfer_amd


def _save(tmp_path, builder, tag):
    pp, bp = str(tmp_path / (tag + ".pnnx.param")), str(tmp_path / (tag + ".pnnx.bin"))
    builder.save(pp, bp)
    return pp, bp


def _run(si, pp, bp, x, **opts):
    e = si.Engine(**opts)
    e.load_model(pp, bp)
    (iname,), (oname,) = e.input_names(), e.output_names()
    e.input(iname, x)
    e.forward()
    return e, oname, e.extract(oname)


MODELS = {
    "toy_yolo": (lambda mg: mg.build_toy_yolo(2, 64), (2, 64, 64, 3)),
    "toy_classifier": (lambda mg: mg.build_toy_classifier(2, 32), (2, 32, 32, 3)),
    "resnet18_small": (lambda mg: mg.build_resnet18(2, 64, num_classes=100, base=16), (2, 64, 64, 3)),
    "yolov5s_160": (lambda mg: mg.build_yolov5s(2, 160), (2, 160, 160, 3)),
    "mobilenetv3_small_96": (lambda mg: mg.build_mobilenetv3_small(2, 96, num_classes=100), (2, 96, 96, 3)),
}


@pytest.mark.parametrize("name", sorted(MODELS))
def test_graph_parity_vs_oracle(si, orc, tmp_path, name):
    mk, shape = MODELS[name]
    pp, bp = _save(tmp_path, mk(si.modelgen), name)
    x = si.modelgen.synth_input(shape)
    ref = orc.run_graph(pp, bp, {"0": x})
    e, oname, got = _run(si, pp, bp, x)
    # Detect outputs are held per column group (scores absolute), everything else against the tensor's own scale
    check = assert_detect_parity if "yolo" in name else assert_parity
    check(got, ref[oname], what=name)
    # plain schedule (no fusion, no aliasing: one launch per reference layer) gives the same numbers
    _, _, plain = _run(si, pp, bp, x, fuse=0, alias_cat=0)
    check(plain, ref[oname], what=name + " unfused")
    if "yolo" in name:
        assert_detect_parity(got, plain, 1e-6, 1e-6, what=name + " fused vs unfused")
    else:
        assert_parity(got, plain, 1e-6, what=name + " fused vs unfused")


def test_winograd_schedules_agree(si, orc, tmp_path):
    """winograd = 0 (implicit GEMM everywhere) / 1 (F(2,3) where faster) / 2 (F(4,3) on those layers) are the same
    function; each is held to the parity bar against the oracle."""
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(2, 160), "yw")
    x = si.modelgen.synth_input((2, 160, 160, 3))
    ref = orc.run_graph(pp, bp, {"0": x})
    kernels = {}
    for w in (0, 1, 2):
        e, oname, got = _run(si, pp, bp, x, winograd=w)
        assert_detect_parity(got, ref[oname], what="winograd=%d" % w)
        kernels[w] = {L["kernel"] for L in e.profile()}
    assert not any("wino" in k for k in kernels[0])
    assert any("conv_wino23" in k for k in kernels[1]) and not any("conv_wino43" in k for k in kernels[1])
    assert any("conv_wino43" in k for k in kernels[2])


def test_f32_split_option_holds_the_fp32_parity_bars(si, orc, tmp_path):
    """Engine option f32_split (round 5, opt-in): fp32 tensors, the dense convs over multiples of 64 channels contracted on the fp16
    matrix cores from three fp16 products per fp32 product (conv_split3.hip).  Held to the UNCHANGED fp32 bars against the oracle
    (src/layer/conv_2d.cpp:207-283: 1e-4 of the tensor's scale; Detect per column group), on YOLOv5s and ResNet18; the schedule
    really runs the split kernel, an image's result does not depend on its batch, and the default engine does not use it."""
    for name, mk, shape in (("ys", lambda mg: mg.build_yolov5s(2, 160), (2, 160, 160, 3)), ("rs", lambda mg: mg.build_resnet18(2, 64), (2, 64, 64, 3))):
        pp, bp = _save(tmp_path, mk(si.modelgen), name)
        x = si.modelgen.synth_input(shape)
        ref = orc.run_graph(pp, bp, {"0": x})
        e, oname, got = _run(si, pp, bp, x, f32_split=1)
        (assert_detect_parity if name == "ys" else assert_parity)(got, ref[oname], what="f32_split " + name)
        kernels = [L["kernel"] for L in e.profile()]
        assert sum(k == "conv_split3_f32_kernel" for k in kernels) >= (10 if name == "ys" else 5), kernels
        e0, _, got0 = _run(si, pp, bp, x)
        assert not any("split3" in L["kernel"] for L in e0.profile())
        one = si.Engine(f32_split=1, batch=1)
        one.load_model(pp, bp)
        one.input("0", x[1:])
        one.forward()
        assert_exact(one.extract(oname), got[1:], "f32_split: the second image alone vs in the batch")


def _one_conv(mg, n, c, hw, cout, k, s, act=True):
    b = mg.PnnxBuilder(seed=3)
    x = b.input((n, c, hw, hw))
    y = b.conv(x, cout, k, s)
    b.output(b.silu(y) if act else y)
    return b


@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("c,cout,k,s,kern", [(128, 128, 3, 2, "conv_split3_f32_kernel"),    # strided spatial conv: the split implicit GEMM
                                            (64, 64, 3, 1, "conv_wino23s_kernel"),          # 3x3 s1: the split form of the fused Winograd kernel
                                            (512, 256, 1, 1, "conv_split3_f32_kernel")])    # wide 1x1
def test_f32_split_range_guard_reruns_the_step_in_true_fp32(si, tmp_path, c, cout, k, s, kern, graph):
    """VERDICT r05 missing 2: the reference convolves any finite fp32 (src/layer/conv_2d.cpp:207-283); the opt-in f32_split arithmetic cannot
    split a value that rounds to fp16 infinity.  The kernels report it (SiConv2dDesc::range_flag), and Forward() -- at the sync it performs
    anyway -- sends that layer back to the true-fp32 kernels for the engine's lifetime and re-runs the step in place: the caller gets the
    DEFAULT engine's bits, never an Inf.  Eager and under hipGraph replay (the demoted layer re-packs its weights, so the captured graph
    is dropped and the next step runs eagerly once)."""
    mg = si.modelgen
    pp, bp = _save(tmp_path, _one_conv(mg, 2, c, 20, cout, k, s), "guard")
    x = mg.synth_input((2, 20, 20, c))
    hot = x.copy()
    hot[1, 7, 9, 5] = 1.0e5
    e0, oname, want_x = _run(si, pp, bp, x)
    e0.input("0", hot)
    e0.forward()
    want_hot = e0.extract(oname).copy()
    assert np.isfinite(want_hot).all()
    e = si.Engine(f32_split=1, graph=graph)
    e.load_model(pp, bp)
    e.input("0", x)
    for _ in range(3):       # (graph=1: eager, capture, replay)
        e.forward()
    assert [L["kernel"] for L in e.profile() if L["type"] == "nn.Conv2d"] == [kern]
    assert_parity(e.extract(oname), want_x, what="f32_split on ordinary data")
    assert e.schedule()["split_reruns"] == 0
    e.input("0", hot)
    e.forward()
    assert_exact(e.extract(oname), want_hot, "one activation at 1e5: the step was re-run on the true-fp32 kernels")
    sch = e.schedule()
    assert sch["split_reruns"] == 1 and sch["split_demoted"] == ["conv_0"], sch
    assert "split" not in [L["kernel"] for L in e.profile() if L["type"] == "nn.Conv2d"][0] and "wino23s" not in e.profile()[0]["kernel"]
    e.input("0", x)
    for _ in range(3):
        e.forward()
    assert_exact(e.extract(oname), want_x, "after a trip the layer stays on the true-fp32 kernels")
    assert e.schedule()["split_reruns"] == 1


def test_f32_split_stem_on_the_split_kernel(si, orc, tmp_path):
    """f32_split_policy = 4: the RGB stem (YOLOv5's 6x6 s2, ResNet18's 7x7 s2) on the split form of the fp16 stem kernel -- parity at the fp32 bars
    against the oracle, the kernel really runs, and an image holding a value fp16 cannot hold sends the stem back to the fp32 stem kernel with
    the default engine's bits."""
    mg = si.modelgen
    for name, mk, shape in (("ys", lambda: mg.build_yolov5s(2, 160), (2, 160, 160, 3)), ("rs", lambda: mg.build_resnet18(2, 64), (2, 64, 64, 3))):
        pp, bp = _save(tmp_path, mk(), "stem" + name)
        x = mg.synth_input(shape)
        ref = orc.run_graph(pp, bp, {"0": x})
        e, oname, got = _run(si, pp, bp, x, f32_split=1, f32_split_policy=4)
        (assert_detect_parity if name == "ys" else assert_parity)(got, ref[oname], what="f32_split stem " + name)
        convs = [L for L in e.profile() if L["type"] == "nn.Conv2d"]
        assert convs[0]["kernel"] == "conv_stem_split_f32_kernel", convs[0]
        e3, _, _ = _run(si, pp, bp, x, f32_split=1, f32_split_policy=3)
        assert "split" not in [L for L in e3.profile() if L["type"] == "nn.Conv2d"][0]["kernel"]
    b = _one_conv(mg, 2, 3, 64, 32, 6, 2)
    pp, bp = _save(tmp_path, b, "stemguard")
    x = mg.synth_input((2, 64, 64, 3))
    hot = x.copy()
    hot[1, 30, 31, 2] = 1.0e5
    e0, oname, _ = _run(si, pp, bp, hot)
    want = e0.extract(oname).copy()
    e, _, got = _run(si, pp, bp, hot, f32_split=1, f32_split_policy=4)
    assert_exact(got, want, "a pixel at 1e5: the stem re-ran on the true-fp32 kernel")
    sch = e.schedule()
    assert sch["split_reruns"] == 1 and sch["split_demoted"] == ["conv_0"], sch
    # an image whose rows do not start on 16-byte boundaries (33 x 3 floats): the split stem refuses it, the layer runs on the fp32 stem kernel
    pp, bp = _save(tmp_path, _one_conv(mg, 2, 3, 33, 32, 6, 2), "stemodd")
    x = mg.synth_input((2, 33, 33, 3))
    _, oname, want = _run(si, pp, bp, x)
    e, _, got = _run(si, pp, bp, x, f32_split=1, f32_split_policy=4)
    assert_exact(got, want, "odd image width: the fp32 stem kernel")
    sch = e.schedule()
    assert sch["split_reruns"] == 0 and sch["split_demoted"] == ["conv_0"], sch


def test_f32_split_weights_out_of_range_keep_the_layer_in_fp32(si, tmp_path):
    """weights are checked where they are split, at load: a layer with a weight fp16 cannot hold never runs on the split kernels"""
    mg = si.modelgen
    b = _one_conv(mg, 2, 128, 16, 128, 3, 2)
    b.attrs["conv_0.weight"][5, 6, 0, 0] = 1.0e5
    pp, bp = _save(tmp_path, b, "wguard")
    x = mg.synth_input((2, 16, 16, 128))
    _, oname, want = _run(si, pp, bp, x)
    e, _, got = _run(si, pp, bp, x, f32_split=1)
    assert_exact(got, want, "a weight at 1e5: true-fp32 kernels from the first step")
    sch = e.schedule()
    assert sch["split_reruns"] == 0 and sch["split_demoted"] == ["conv_0"], sch


def test_f32_split_range_guard_in_a_whole_network(si, tmp_path):
    """YOLOv5s with one input pixel at 1e6: the stem's output overflows fp16 in front of the first split layer; every split layer downstream of
    it sees non-finite operands in that pass, all of them go back to true fp32 in ONE re-run, and the caller's result is the default engine's.
    Also through the two-lane schedule and the sliced host pipeline (child engines carry their own flags)."""
    mg = si.modelgen
    pp, bp = _save(tmp_path, mg.build_yolov5s(8, 160), "yguard")
    x = mg.synth_input((8, 160, 160, 3))
    x[5, 80, 80, 1] = 1.0e6
    _, oname, want = _run(si, pp, bp, x)
    assert np.isfinite(want).all()
    for opts in ({}, {"streams": 2}, {"host_slices": 2}, {"graph": 1}):
        e = si.Engine(f32_split=1, **opts)
        e.load_model(pp, bp)
        e.input("0", x)
        e.forward()
        got = e.extract(oname)
        assert np.isfinite(got).all(), opts
        assert_detect_parity(got, want, what="f32_split guard, %s" % (opts,))
        sch = e.schedule()
        assert sch["split_reruns"] >= 1 and len(sch["split_demoted"]) >= 1, (opts, sch)
        e.forward()
        assert_exact(e.extract(oname), got, "second forward after the trip, %s" % (opts,))


def test_yolov5s_640_batch1_parity(si, orc, tmp_path):
    """BASELINE.json configs[1]: YOLOv5s 1x3x640x640 fp32 parity vs the CPU outputs."""
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(1, 640), "y1")
    x = si.modelgen.synth_input((1, 640, 640, 3))
    ref = orc.run_graph(pp, bp, {"0": x})
    e, oname, got = _run(si, pp, bp, x)
    assert got.shape == (1, 25200, 85)
    assert_detect_parity(got, ref[oname], what="yolov5s 640")
    # known-answer style input of the reference's demo (test_yolo2.cpp:26 uses a constant image)
    xc = np.full((1, 640, 640, 3), 0.5, np.float32)
    e.input("0", xc)
    e.forward()
    assert_detect_parity(e.extract(oname), orc.run_graph(pp, bp, {"0": xc})[oname], what="constant input")


def _profile_after_forward(si, e, shape):
    e.input(e.input_names()[0], si.modelgen.synth_input(shape))
    e.forward()
    return e.profile()


def test_schedule_fusion_and_aliasing(si, tmp_path):
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(1, 64), "sched")  # full-width YOLOv5s at 64x64
    e = si.Engine()
    e.load_model(pp, bp)
    s = e.schedule()
    # all 57 SiLUs and the 7 residual adds of YOLOv5s fold into conv epilogues
    assert sum(n.startswith("silu_") for n in s["fused"]) == 57
    assert sum(n.startswith("add_") for n in s["fused"]) == 7
    assert not any(n.startswith("silu_") for n in s["run"])
    assert len(s["alias"]) >= 20  # concat inputs written in place
    # the 8 C3 blocks: cv2 is computed by cv1's launch (same input, same 1x1 geometry)
    siblings = [n for n in s["fused"] if n.startswith("conv_")]
    assert len(siblings) == 8
    # SPPF: the second and third 5x5 pool are produced by the first one's launch
    assert sorted(n for n in s["fused"] if n.startswith("maxpool")) == ["maxpool_1", "maxpool_2"] and "maxpool_0" in s["run"]
    assert "maxpool5_chain3" in {L["kernel"] for L in _profile_after_forward(si, e, (1, 64, 64, 3))}
    e2 = si.Engine(fuse=0, alias_cat=0)
    e2.load_model(pp, bp)
    s2 = e2.schedule()
    # 57 SiLU + 7 add, 8 sibling convs, 2 pools of the SPPF chain, 2 upsamples read at the source
    assert sorted(n for n in s["fused"] if n.startswith("upsample")) == ["upsample_0", "upsample_1"]
    assert not s2["fused"] and not s2["alias"] and len(s2["run"]) == len(s["run"]) + 64 + 8 + 2 + 2


def test_upsample_read_at_the_source_is_bit_identical(si, tmp_path):
    """FuseUpsampleIntoConvs: the PAN top-down path's two upsamples disappear from the schedule (the convs behind the concat read
    the low-resolution tensor with the reference's index rule, upsample.cpp:85-92) and the network output does not change by a
    bit against the schedule that materialises them."""
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(2, 160), "ups")
    x = si.modelgen.synth_input((2, 160, 160, 3))
    e1, oname, fused = _run(si, pp, bp, x)
    e0, _, plain = _run(si, pp, bp, x, fuse_upsample=0)
    assert_exact(fused, plain, "upsample read at the source vs materialised")
    s1, s0 = e1.schedule(), e0.schedule()
    assert len(s0["run"]) == len(s1["run"]) + 2 and not any(n.startswith("upsample") for n in s1["run"])
    k1 = [L["kernel"] for L in e1.profile()]
    assert sum(", false, true, false, false, " in k for k in k1) == 2 and "upsample_nearest" not in k1, k1   # (the UPS instantiation)
    assert [L["kernel"] for L in e0.profile()].count("upsample_nearest") == 2


def test_upsample_read_at_the_source_with_fp16_storage(si, tmp_path):
    """The same with fp16 storage (round 4, si_hip_conv2d_upcat_f16): no upsample launch in the fp16 schedule either, and not a bit
    of difference against the fp16 schedule that materialises the upsampled tensors."""
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(2, 160), "ups16")
    x = si.modelgen.synth_input((2, 160, 160, 3))
    e1, oname, fused = _run(si, pp, bp, x, fp16=1)
    e0, _, plain = _run(si, pp, bp, x, fp16=1, fuse_upsample=0)
    assert_exact(fused, plain, "fp16: upsample read at the source vs materialised")
    s1, s0 = e1.schedule(), e0.schedule()
    assert len(s0["run"]) == len(s1["run"]) + 2 and not any(n.startswith("upsample") for n in s1["run"])
    assert not any("upsample" in L["kernel"] for L in e1.profile())


def test_stem_pair_in_one_launch_with_fp16_storage(si, tmp_path):
    """Round 4 (FuseStemPairs, si_hip_conv2d_stem_s2c32_f16): with fp16 storage YOLOv5's conv_0 (RGB stem) and conv_1 (3x3 s2 over its 32
    channels) are ONE launch; the 32-channel intermediate is never allocated or written.  Not a bit of difference against the schedule
    that runs them separately, at two batches and with a ragged tile grid; the fp32 schedule is untouched."""
    for n, size, tag in ((2, 160, "sp"), (3, 96, "sp3")):
        pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(n, size), tag)
        x = si.modelgen.synth_input((n, size, size, 3))
        e1, oname, fused = _run(si, pp, bp, x, fp16=1, fuse_stem=1)
        e0, _, plain = _run(si, pp, bp, x, fp16=1, fuse_stem=0)
        assert_exact(fused, plain, "fp16: stem pair in one launch vs two")
        s1, s0 = e1.schedule(), e0.schedule()
        assert len(s0["run"]) == len(s1["run"]) + 1 and "conv_0" in s1["fused"] and "conv_0" not in s1["run"] and "conv_0" in s0["run"]
        k1 = [L["kernel"] for L in e1.profile()]
        assert k1.count("conv_stem_s2c32_f16_kernel<false>") == 1 and "conv_stem_f16_kernel" not in k1, k1
        assert "conv_stem_f16_kernel" in [L["kernel"] for L in e0.profile()]
        for _ in range(3):
            e1.forward()
            assert_exact(e1.extract(oname), plain, "repeated forwards")
        # round 6 (FuseStemTriples, si_hip_conv2d_stem_s2c32_pw_f16; the default): the first C3's cv1 | cv2 -- the 1x1 conv(s) that read conv_1's
        # 64 channels -- in the same launch; conv_1's output is never allocated either.  One launch and one tensor round trip less, the same bits
        # (also under hipGraph replay and at a re-batched engine).
        for opts in ({}, {"graph": 1}):
            e2, _, triple = _run(si, pp, bp, x, fp16=1, **opts)
            assert_exact(triple, plain, "fp16: stem + 3x3 s2 + 1x1 in one launch vs three")
            s2 = e2.schedule()
            assert len(s0["run"]) == len(s2["run"]) + 2 and "conv_1" in s2["fused"] and "conv_1" not in s2["run"], s2["run"][:6]
            k2 = [L["kernel"] for L in e2.profile()]
            assert k2.count("conv_stem_s2c32_f16_kernel<true>") == 1 and not any(k.startswith("conv_stem_s2c32_f16_kernel<false>") for k in k2), k2
            for _ in range(3):
                e2.forward()
                assert_exact(e2.extract(oname), plain, "repeated forwards")
    e32 = si.Engine()
    e32.load_model(pp, bp)
    assert "conv_0" in e32.schedule()["run"]


def test_bottleneck_pairs_in_one_launch_with_fp16_storage(si, tmp_path):
    """Round 5 (FuseBottleneckPairs, si_hip_conv2d_pw_slab_f16): with fp16 storage the 1x1 conv of a C3 bottleneck is computed inside
    the kernel of the 3x3 conv behind it -- over 128 channels in the slab kernel when its grid covers the chip, over 64 channels in the
    persistent patch kernel (also the 160x160x32 pair).  YOLOv5s 640x640 at batch 32: the five 40x40 pairs, the three 80x80 pairs and
    the 160x160 one, nine launches and nine tensor round trips less.  Not a bit of difference against fuse_pw=0; at a small batch the 128-channel pairs keep two launches;
    hipGraph replay and repeated forwards agree; the fp32 schedule is untouched."""
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(32, 640), "bp32")
    x = si.modelgen.synth_input((32, 640, 640, 3))
    e1, oname, fused = _run(si, pp, bp, x, fp16=1, fuse_pw=1)
    e0, _, plain = _run(si, pp, bp, x, fp16=1, fuse_pw=0)
    assert_exact(fused, plain, "fp16: bottleneck pairs in one launch vs two")
    s1, s0 = e1.schedule(), e0.schedule()
    assert len(s0["run"]) == len(s1["run"]) + 9, (len(s0["run"]), len(s1["run"]))
    k1 = [L["kernel"] for L in e1.profile()]
    assert k1.count("conv3x3s1_slab_f16_kernel<pw + 3x3>") == 5 and k1.count("conv_pw_patch_f16_kernel<pw + 3x3>") == 4, k1
    assert not any("pw + 3x3" in L["kernel"] for L in e0.profile())
    for _ in range(2):
        e1.forward()
        assert_exact(e1.extract(oname), plain, "repeated forwards")
    # round 6 (FuseCv3IntoPairs, si_hip_conv2d_pw_cv3_f16; fuse_pw = 2, OPT-IN: measured 0.74-0.84x of the launches it replaces): the two 80x80
    # C3s' closing convs behind their last bottleneck pair in the same launch -- per C3 the pair's step and the concat's disappear, neither the
    # pair's output nor the concat buffer is allocated (the arena shrinks), and not a bit changes; eager, replayed as a hipGraph, repeatedly
    for opts in ({}, {"graph": 1}):
        e3, _, tail = _run(si, pp, bp, x, fp16=1, fuse_pw=2, **opts)
        assert_exact(tail, plain, "fp16: C3 tails (pair + concat + cv3) in one launch")
        s3 = e3.schedule()
        assert len(s1["run"]) == len(s3["run"]) + 4, (len(s1["run"]), len(s3["run"]))
        assert sum(n.startswith("cat_") for n in s3["fused"]) == 2 and s3["arena_bytes"] < s1["arena_bytes"], (s3["fused"], s3["arena_bytes"], s1["arena_bytes"])
        k3 = [L["kernel"] for L in e3.profile()]
        assert k3.count("conv_pw_patch_f16_kernel<pw + 3x3 + cv3>") == 2 and k3.count("conv_pw_patch_f16_kernel<pw + 3x3>") == 2, k3
        for _ in range(2):
            e3.forward()
            assert_exact(e3.extract(oname), plain, "repeated forwards")
    # the same file served at batch 2: the slab grids do not cover the chip, only the 64-channel pairs are fused, and an image's bits are the same
    e2 = si.Engine(fp16=1, batch=2)
    e2.load_model(pp, bp)
    e2.input("0", x[:2])
    e2.forward()
    k2 = [L["kernel"] for L in e2.profile()]
    assert not any("slab_f16_kernel<pw + 3x3>" in k for k in k2) and sum(k.startswith("conv_pw_patch_f16_kernel<pw + 3x3") for k in k2) == 4, k2
    assert_exact(e2.extract(oname), plain[:2], "batch 2 (two launches per pair) vs batch 32 (one)")
    e32 = si.Engine()
    e32.load_model(pp, bp)
    assert len(e32.schedule()["run"]) == len(s0["run"]) + 2      # (fp32: no stem triple, no bottleneck pairs)


@pytest.mark.parametrize("graph", [0, 1])
def test_detect_levels_on_a_second_stream_are_bit_identical(si, tmp_path, graph):
    """Option detect_stream (on by default): the two finer Detect levels launch on a second stream right after the step that completes their
    feature map (fork / join by events, captured into the hipGraph the same way); nothing else changes -- same launches, same
    bits, forward after forward, and per-layer profiling still sees Detect as one step."""
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(2, 160), "dets")
    x = si.modelgen.synth_input((2, 160, 160, 3))
    e0, oname, ref = _run(si, pp, bp, x, graph=graph, detect_stream=0)
    e1, _, got = _run(si, pp, bp, x, graph=graph, detect_stream=2)   # (1, the default, only forks for levels of 4 GFLOP and up)
    assert_exact(got, ref, "detect_stream vs single stream")
    for _ in range(4):   # (with graph=1 the second forward captures, the later ones replay)
        e1.forward()
        assert_exact(e1.extract(oname), ref, "repeated forwards with detect_stream")
    assert e1.schedule()["run"] == e0.schedule()["run"]
    assert [L["kernel"] for L in e1.profile()] == [L["kernel"] for L in e0.profile()]
    e1.forward()
    assert_exact(e1.extract(oname), ref, "forward after a profile pass")


@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("name,batch", [("yolov5s_160", 4), ("resnet18_small", 6), ("mobilenetv3_small_96", 2)])
def test_two_half_batch_lanes_are_bit_identical_to_one_stream(si, tmp_path, name, batch, graph):
    """Option streams=2: the batch runs as two half-batch lanes on two streams that read / write slab views of the engine's own
    input and output buffers (fork / join by events; under graph=1 every lane replays its own captured graph).  Same function: bit-identical to the
    one-stream schedule, forward after forward, with host and with device-resident tensors; the profile lists every layer once
    per lane; an odd batch is a load-time Status."""
    mg = si.modelgen
    mk = {"yolov5s_160": lambda b: mg.build_yolov5s(b, 160), "resnet18_small": lambda b: mg.build_resnet18(b, 64, num_classes=100, base=16),
          "mobilenetv3_small_96": lambda b: mg.build_mobilenetv3_small(b, 96, num_classes=100)}[name]
    size = {"yolov5s_160": 160, "resnet18_small": 64, "mobilenetv3_small_96": 96}[name]
    pp, bp = _save(tmp_path, mk(batch), name)
    x = mg.synth_input((batch, size, size, 3))
    e1, oname, ref = _run(si, pp, bp, x, streams=1, graph=graph)
    e2, _, got = _run(si, pp, bp, x, streams=2, graph=graph)
    assert_exact(got, ref, "two lanes vs one stream")
    x2 = mg.synth_input((batch, size, size, 3), seed=7)
    e1.input(e1.input_names()[0], x2); e1.forward()
    ref2 = e1.extract(oname)
    for _ in range(3):   # (with graph=1 the second forward captures, the later ones replay; the host input is read at Forward time)
        e2.input(e2.input_names()[0], x2); e2.forward()
        assert_exact(e2.extract(oname), ref2, "repeated forwards on two lanes")
    p1, p2 = e1.profile(), e2.profile()
    assert [L["kernel"].split("<")[0] for L in p2] == [L["kernel"].split("<")[0] for L in p1] * 2
    assert abs(sum(L["flops"] for L in p2) - sum(L["flops"] for L in p1)) <= 1e-6 * sum(L["flops"] for L in p1)
    assert e2.schedule()["run"] == e1.schedule()["run"]
    e2.forward()
    assert_exact(e2.extract(oname), ref2, "forward after a profile pass")


@pytest.mark.parametrize("graph,pin", [(0, 1), (1, 1), (0, 0)])
@pytest.mark.parametrize("slices", [0, 2, 4])
def test_host_tensor_pipeline_is_bit_identical(si, tmp_path, slices, graph, pin):
    """The reference's calling convention -- Input() borrows a HOST tensor that is read at Forward() time, Extract() returns host
    memory (src/engine_impl.cpp:522-555) -- served as a pipeline of batch slices inside one synchronous Forward() (option
    host_slices; 0 = auto: slices of 4 images here): upload of slice g, compute of slice g-1 and download of slice g-2
    overlap.  Same bits as the unsliced schedule, forward after forward, with the input buffer rewritten in place between
    forwards (with pin_inputs=1 it is pinned in place from the second forward on) and with a different buffer."""
    mg = si.modelgen
    batch = 16
    pp, bp = _save(tmp_path, mg.build_yolov5s(batch, 96), "hs")
    x = mg.synth_input((batch, 96, 96, 3))
    e0, oname, ref = _run(si, pp, bp, x, host_slices=1, graph=graph)
    ref = ref.copy()
    e1 = si.Engine(host_slices=slices, graph=graph, pin_inputs=pin)
    e1.load_model(pp, bp)
    iname = e1.input_names()[0]
    buf = x.copy()
    e1.input(iname, buf)
    for it in range(4):
        e1.forward()
        assert_exact(e1.extract(oname), ref, "sliced host pipeline, forward %d" % it)
    # the borrowed buffer is read at Forward() time: rewrite it in place (it is pinned by now), then hand over another one
    x2 = mg.synth_input((batch, 96, 96, 3), seed=9)
    e0.input(iname, x2); e0.forward()
    ref2 = e0.extract(oname).copy()
    buf[...] = x2
    e1.forward()
    assert_exact(e1.extract(oname), ref2, "input rewritten in place")
    other = x.copy()
    e1.input(iname, other)
    e1.forward()
    assert_exact(e1.extract(oname), ref, "a different input buffer")
    # a device-resident input on the same engine takes the ordinary path
    from simpleinfer_amd import hipops
    dx = hipops.DeviceBuffer.from_numpy(x2)
    e1.input_device(iname, dx.ptr)
    e1.forward()
    assert_exact(e1.extract(oname), ref2, "device-resident input after host inputs")
    e1.release()
    dx.free()


def test_borrowed_input_may_be_freed_after_forward_and_a_failed_pipeline_falls_back(si, tmp_path):
    """(1) by default nothing of the caller's stays registered with the driver across calls, so a borrowed buffer
    may be freed after Forward() and a new one -- at the same address or not -- handed over (the reference's borrow rule,
    src/engine_impl.cpp:522-530); (2) if the sliced pipeline cannot be set up, Forward() serves the engine unsliced instead of
    failing, and keeps doing so."""
    mg = si.modelgen
    batch = 16
    pp, bp = _save(tmp_path, mg.build_yolov5s(batch, 96), "hf")
    x = mg.synth_input((batch, 96, 96, 3))
    e0, oname, ref = _run(si, pp, bp, x, host_slices=1)
    ref = ref.copy()
    e1 = si.Engine()                      # auto slices, no pinning
    e1.load_model(pp, bp)
    iname = e1.input_names()[0]
    for it in range(4):
        buf = x.copy()
        e1.input(iname, buf)              # drops (frees) the previous forward's buffer
        e1.forward(); e1.forward()
        assert_exact(e1.extract(oname), ref, "fresh buffer, round %d" % it)
        del buf
    e1.release()
    e2 = si.Engine(_fail_slicer=1)
    e2.load_model(pp, bp)
    e2.input(iname, x)
    for it in range(3):
        e2.forward()
        assert_exact(e2.extract(oname), ref, "unsliced fallback, forward %d" % it)
    e2.release()


def test_two_lanes_need_an_even_batch(si, tmp_path):
    pp, bp = _save(tmp_path, si.modelgen.build_toy_yolo(3, 64), "odd")
    e = si.Engine(streams=2)
    with pytest.raises(si.StatusError):
        e.load_model(pp, bp)
    e = si.Engine()      # auto: an odd batch simply runs on one stream
    e.load_model(pp, bp)


@pytest.mark.parametrize("name", ["yolov5s_160", "resnet18_small", "mobilenetv3_small_96", "toy_yolo"])
def test_activation_arena_changes_nothing_but_the_footprint(si, tmp_path, name):
    """Intermediate operands share one HBM arena by lifetime (the reference allocates every operand and never reuses,
    src/engine_impl.cpp:465-482): bit-identical outputs against one-allocation-per-operand, under every schedule, repeated
    forwards and hipGraph replay, at a fraction of the footprint."""
    mk, shape = MODELS[name]
    pp, bp = _save(tmp_path, mk(si.modelgen), name)
    x = si.modelgen.synth_input(shape)
    e0, oname, ref = _run(si, pp, bp, x, arena=0)
    for opts in ({}, {"fuse": 0, "alias_cat": 0}, {"graph": 1}, {"winograd": 0}):
        e1, _, got = _run(si, pp, bp, x, arena=1, **opts)
        assert_exact(got, _run(si, pp, bp, x, arena=0, **opts)[2], "arena vs per-operand allocations %s" % opts)
        for _ in range(3):
            e1.forward()
        assert_exact(e1.extract(oname), got, "repeated forwards %s" % opts)
    s0, s1 = e0.schedule(), _run(si, pp, bp, x)[0].schedule()
    assert s0["arena_bytes"] == s0["per_operand_bytes"] == s1["per_operand_bytes"]
    assert s1["arena_bytes"] < 0.5 * s1["per_operand_bytes"], s1
    print("%s: %.1f MB shared vs %.1f MB per operand" % (name, s1["arena_bytes"] / 1e6, s1["per_operand_bytes"] / 1e6))


def test_forward_is_repeatable_and_input_is_read_at_forward_time(si, tmp_path):
    pp, bp = _save(tmp_path, si.modelgen.build_toy_yolo(2, 64), "rep")
    x = si.modelgen.synth_input((2, 64, 64, 3))
    e, oname, a = _run(si, pp, bp, x)
    e.forward()
    assert_exact(e.extract(oname), a)
    # reference semantics: Input() borrows; the buffer is read when Forward() runs (bench_yolo.cpp:22-27)
    buf = e._inputs["0"]
    buf[...] = si.modelgen.synth_input((2, 64, 64, 3), seed=7)
    e.forward()
    b = e.extract(oname)
    assert not np.array_equal(a, b)
    e2, _, c = _run(si, pp, bp, buf.copy())
    assert_exact(b, c)


def test_hipgraph_replay_matches_eager(si, tmp_path):
    pp, bp = _save(tmp_path, si.modelgen.build_toy_yolo(2, 64), "graph")
    x = si.modelgen.synth_input((2, 64, 64, 3))
    _, oname, eager = _run(si, pp, bp, x)
    e = si.Engine(graph=1)
    e.load_model(pp, bp)
    e.input("0", x)
    for _ in range(4):  # eager warm-up, capture, replay, replay
        e.forward()
        assert_exact(e.extract(oname), eager)


def test_device_resident_io(si, tmp_path):
    from simpleinfer_amd import hipops
    pp, bp = _save(tmp_path, si.modelgen.build_toy_classifier(2, 32), "dev")
    x = si.modelgen.synth_input((2, 32, 32, 3))
    _, oname, ref = _run(si, pp, bp, x)
    e = si.Engine(outputs_to_host=0)
    e.load_model(pp, bp)
    dx = hipops.DeviceBuffer.from_numpy(x)
    e.input_device("0", dx.ptr)
    e.forward()
    ptr, on_dev = e.extract_ptr(oname)
    assert on_dev and ptr
    assert_exact(e.extract(oname), ref)


def test_batch_sharding_is_bit_exact(si, tmp_path):
    """section 8(e): the path shards by image with no halo -- a rank's slab equals the same rows of the full run."""
    mg = si.modelgen
    x = mg.synth_input((4, 64, 64, 3))
    pp, bp = _save(tmp_path, mg.build_toy_yolo(4, 64), "b4")
    _, oname, full = _run(si, pp, bp, x)
    pp2, bp2 = _save(tmp_path, mg.build_toy_yolo(2, 64), "b2")
    for r in range(2):
        _, _, part = _run(si, pp2, bp2, x[2 * r:2 * r + 2])
        assert_exact(part, full[2 * r:2 * r + 2])


def test_expression_graph_with_unary_and_scalar_ops(si, orc, tmp_path):
    """expressions that lower to UnaryOp and to BinaryOp's scalar / sub / div / pow forms
    (expand_expression.cpp:123-244).  The reference registers no UnaryOp layer and its BinaryOp::Init reads neither the scalar
    params nor any code but add / mul, so it fails LoadModel on such a file; here it loads, runs and matches the oracle."""
    mg = si.modelgen
    b = mg.PnnxBuilder(3)
    x = b.input((2, 3, 32, 32))
    c1 = mg._Conv(b, x, 16, 3, 1)
    c2 = mg._Conv(b, x, 16, 3, 1)
    y = b.expression("add(@0,mul(@1,2.0))", [c1, c2])                 # scalar mul feeding a tensor add
    y = b.expression("sqrt(add(pow(@0,2),1.0))", [y])                  # pow(x, 2) -> square; add scalar; UnaryOp sqrt
    z = b.expression("div(1.0,add(exp(neg(@0)),1.0))", [c1])           # a hand-written sigmoid: neg, exp, +1, reversed div
    y = b.expression("sub(@0,mul(@1,@2))", [y, z, c2])                  # three-operand expression, tensor sub
    y = mg._Conv(b, y, 8, 1, 1)
    b.output(b.expression("tanh(div(@0,4.0))", [y]))
    pp, bp = _save(tmp_path, b, "expr")
    xin = mg.synth_input((2, 32, 32, 3))
    ref = orc.run_graph(pp, bp, {"0": xin})
    e, oname, got = _run(si, pp, bp, xin)
    (want,) = ref.values()   # expression lowering renames the output operand on the engine side
    assert_parity(got, want, what="expression graph")
    kernels = [L["kernel"] for L in e.profile()]
    assert kernels.count("unary") >= 4 and "binary_scalar" in kernels and "binary" in kernels, kernels
    _, _, plain = _run(si, pp, bp, xin, fuse=0, alias_cat=0)
    assert_parity(plain, want, what="expression graph, plain schedule")


def test_errors_are_statuses(si, tmp_path):
    e = si.Engine()
    with pytest.raises(si.StatusError):
        e.load_model(str(tmp_path / "missing.param"), str(tmp_path / "missing.bin"))
    # an operator the reference does not register either -> kEmpty (engine_impl.cpp:247-250)
    b = si.modelgen.PnnxBuilder()
    x = b.input((1, 4, 8, 8))
    y = b._unary("nn.GELU", "gelu", x)
    b.output(y)
    pp, bp = _save(tmp_path, b, "bad")
    with pytest.raises(si.StatusError) as ei:
        e.load_model(pp, bp)
    assert ei.value.status == si.Status.kEmpty
    pp, bp = _save(tmp_path, si.modelgen.build_toy_classifier(1, 16), "ok")
    e.load_model(pp, bp)
    with pytest.raises(si.StatusError):
        e.forward()  # no input bound
    with pytest.raises(si.StatusError):
        e.input("nope", np.zeros((1, 16, 16, 3), np.float32))
    with pytest.raises(si.StatusError):
        e.input("0", np.zeros((1, 8, 8, 3), np.float32))
    e.release()
    assert e.input_names() == []


# ---- fp16 storage path (BASELINE.json configs[3]; no reference parity target: the yardstick is the fp32 oracle) ----
# Round 6: the bars sit ~10x above what is measured, per model -- they were 5e-3 / 2e-3 for everything, 100x above YOLOv5s'
# 3e-5 (a 100x regression would have passed).  YOLOv5s: box columns 3e-4 of their scale (measured 3e-5), scores 5e-4 absolute; everything else
# (ResNet18 logits measured 5e-4, the small single-purpose graphs): 2e-3 of the tensor's scale.
F16_YOLO_TOL = 3e-4
F16_GRAPH_TOL = 2e-3
F16_SCORE_ABS = 5e-4   # objectness / class scores of a Detect output, absolute (sigmoid outputs in (0, 1))


def _f16_check(name):
    if "yolo" in name:
        return lambda got, ref, what: assert_detect_parity(got, ref, F16_YOLO_TOL, F16_SCORE_ABS, what=what)
    return lambda got, ref, what: assert_parity(got, ref, F16_GRAPH_TOL, what=what)


@pytest.mark.parametrize("name", ["yolov5s_160", "resnet18_b32"])
def test_fp16_graph_vs_fp32_oracle(si, orc, tmp_path, name):
    mg = si.modelgen
    if name == "yolov5s_160":
        builder, shape = mg.build_yolov5s(2, 160), (2, 160, 160, 3)
    else:
        builder, shape = mg.build_resnet18(2, 64, num_classes=100, base=32), (2, 64, 64, 3)
    pp, bp = _save(tmp_path, builder, name)
    x = mg.synth_input(shape)
    ref = orc.run_graph(pp, bp, {"0": x})
    e, oname, got = _run(si, pp, bp, x, fp16=1)
    assert got.dtype == np.float32                       # Extract() hands out fp32 whatever the internal storage
    err = _f16_check(name)(got, ref[oname], name + " fp16")
    print("fp16 %s: %s" % (name, err))
    kernels = {L["kernel"] for L in e.profile()}
    assert any("f16" in k for k in kernels), kernels
    # fp16 internal tensors really are half the bytes: the plain schedule agrees with the fused one
    _, _, plain = _run(si, pp, bp, x, fp16=1, fuse=0, alias_cat=0)
    _f16_check(name)(plain, ref[oname], name + " fp16 unfused")
    # batch invariance still holds bit for bit
    e1, _, one = _run(si, *_save(tmp_path, mg.build_yolov5s(1, 160) if name == "yolov5s_160" else
                                  mg.build_resnet18(1, 64, num_classes=100, base=32), name + "_b1"), x[1:2], fp16=1)
    assert_exact(one[0], got[1], "fp16: an image's result does not depend on the batch")


def test_fp16_mobilenetv3_vs_fp32_oracle(si, orc, tmp_path):
    """MobileNetV3-Small with fp16 storage (round 3: depthwise convs on their own fp16 kernel, pointwise / squeeze-excite convs
    with 16 / 24 / 40 / 72 / 96 ... channels through a zero-padded K axis, the squeeze-excite scale as a broadcast multiply,
    Linear -> Hardswish -> Linear with an fp16 hidden vector): the logits against the fp32 oracle, every kernel family present,
    batch invariance bit for bit."""
    mg = si.modelgen
    pp, bp = _save(tmp_path, mg.build_mobilenetv3_small(2, 96, num_classes=100), "mnv3h")
    x = mg.synth_input((2, 96, 96, 3))
    ref = orc.run_graph(pp, bp, {"0": x})
    e, oname, got = _run(si, pp, bp, x, fp16=1)
    assert got.dtype == np.float32
    assert_parity(got, ref[oname], 5e-3, what="MobileNetV3-Small fp16 storage vs fp32 oracle")   # ~60 layers of fp16 roundings
    kernels = {L["kernel"].split("<")[0] for L in e.profile()}
    assert "conv_depthwise_f16_kernel" in kernels and "conv_igemm_f16_kernel" in kernels and "conv_stem_f16_kernel" in kernels, kernels
    _, _, plain = _run(si, pp, bp, x, fp16=1, fuse=0, alias_cat=0)
    assert_parity(plain, ref[oname], 5e-3, what="unfused")
    e1, _, one = _run(si, *_save(tmp_path, mg.build_mobilenetv3_small(1, 96, num_classes=100), "mnv3h_b1"), x[1:2], fp16=1)
    assert_exact(one[0], got[1], "fp16: an image's result does not depend on the batch")


@pytest.mark.parametrize("tail", ["silu_unfused", "maxpool", "cat", "add", "upsample"])
def test_fp16_graph_output_from_a_non_conv_layer(si, orc, tmp_path, tail):
    """Extract() stays fp32 when the last layer is not one that converts in its own epilogue: the engine appends a
    convert step behind a half staging operand (and concat aliasing still applies to that staging operand)."""
    mg = si.modelgen
    b = mg.PnnxBuilder(0)
    x = b.input((2, 3, 66, 66))
    y = mg._Conv(b, x, 32, 6, 2)
    opts = dict(fp16=1)
    if tail == "silu_unfused":
        opts["fuse"] = 0
    elif tail == "maxpool":
        y = b.maxpool(y, 5, 1, 2)
    elif tail == "cat":
        y = b.cat([mg._Conv(b, y, 32, 1), b.maxpool(y, 3, 1, 1)], 1)
    elif tail == "add":
        y = b.add(y, mg._Conv(b, y, 32, 3))
        opts["fuse"] = 0
    else:
        y = b.upsample(y, 2.0)
    b.output(y)
    pp, bp = _save(tmp_path, b, "tail_" + tail)
    xin = mg.synth_input((2, 66, 66, 3))
    ref = orc.run_graph(pp, bp, {"0": xin})
    e, oname, got = _run(si, pp, bp, xin, **opts)
    assert got.dtype == np.float32
    (want,) = ref.values()  # one output; expression lowering renames the operand on the engine side
    assert_parity(got, want, F16_GRAPH_TOL, what="fp16 graph ending in " + tail)
    sch = e.schedule()
    assert sch["run"][-1].endswith(".to_f32"), sch["run"]
    assert "convert_f16_f32" in {L["kernel"] for L in e.profile()}
    if tail == "cat":
        assert len(sch["alias"]) == 2, sch


@pytest.mark.parametrize("hw,c", [(72, 32), (12, 6)])
def test_pool_chain_falls_back_to_three_pools(si, orc, tmp_path, hw, c):
    """SPPF-shaped pool chains the fused kernel does not take (a map too large for LDS; a channel count that is not a
    multiple of 4) still run, as three launches, and stay exact."""
    mg = si.modelgen
    b = mg.PnnxBuilder(0)
    x = b.input((2, c, hw, hw))
    y1 = b.maxpool(x, 5, 1, 2)
    y2 = b.maxpool(y1, 5, 1, 2)
    y3 = b.maxpool(y2, 5, 1, 2)
    b.output(b.cat([x, y1, y2, y3], 1))
    pp, bp = _save(tmp_path, b, "chain_%d_%d" % (hw, c))
    xin = mg.synth_input((2, hw, hw, c))
    e, oname, got = _run(si, pp, bp, xin)
    assert_exact(got, orc.run_graph(pp, bp, {"0": xin})[oname], "pool chain fallback")
    assert e.schedule()["run"].count("maxpool_0") == 1 and "maxpool_1" in e.schedule()["fused"]


def test_yolov5s_at_another_input_size(si, orc, tmp_path):
    """416x416: 208-pixel stem rows (ragged 160-pixel column tiles), 13x13 / 26x26 / 52x52 maps (odd pixel counts per image,
    SPPF on a 13x13 map) -- fp32 against the oracle, the fp16 path against fp32."""
    mg = si.modelgen
    pp, bp = _save(tmp_path, mg.build_yolov5s(2, 416), "y416")
    x = mg.synth_input((2, 416, 416, 3))
    ref = orc.run_graph(pp, bp, {"0": x})
    e, oname, got = _run(si, pp, bp, x)
    assert got.shape == (2, 3 * (52 * 52 + 26 * 26 + 13 * 13), 85)
    assert_detect_parity(got, ref[oname], what="416x416 fp32")
    _, _, half = _run(si, pp, bp, x, fp16=1)
    assert_detect_parity(half, ref[oname], F16_YOLO_TOL, F16_SCORE_ABS, what="416x416 fp16")
    _, _, one = _run(si, *_save(tmp_path, mg.build_yolov5s(1, 416), "y416b1"), x[1:2])
    assert_exact(one[0], got[1], "416x416: an image's result does not depend on the batch")


def test_fp16_layers_without_an_fp16_kernel_run_in_fp32_between_casts(si, orc, tmp_path):
    """Round 4: an fp16 engine no longer refuses a graph because ONE layer has no fp16 kernel -- toy_yolo's 3x3
    convs whose channel counts are not multiples of 32 (a UnaryOp too until round 5 gave it an fp16 kernel) -- it runs that layer's fp32 kernel on fp32 shadows of its half
    operands (a cast step on either side, EngineImpl::InsertFp32Fallbacks) and keeps fp16 storage everywhere else.  The result
    meets the fp16 bar against the fp32 oracle, fused and unfused schedules alike, and does not depend on the batch."""
    mg = si.modelgen
    ub = mg.PnnxBuilder(0)
    ux = ub.input((2, 3, 32, 32))
    ub.output(ub.expression("sqrt(@0)", [mg._Conv(ub, ux, 32, 3, 2)]))
    # (round 5: a narrow YOLOv5's convs -- multiples of 8 channels -- have fp16 kernels now; what still has none is a conv over a
    # channel count that is not a multiple of 8: the 3x3 conv over 12 channels in 32 -> 12 -> 24 -> 16)
    ob = mg.PnnxBuilder(1)
    ox = ob.input((2, 3, 32, 32))
    oy = mg._Conv(ob, mg._Conv(ob, ox, 32, 3, 2), 12, 1, 1)
    ob.output(mg._Conv(ob, mg._Conv(ob, oy, 24, 3, 1), 16, 1, 1))
    for name, builder, shape in (("odd12", ob, (2, 32, 32, 3)), ("unary", ub, (2, 32, 32, 3))):
        pp, bp = _save(tmp_path, builder, name)
        x = mg.synth_input(shape)
        ref = orc.run_graph(pp, bp, {"0": x})
        e, oname, got = _run(si, pp, bp, x, fp16=1)
        assert got.dtype == np.float32
        check = _f16_check(name)
        if name == "unary":
            # sqrt amplifies the relative error of small arguments twofold in relative terms and this 32-channel graph's output scale is ~1:
            # measured 3.0e-3, the one graph of the suite above the 2e-3 bar
            check = lambda got, ref, what: assert_parity(got, ref, 6e-3, what=what)
        (want,) = ref.values()   # (expression lowering renames the output operand on the engine side)
        if name == "unary":
            # sqrt of a negative SiLU output is NaN on both sides; fp16 storage of the conv output may flip the sign of values
            # within rounding of zero, so the NaN pattern is compared where the fp32 value is clear of it
            clear = ~(np.abs(np.nan_to_num(want)) < 2e-2)
            assert np.array_equal(np.isnan(got)[clear], np.isnan(want)[clear])
            got = np.where(clear, np.nan_to_num(got), 0.0).astype(np.float32)
            want = np.where(clear, np.nan_to_num(want), 0.0).astype(np.float32)
        check(got, want, name + " fp16 with fp32 fallback layers")
        run = e.schedule()["run"]
        kernels = {L["kernel"] for L in e.profile()}
        if name == "unary":
            # (round 5: UnaryOp has an fp16 kernel of its own -- no fp32 shadow around it, only the cast in front of the graph output)
            assert not any(".in_to_f32." in n for n in run), run
        else:
            assert any(".in_to_f32." in n for n in run) and any(".out_to_f16." in n or ".to_f32" in n for n in run), run
        assert "convert_f16_f32" in kernels, kernels
        _, _, plain = _run(si, pp, bp, x, fp16=1, fuse=0, alias_cat=0)
        if name == "unary":
            plain = np.where(clear, np.nan_to_num(plain), 0.0).astype(np.float32)
        check(plain, want, name + " fp16 unfused")
        e32, _, got32 = _run(si, pp, bp, x)       # (the same file without the option: fp32 kernels only)
        assert not any("convert" in L["kernel"] for L in e32.profile())


def test_rebatch_serves_any_batch_from_one_file(si, tmp_path):
    """SetOption("batch", N): the file bakes its batch into every operand shape; the engine rewrites it."""
    mg = si.modelgen
    p1 = _save(tmp_path, mg.build_yolov5s(1, 160), "b1")
    p3 = _save(tmp_path, mg.build_yolov5s(3, 160), "b3")
    x = mg.synth_input((3, 160, 160, 3))
    _, oname, ref = _run(si, *p3, x)
    e = si.Engine(batch=3)
    e.load_model(*p1)
    assert e.operand_shape("0") == (3, 160, 160, 3) and e.operand_shape(oname)[0] == 3
    e.input("0", x)
    e.forward()
    assert_exact(e.extract(oname), ref, "batch-1 file served at batch 3 == batch-3 file")
    # and a classifier (rank-2 output, flatten / linear in the path)
    r1 = _save(tmp_path, mg.build_resnet18(1, 64, num_classes=10, base=16), "r1")
    r4 = _save(tmp_path, mg.build_resnet18(4, 64, num_classes=10, base=16), "r4")
    xr = mg.synth_input((4, 64, 64, 3))
    _, on, refr = _run(si, *r4, xr)
    _, _, got = _run(si, *r1, xr, batch=4)
    assert_exact(got, refr, "resnet re-batched")
    # the same rewrite as a file-to-file tool (si_pnnx_save: C++ loader -> Graph::save), expressions lowered on the way
    q3 = (str(tmp_path / "q3.pnnx.param"), str(tmp_path / "q3.pnnx.bin"))
    si.engine.pnnx_save(*p1, *q3, expand=True, batch=3)
    _, _, got3 = _run(si, *q3, x)
    assert_exact(got3, ref, "batch-1 file rewritten to batch 3 by the C++ writer")


def test_output_binding_writes_into_caller_memory(si, tmp_path):
    """Engine::Output (extension): an output operand is written into caller-owned device memory; alternating two buffers is
    what the overlapped multi-GPU all-gather does.  Also under hipGraph replay (pointers are baked into the capture)."""
    from simpleinfer_amd import hipops
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(2, 160), "ob")
    x = si.modelgen.synth_input((2, 160, 160, 3))
    _, oname, ref = _run(si, pp, bp, x)
    for graph in (0, 1):
        e = si.Engine(outputs_to_host=0, graph=graph)
        e.load_model(pp, bp)
        e.input("0", x)
        bufs = [hipops.DeviceBuffer(ref.nbytes) for _ in range(2)]
        for step in range(5):
            b = bufs[step & 1]
            b.fill(0)
            e.bind_output(oname, b.ptr)
            e.forward()
            assert e.extract_ptr(oname)[0] == b.ptr
            assert_exact(b.to_numpy(ref.shape), ref, "graph=%d step %d" % (graph, step))
        e.bind_output(oname, None)                      # back to the engine's own buffer
        e.forward()
        assert e.extract_ptr(oname)[0] not in (bufs[0].ptr, bufs[1].ptr)
        assert_exact(e.extract(oname), ref, "engine-owned buffer again")
    with pytest.raises(si.StatusError):
        e.bind_output("no_such_operand", bufs[0].ptr)


def test_demo_pipeline_matches_oracle_pipeline(si, orc, tmp_path):
    """examples/yolo_demo.py (letterbox -> Forward -> post-processing, all on the device) against the same pipeline on
    the oracle: packing exact, network within the fp32 bar, and -- fed the SAME predictions -- identical boxes."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("yolo_demo", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "yolo_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    shapes = ((120, 160), (270, 200))
    x, adjust, pred, dets, counts = demo.run(size=160, images=shapes, seed=3)
    rng = np.random.Generator(np.random.Philox(3))
    for b, (h, w) in enumerate(shapes):                     # the oracle's letterbox on the same synthetic pixels
        hr, wr, scale, pt, pl = orc.letterbox_geometry(h, w, 160, 160)
        resized = rng.integers(0, 256, (hr, wr, 3), dtype=np.uint8)
        assert_exact(x[b], orc.letterbox(resized, 160, 160, pt, pl), "letterbox image %d" % b)
        assert tuple(adjust[b]) == (pl, pt, np.float32(scale), w, h)
    pp, bp = _save(tmp_path, si.modelgen.build_yolov5s(2, 160), "demo")
    ref = orc.run_graph(pp, bp, {"0": x})
    assert_detect_parity(pred, list(ref.values())[0], what="demo forward")
    # The letterbox padding is a constant region, so a random-init network repeats itself there: EQUAL confidences, whose
    # order the reference leaves to an unstable quicksort.  Rows whose confidence is not unique are silenced before the
    # two post-processing implementations are compared (tie handling has its own test in test_gpu_ops.py).
    from simpleinfer_amd import hipops
    cls = pred[..., 5:]
    conf = pred[..., 4] * cls.max(-1)
    pred_u = pred.copy()
    for b in range(pred.shape[0]):
        vals, inv, cnt = np.unique(conf[b], return_inverse=True, return_counts=True)
        pred_u[b, cnt[inv] > 1, 4] = 0.0
    dets_u, counts_u = hipops.yolo_postprocess(pred_u, 0.25, 0.45, adjust=adjust)
    rdets, rcounts = orc.yolo_postprocess(pred_u, 0.25, 0.45, False, adjust)
    assert list(counts_u) == list(rcounts) and all(c > 0 for c in rcounts)
    for b in range(2):
        assert_exact(dets_u[b], rdets[b], "boxes image %d" % b)
        assert (dets_u[b][:, 0] >= 0).all() and (dets_u[b][:, 0] + dets_u[b][:, 2] <= shapes[b][1] - 1 + 1e-3).all()
    assert len(dets) == 2 and all(len(d) <= 300 for d in dets)   # the demo's own capped result


def test_full_size_properties_yolov5s_640_batch32(si, orc, tmp_path):
    """BASELINE.json's metric configuration itself (YOLOv5s 640x640 fp32 batch 32), where the CPU oracle is too slow to
    run the whole batch: size-independent properties instead.
      * sharding / batch invariance: image k of the batch-32 forward is bit-identical to the batch-1 engine on image k
        (the re-batch option makes both engines from ONE file), for images spread over the batch;
      * one image is checked against the oracle end to end (1e-4);
      * outputs are finite, confidences and class scores are sigmoid outputs in (0, 1), box sizes positive;
      * the schedule is the advertised one: 57 launches."""
    mg = si.modelgen
    pp, bp = _save(tmp_path, mg.build_yolov5s(1, 640), "y640")
    x = mg.synth_input((32, 640, 640, 3))
    e32 = si.Engine(batch=32)
    e32.load_model(pp, bp)
    oname = e32.output_names()[0]
    e32.input("0", x)
    e32.forward()
    full = e32.extract(oname)
    assert full.shape == (32, 25200, 85) and np.isfinite(full).all()
    assert (full[..., 4:] > 0).all() and (full[..., 4:] < 1).all() and (full[..., 2:4] > 0).all()
    # 51 launching layers (1 stem + 48 conv + Detect + 1 fused pool chain; Detect is 3 launches -> 53 launches; the two upsamples
    # are read at the source by the convs behind the concat) + 13 no-op cats
    run = e32.schedule()["run"]
    assert len([r for r in run if not r.startswith("cat")]) == 51 and len(run) == 64
    assert sorted(n for n in e32.schedule()["fused"] if n.startswith("upsample")) == ["upsample_0", "upsample_1"]
    e1 = si.Engine()
    e1.load_model(pp, bp)
    for k in (0, 13, 31):
        e1.input("0", x[k:k + 1])
        e1.forward()
        assert_exact(e1.extract(oname)[0], full[k], "image %d: batch 32 == batch 1" % k)
    ref = orc.run_graph(pp, bp, {"0": x[13:14]})[oname]
    assert_detect_parity(full[13:14], ref, what="image 13 of the batch vs the oracle")
    # same batch through the fp16 storage path: within the fp16 bar of the fp32 result
    e16 = si.Engine(batch=32, fp16=1)
    e16.load_model(pp, bp)
    e16.input("0", x)
    e16.forward()
    assert_detect_parity(e16.extract(oname), full, F16_YOLO_TOL, F16_SCORE_ABS, what="fp16 storage vs fp32 at full size")
    e16.release()
    # ... and through the opt-in f32_split arithmetic (round 6, VERDICT r05 item 2b: round 5 tested it at 160x160 batch 2 only): image 13 against
    # the oracle at the UNCHANGED fp32 bars, nothing tripped the range guard, batch invariance bit for bit
    es = si.Engine(batch=32, f32_split=1)
    es.load_model(pp, bp)
    es.input("0", x)
    es.forward()
    split = es.extract(oname)
    assert_detect_parity(split[13:14], ref, what="f32_split: image 13 of the batch vs the oracle")
    assert_detect_parity(split, full, what="f32_split vs true fp32 at full size")
    sch = es.schedule()
    assert sch["split_reruns"] == 0 and sch["split_demoted"] == [], sch
    assert sum(L["kernel"] in ("conv_split3_f32_kernel", "conv_wino23s_kernel") for L in es.profile()) >= 20
    es1 = si.Engine(f32_split=1)
    es1.load_model(pp, bp)
    es1.input("0", x[13:14])
    es1.forward()
    assert_exact(es1.extract(oname)[0], split[13], "f32_split: image 13 alone == in the batch")


def test_full_size_properties_resnet18_224_batch64(si, orc, tmp_path):
    """BASELINE.json configs[2] (ResNet18 224x224 fp32 batch 64) at full size -- the 7x7 s2 stem row kernel, the k3 s2
    max pool on 112x112, thirteen 56/28/14/7-pixel Winograd layers, the 1x1 s2 downsample convs, global average pool
    (adaptive_avg_pool_2d.cpp:54-116), flatten (flatten.cpp:55-88) and the Linear head (linear.cpp:74-117) end to end.
    The oracle is too slow for 64 images, so: images spread over the batch are bit-identical to the batch-1 engine made
    from the SAME file, one image is checked against the oracle (1e-4), fp16 storage stays within its bar of fp32, and the
    schedule is the advertised one."""
    mg = si.modelgen
    pp, bp = _save(tmp_path, mg.build_resnet18(1, 224), "r224")
    x = mg.synth_input((64, 224, 224, 3))
    e64 = si.Engine(batch=64)
    e64.load_model(pp, bp)
    oname = e64.output_names()[0]
    e64.input("0", x)
    e64.forward()
    full = e64.extract(oname)
    assert full.shape == (64, 1000) and np.isfinite(full).all()
    prof = e64.profile()
    kernels = [L["kernel"] for L in prof]
    assert sum("conv_wino23" in k for k in kernels) == 13, kernels          # every 3x3 s1 conv
    assert sum("conv_stem_roll" in k for k in kernels) == 1, kernels         # the 7x7 s2 stem (rolling-window kernel)
    assert sum("conv_igemm_f32_fast" in k for k in kernels) == 6, kernels    # 3 3x3 s2 + 3 1x1 s2 downsample convs
    assert prof[-1]["type"] == "nn.Linear" and prof[-1]["kernel"].startswith("conv_igemm"), prof[-1]   # the head runs as a 1x1 conv
    s = e64.schedule()
    assert sum(n.startswith("relu_") for n in s["fused"]) == 17 and sum(n.startswith("add_") for n in s["fused"]) == 8
    e1 = si.Engine()
    e1.load_model(pp, bp)
    for k in (0, 31, 63):
        e1.input("0", x[k:k + 1])
        e1.forward()
        assert_exact(e1.extract(oname)[0], full[k], "image %d: batch 64 == batch 1" % k)
    ref = orc.run_graph(pp, bp, {"0": x[31:32]})[oname]
    assert_parity(full[31:32], ref, what="image 31 of the batch vs the oracle")
    # the constant-input known-answer style of the reference's classifier demo (test_classify.cpp:25 feeds 2.0)
    xc = np.full((1, 224, 224, 3), 2.0, np.float32)
    e1.input("0", xc)
    e1.forward()
    assert_parity(e1.extract(oname), orc.run_graph(pp, bp, {"0": xc})[oname], what="constant 2.0 input")
    e16 = si.Engine(batch=64, fp16=1)
    e16.load_model(pp, bp)
    e16.input("0", x)
    e16.forward()
    assert_parity(e16.extract(oname), full, F16_GRAPH_TOL, what="fp16 storage vs fp32 at full size")


@pytest.mark.parametrize("module", ["simpleinfer", "simpleinfer_pybind"])
def test_reference_python_module_call_sequence(si, orc, tmp_path, module):
    """the reference's Python surface (python/pybind11_main.cpp:13-68), name for name, in `python/simpleinfer.py` (ctypes over
    the C-ABI) and -- round 6 -- as a COMPILED pybind11 module over the C++ Engine / Tensor (`python/pybind11_main.cpp` ->
    simpleinfer_pybind, zero-copy SetTensorDim4 / GetTensorDim4).
    The call sequence of the reference's binding smoke test (InitializeContext, Engine, LoadModel, InputNames / OutputNames,
    Tensor(DataType.Float32, shape), SetTensorDim4(constant 42.0 image), Input, Forward, Extract into an empty Tensor,
    GetTensorDim4) on a synthesized narrow YOLOv5 -- checked against the oracle, which the reference's own test never does."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "python"))
    import importlib
    infer = importlib.import_module(module)
    pp, bp = _save(tmp_path, si.modelgen.build_toy_yolo(4, 96), "py")
    infer.InitializeContext()
    engine = infer.Engine()
    assert engine.LoadModel(str(tmp_path / "missing.param"), str(tmp_path / "missing.bin")) != infer.Status.Success
    assert engine.LoadModel(pp, bp) == infer.Status.Success
    input_names, output_names = engine.InputNames(), engine.OutputNames()
    assert input_names == ["0"] and len(output_names) == 1
    input_shape = [4, 96, 96, 3]                                     # NHWC
    input_np = np.ones(input_shape, dtype=np.float32) * 42.0
    input_tensor = infer.Tensor(infer.DataType.Float32, input_shape)
    assert input_tensor.GetDataType() == infer.DataType.Float32 and input_tensor.Shape() == input_shape
    assert input_tensor.SetTensorDim4(input_np) == infer.Status.Success
    assert engine.Input(input_names[0], input_tensor) == infer.Status.Success
    assert engine.Input("nope", input_tensor) == infer.Status.Fail
    assert engine.Forward() == infer.Status.Success
    output_tensor = infer.Tensor()
    assert output_tensor.GetDataType() == getattr(infer.DataType, "None")
    assert engine.Extract(output_names[0], output_tensor) == infer.Status.Success
    output_np = output_tensor.GetTensorDim4()
    rows = 3 * (12 * 12 + 6 * 6 + 3 * 3)
    assert output_np.dtype == np.float32 and output_np.shape == (1, 4, rows, 8) and output_tensor.Shape() == [4, rows, 8]
    ref = orc.run_graph(pp, bp, {"0": input_np})[output_names[0]]
    assert_detect_parity(output_np[0], ref, what="constant-42 image through the reference-shaped module")
    # borrow semantics: the array is read when Forward() runs, not when Input() was called (bench_yolo.cpp:22-27)
    input_np[...] = si.modelgen.synth_input(tuple(input_shape), seed=5)
    assert engine.Forward() == infer.Status.Success
    assert engine.Extract(output_names[0], output_tensor) == infer.Status.Success
    assert_detect_parity(output_tensor.GetTensorDim4()[0], orc.run_graph(pp, bp, {"0": input_np})[output_names[0]], what="second forward")
    wrong = infer.Tensor(infer.DataType.Float32, [1, 8, 8, 3])
    wrong.SetTensorDim4(np.zeros([1, 8, 8, 3], np.float32))
    assert engine.Input(input_names[0], wrong) == infer.Status.ErrorShape
    assert engine.Release() == infer.Status.Success and engine.InputNames() == []
