"""CPU: the pnnx model synthesizer produces the graphs BASELINE.json names."""
import numpy as np

from simpleinfer_amd import modelgen as mg


def test_yolov5s_work_matches_survey():
    b = mg.build_yolov5s(1, 640)
    # 8.217 GMAC = 16.43 GFLOP per 640x640 image
    assert abs(mg.conv_flops(b) / 1e9 - 16.43) < 0.02
    types = [ln.split()[0] for ln in b.lines]
    assert types.count("nn.Conv2d") == 57 and types.count("nn.SiLU") == 57
    assert types.count("pnnx.Expression") == 7 and types.count("torch.cat") == 13
    assert types.count("nn.MaxPool2d") == 3 and types.count("nn.Upsample") == 2
    assert b.shapes[str(b._n_operand - 1)] == (1, 25200, 85)
    assert mg.conv_flops(mg.build_yolov5s(32, 640)) == 32 * mg.conv_flops(b)


def test_resnet18_work_matches_survey():
    b = mg.build_resnet18(1, 224)
    assert abs(mg.conv_flops(b) / 1e9 - 3.63) < 0.01
    types = [ln.split()[0] for ln in b.lines]
    assert types.count("nn.Conv2d") == 20 and types.count("nn.ReLU") == 17 and types.count("pnnx.Expression") == 8


def test_weights_are_portable_and_seeded():
    a = mg.seeded_uniform("conv_0.weight", (4, 3), -1, 1)
    b = mg.seeded_uniform("conv_0.weight", (4, 3), -1, 1)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, mg.seeded_uniform("conv_1.weight", (4, 3), -1, 1))
    # known answer: splitmix64 stream for seed 0 (portable across numpy versions / machines)
    u = mg.splitmix_uniform(0, 3)
    assert np.allclose(u, [0.8833108, 0.43152797, 0.02643377], atol=1e-6), u


def test_mobilenetv3_small_topology():
    b = mg.build_mobilenetv3_small(1, 224)
    types = [ln.split()[0] for ln in b.lines]
    # torchvision mobilenet_v3_small: 11 inverted-residual blocks, 9 of them with squeeze-excite, ~56.5 MMAC of conv work
    assert types.count("nn.AdaptiveAvgPool2d") == 9 + 1 and types.count("nn.Hardsigmoid") == 9
    depthwise = [ln for ln in b.lines if ln.startswith("nn.Conv2d") and "groups=1 " not in ln + " "]
    assert len(depthwise) == 11
    assert 0.10 < mg.conv_flops(b) / 1e9 < 0.13
    assert b.shapes[str(b._n_operand - 1)] == (1, 1000)
