"""CPU: the product's pnnx loader (C++, through the C-ABI si_pnnx_dump) against the REFERENCE's own
loader -- live (oracle/_ref/ref_pnnx_dump, compiled from /root/reference/src/pnnx) when that binary is
present, and always against the committed dumps it produced (tests/golden/*.refdump.txt)."""
import os
import subprocess

import pytest

from simpleinfer_amd import engine, modelgen as mg

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_pnnx_dump")

MODELS = {"toy_yolo": lambda: mg.build_toy_yolo(2, 64), "toy_classifier": lambda: mg.build_toy_classifier(2, 32)}


def _ours(tmp_path, builder, expand):
    pp, bp, out = str(tmp_path / "m.pnnx.param"), str(tmp_path / "m.pnnx.bin"), str(tmp_path / "dump.txt")
    builder.save(pp, bp)
    engine.pnnx_dump(pp, bp, expand, out)
    return pp, bp, open(out).read()


@pytest.mark.parametrize("name", sorted(MODELS))
@pytest.mark.parametrize("expand", [False, True])
def test_loader_matches_committed_reference_dump(native_libs, tmp_path, name, expand):
    _, _, ours = _ours(tmp_path, MODELS[name](), expand)
    gold = open(os.path.join(HERE, "golden", "%s.%s.refdump.txt" % (name, "expanded" if expand else "raw"))).read()
    assert ours == gold


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref not built (reference sources absent)")
@pytest.mark.parametrize("which", ["yolov5s", "resnet18", "toy_classifier"])
def test_loader_matches_live_reference_loader(native_libs, tmp_path, which):
    b = {"yolov5s": lambda: mg.build_yolov5s(1, 64), "resnet18": lambda: mg.build_resnet18(1, 32),
         "toy_classifier": lambda: mg.build_toy_classifier(1, 16)}[which]()
    for expand in (False, True):
        pp, bp, ours = _ours(tmp_path, b, expand)
        ref = subprocess.run([REF_BIN, pp, bp] + (["--expand"] if expand else []), check=True, capture_output=True,
                             text=True).stdout
        assert ours == ref


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref not built")
def test_nested_and_scalar_expressions_lower_like_the_reference(native_libs, tmp_path):
    """expand_expression corner cases: nesting, scalar operands, pow->square, unsupported tokens."""
    b = mg.PnnxBuilder()
    x = b.input((1, 4, 8, 8))
    y = b.relu(x)
    for expr in ("add(@0,mul(@1,2.0))", "sub(3.0,@0)", "pow(@0,2)", "div(@0,@1)", "sqrt(add(@0,@1))", "mul(@0,size(@1,2))"):
        z = b.expression(expr, [x, y])
        x = b.relu(z)
    b.output(x)
    pp, bp, ours = _ours(tmp_path, b, True)
    ref = subprocess.run([REF_BIN, pp, bp, "--expand"], check=True, capture_output=True, text=True).stdout
    assert ours == ref
    assert "pnnx.Expression" in ours  # the size() expression is left in place, as the reference leaves it


def test_parameter_value_syntax(native_libs, tmp_path):
    """value classification of reference ir.cpp:479-550 via a hand-written .param"""
    pp, bp, out = tmp_path / "m.pnnx.param", tmp_path / "m.pnnx.bin", tmp_path / "d.txt"
    import zipfile
    zipfile.ZipFile(bp, "w").close()
    pp.write_text("7767517\n2 1\npnnx.Input in0 0 1 0 #0=(1,?,4)f32\n"
                  "nn.ReLU r0 1 0 0 a=None b=True c=-3 d=1.5e-1 e=zeros f=(1,2) g=(2.0,3.5) h=(x,y) i=() j=-x k=7. #0=(1,?,4)f32\n")
    engine.pnnx_dump(str(pp), str(bp), False, str(out))
    txt = out.read_text()
    for line in ("param a type=0", "param b type=1 b=1", "param c type=2 i=-3", "param d type=3 f=0.150000006",
                 "param e type=4 s=zeros", "param f type=5 ai=1,2,", "param g type=6 af=2,3.5,", "param h type=7 as=x,y,",
                 "param i type=0", "param j type=4 s=-x", "param k type=3 f=7", "operand 0 type=1 shape=1,-1,4,"):
        assert line in txt, line


def test_malformed_files_are_rejected_not_fatal(native_libs):
    """The reference's loader throws (std::stoi / map::at) or indexes blindly on malformed files; behind a C-ABI that is
    a crash.  Mutated model files must come back as a return code (tests/fuzz_loader.py; ASan build: tools/asan_host.sh)."""
    from fuzz_loader import fuzz
    ok, bad = fuzz(160, seed=3)
    assert ok + bad == 320 and bad > 0


# ---- writer: Graph::save + StoreZipWriter through si_pnnx_save ----------------------------------
WRITER_MODELS = {"yolov5s": lambda: mg.build_yolov5s(2, 64), "resnet18": lambda: mg.build_resnet18(2, 32),
                 "mobilenetv3": lambda: mg.build_mobilenetv3_small(2, 64, num_classes=10),
                 "toy_classifier": lambda: mg.build_toy_classifier(2, 16)}


def _dump(pp, bp, expand, out):
    engine.pnnx_dump(pp, bp, expand, out)
    return open(out).read()


@pytest.mark.parametrize("which", sorted(WRITER_MODELS))
def test_saved_model_loads_back_to_the_same_graph(native_libs, tmp_path, which):
    import zipfile
    pp, bp = str(tmp_path / "a.pnnx.param"), str(tmp_path / "a.pnnx.bin")
    WRITER_MODELS[which]().save(pp, bp)
    for expand in (False, True):
        qp, qb = str(tmp_path / ("b%d.pnnx.param" % expand)), str(tmp_path / ("b%d.pnnx.bin" % expand))
        engine.pnnx_save(pp, bp, qp, qb, expand=expand)
        want = _dump(pp, bp, expand, str(tmp_path / "d0.txt"))
        got = _dump(qp, qb, False, str(tmp_path / "d1.txt"))            # params, attrs (hashed bytes), shapes, wiring
        if expand:  # lowering appends its new operands at the end of the list; a reload lists them in file order
            key = lambda t: sorted(ln for ln in t.splitlines() if ln.startswith("operand "))
            rest = lambda t: [ln for ln in t.splitlines() if not ln.startswith("operand ")]
            assert key(got) == key(want) and rest(got) == rest(want)
        else:
            assert got == want
        # any unzip reads the container: CRCs check out and the entries are the original bytes
        with zipfile.ZipFile(bp) as za, zipfile.ZipFile(qb) as zb:
            assert zb.testzip() is None
            assert all(i.compress_type == zipfile.ZIP_STORED for i in zb.infolist())
            if not expand:
                assert sorted(za.namelist()) == sorted(zb.namelist())
            for n in zb.namelist():
                if n in za.namelist():
                    assert za.read(n) == zb.read(n), n
        if os.path.exists(REF_BIN):  # and the reference's own loader reads what we wrote
            ref = subprocess.run([REF_BIN, qp, qb], check=True, capture_output=True, text=True).stdout
            assert ref == got


def test_save_rebatches_every_operand(native_libs, tmp_path):
    pp, bp = str(tmp_path / "a.pnnx.param"), str(tmp_path / "a.pnnx.bin")
    mg.build_yolov5s(2, 64).save(pp, bp)
    qp, qb = str(tmp_path / "b.pnnx.param"), str(tmp_path / "b.pnnx.bin")
    engine.pnnx_save(pp, bp, qp, qb, batch=7)
    before = _dump(pp, bp, False, str(tmp_path / "d0.txt")).splitlines()
    after = _dump(qp, qb, False, str(tmp_path / "d1.txt")).splitlines()
    assert len(before) == len(after)
    n_shapes = 0
    for a, b in zip(before, after):
        if a.startswith("operand "):
            n_shapes += 1
            assert a.replace("shape=2,", "shape=7,") == b
        else:
            assert a == b
    assert n_shapes > 100
    with pytest.raises(RuntimeError):
        engine.pnnx_save(pp + ".missing", bp, qp, qb)


def test_saved_float_parameters_keep_their_bits(native_libs, tmp_path):
    import zipfile
    pp, bp = tmp_path / "m.pnnx.param", tmp_path / "m.pnnx.bin"
    zipfile.ZipFile(bp, "w").close()
    pp.write_text("7767517\n2 1\npnnx.Input in0 0 1 0 #0=(1,?,4)f32\n"
                  "nn.ReLU r0 1 0 0 a=None b=True c=-3 d=1.5e-1 e=zeros f=(1,2) g=(2.0,3.5,1e-05,0.1) h=(x,y) k=7. "
                  "eps=1.0000000e-05 third=0.333333343 big=3.4e38 $input=0 #0=(1,?,4)f32\n")
    qp, qb = str(tmp_path / "q.pnnx.param"), str(tmp_path / "q.pnnx.bin")
    engine.pnnx_save(str(pp), str(bp), qp, qb)
    assert _dump(qp, qb, False, str(tmp_path / "d1.txt")) == _dump(str(pp), str(bp), False, str(tmp_path / "d0.txt"))
    assert "$input=0" in open(qp).read()
