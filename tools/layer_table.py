"""tools/layer_table.py [--fp16 1] [--batch 32] [--size 640] -- per-layer time / TFLOP/s / GB/s of one YOLOv5s forward
(HIP events around every launch, averaged over a few forwards); the table behind DESIGN.md's "where the time goes"."""
import argparse, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simpleinfer_amd as si

ap = argparse.ArgumentParser()
ap.add_argument("--fp16", type=int, default=0)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--model", default="yolov5s")
ap.add_argument("--passes", type=int, default=5)
a = ap.parse_args()
mg = si.modelgen
with tempfile.TemporaryDirectory() as td:
    pp, bp = os.path.join(td, "m.param"), os.path.join(td, "m.bin")
    b = mg.build_yolov5s(1, a.size) if a.model == "yolov5s" else mg.build_resnet18(1, a.size)
    b.save(pp, bp)
    e = si.Engine(fp16=a.fp16, batch=a.batch, outputs_to_host=0)
    e.load_model(pp, bp)
    e.input("0", mg.synth_input((a.batch, a.size, a.size, 3)))
    e.forward()
    acc = None
    for _ in range(a.passes):
        e.forward()
        p = e.profile()
        if acc is None:
            acc = p
        else:
            for x, y in zip(acc, p):
                x["ms"] += y["ms"]
    tot = 0.0
    for L in acc:
        L["ms"] /= a.passes
        tot += L["ms"]
    print("%-28s %-44s %8s %8s %8s" % ("layer", "kernel", "ms", "TFLOP/s", "GB/s"))
    for L in acc:
        print("%-28s %-44s %8.4f %8.1f %8.0f" % (L["name"][:28], L["kernel"][:44], L["ms"], L["flops"] / L["ms"] / 1e9 if L["ms"] else 0,
                                                 L["bytes"] / L["ms"] / 1e6 if L["ms"] else 0))
    print("total %.3f ms" % tot)
