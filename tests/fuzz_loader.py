"""Mutation fuzz of the pnnx loader (param text + stored-zip bin): truncations and byte flips of valid model files must be
accepted or rejected with a status -- never crash, throw across the C-ABI, or read out of bounds (run under
tools/asan_host.sh for the last part).  `python tests/fuzz_loader.py [iterations]`; tests/test_pnnx_loader.py runs a
short version."""
import os
import random
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fuzz(iterations=200, seed=1):
    from simpleinfer_amd import _native, modelgen as mg
    L = _native.host()
    rnd = random.Random(seed)
    ok = bad = 0
    with tempfile.TemporaryDirectory() as td:
        fp, fb, out = os.path.join(td, "f.param"), os.path.join(td, "f.bin"), os.path.join(td, "o.txt")
        for builder in (mg.build_toy_classifier(1, 32), mg.build_toy_yolo(1, 64)):
            pp, bp = os.path.join(td, "m.param"), os.path.join(td, "m.bin")
            builder.save(pp, bp)
            param, binb = open(pp, "rb").read(), open(bp, "rb").read()
            for it in range(iterations // 2):
                p, b = bytearray(param), bytearray(binb)
                mode = it % 4
                if mode == 0:
                    p = p[:rnd.randrange(0, len(p))]
                elif mode == 1:
                    for _ in range(rnd.randrange(1, 8)):
                        p[rnd.randrange(len(p))] = rnd.randrange(32, 127)
                elif mode == 2:
                    b = b[:rnd.randrange(0, len(b))]
                else:
                    for _ in range(rnd.randrange(1, 16)):
                        b[rnd.randrange(len(b))] = rnd.randrange(256)
                open(fp, "wb").write(p)
                open(fb, "wb").write(b)
                for expand in (0, 1):
                    rc = L.si_pnnx_dump(fp.encode(), fb.encode(), expand, out.encode())
                    ok += rc == 0
                    bad += rc != 0
    return ok, bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    ok, bad = fuzz(n)
    print("fuzz: %d loads succeeded, %d rejected, no crash" % (ok, bad))
