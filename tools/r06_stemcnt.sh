#!/bin/bash
# tools/r06_stemcnt.sh (GPU box): the fp32 stem kernel with countable memory operations (conv_stem_roll.hip CNT) against the previous commit's library
# (build_variants/prev/): bits, the kernel alone, the headline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -q -x -k "stem" 2>&1 | tail -2
python3 - <<'PY'
import os, subprocess, sys, numpy as np
code = r'''
import numpy as np, sys
from simpleinfer_amd import hipops
rng = np.random.default_rng(5)
outs = []
for (n, h, w, oc, k, p, act) in [(3, 128, 640, 32, 6, 2, "silu"), (2, 224, 224, 64, 7, 3, "relu"), (2, 96, 96, 16, 3, 1, "hardswish"), (1, 70, 330, 32, 6, 2, "silu")]:
    x = rng.random((n, h, w, 3), dtype=np.float32); wt = (rng.random((oc, 3, k, k), dtype=np.float32) - 0.5) * 0.3; b = rng.random(oc, dtype=np.float32)
    outs.append(hipops.conv2d(x, wt, b, (2, 2), (p, p), act1=act))
np.savez(sys.argv[1], *outs)
'''
env = dict(os.environ)
subprocess.run([sys.executable, "-c", code, "/tmp/new.npz"], check=True, env=env)
env["SI_HIP_LIB"] = os.getcwd() + "/build_variants/prev/libsi_hip.so"
subprocess.run([sys.executable, "-c", code, "/tmp/prev.npz"], check=True, env=env)
a, b = np.load("/tmp/new.npz"), np.load("/tmp/prev.npz")
print("fp32 stem, same bits as the previous library:", all(np.array_equal(a[k], b[k]) for k in a.files), [a[k].shape for k in a.files])
PY
for rep in 1 2; do
  SI_HIP_LIB=$PWD/build_variants/prev/libsi_hip.so python3 tools/stem_bench.py --which f32 | sed 's/^/prev /'
  python3 tools/stem_bench.py --which f32 | sed 's/^/new  /'
done
bash tools/ab_prev.sh "" 3
