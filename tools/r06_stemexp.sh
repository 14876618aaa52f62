#!/bin/bash
# tools/r06_stemexp.sh (GPU box): the RGB stem's three kernels on YOLOv5s's shape (tools/stem_bench.py), the split form's ablations (experiment build), and
# the split stem's tests
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests -m gpu -q -x -k "stem_split3 or f32_split_stem" 2>&1 | tail -8
python3 tools/stem_bench.py
export SI_HIP_LIB=$PWD/build_variants/libsi_hip_exp.so
for e in 0 1 2 3 4 7; do SI_STEM_EXP=$e python3 tools/stem_bench.py --which split; done
