#!/bin/bash
# tools/wino_ablate.sh -- compile-time ablations of conv_wino23.hip (build_variants/libsi_hip_w<bits>.so built with
# tools/build_exp.sh w<bits> "SI_WINO_ABLATE=<bits>": 1 no patch prefetch after block 0, 2 no filter loads after the first,
# 4 no commit writes after block 0, 8 no output stores) on the YOLOv5s Winograd shapes, with the phase stamps of every variant
for f in build_variants/libsi_hip_diag.so build_variants/libsi_hip_w*.so; do
  echo "== $f"; SI_HIP_LIB=$f python tools/conv_diag.py --algo wino 2>&1 | grep -E "back-to-back|cycles per|busy"
done
