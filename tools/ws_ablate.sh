#!/bin/bash
# tools/ws_ablate.sh (GPU box): compile-time ablations of conv_wino23s_kernel (SI_WS_ABL bits; libraries built here by tools/hip_variant.sh):
# time per launch with one phase removed each (wrong results, timing only)
echo "== product build"; python tools/split3_check.py --wino 2>&1 | grep -E "40x40x128|28x28x128|80x80x64"
for m in 1 2 4 8 16 32 63; do
  echo "== SI_WS_ABL=$m"
  SI_HIP_LIB=build_variants/libsi_hip_wsabl$m.so python tools/split3_check.py --wino 2>&1 | grep -E "40x40x128|28x28x128|80x80x64"
done
