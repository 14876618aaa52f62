#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r04_s2poly; mkdir -p $O
SH="--shape 32,80,80,128,256,3,2,1 --shape 32,40,40,256,512,3,2,1"
for m in 0 1 2 3 7; do
  lib=simpleinfer_amd/libsi_hip.so; [ $m != 0 ] && lib=build_variants/libsi_hip_polyabl$m.so
  echo "== SI_POLY_ABL=$m" | tee -a $O/abl.txt
  SI_HIP_LIB=$lib timeout 300 python tools/conv_bench.py --min-ms 40 --algo s2poly $SH 2>&1 | grep -E "k3s2" | tee -a $O/abl.txt
done
