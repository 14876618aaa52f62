#!/bin/bash
# tools/slab_ablate.sh "<tags>" [batch] (GPU box): the two stride-1 target layers through build_variants/libsi_hip_slab_<tag>.so
# ("prod" = the product library), standalone, sustained, same box
B=${2:-32}
for t in $1; do
  lib=build_variants/libsi_hip_slab_$t.so; [ $t = prod ] && lib=simpleinfer_amd/libsi_hip.so
  echo "== $t"
  SI_HIP_LIB=$lib python tools/conv_bench.py --f16 --shape $B,40,40,128,128,3,1,1 --shape $B,20,20,256,256,3,1,1 --min-ms 200 --graph 50 2>&1 | grep -E "k3s"
done
