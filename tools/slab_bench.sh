#!/bin/bash
# tools/slab_bench.sh [batch] -- the fp16 3x3 layers over 128-512 channels, standalone, sustained, same box, interleaved:
# generic tiles (SI_CONV_F16_SLAB=0) against the one-shot slab kernels of conv_slab_f16.hip (round 5).
B=${1:-32}
SH="--shape $B,40,40,128,128,3,1,1 --shape $B,20,20,256,256,3,1,1 --shape $B,80,80,128,256,3,2,1 --shape $B,40,40,256,512,3,2,1 --shape $B,80,80,128,128,3,2,1 --shape $B,40,40,256,256,3,2,1"
for r in 1 2; do
  for on in 0 1; do
    echo "== SI_CONV_F16_SLAB=$on (round $r, batch $B)"
    SI_CONV_F16_SLAB=$on python tools/conv_bench.py --f16 $SH --min-ms 300 --graph 50 2>&1 | grep -E "k3s"
  done
done
