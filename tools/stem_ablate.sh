#!/bin/bash
# tools/stem_ablate.sh -- compile-time ablations of conv_stem_roll.hip (build_variants/libsi_hip_stem<bits>.so, SI_STEM_ABLATE bits:
# 1 no output stores, 2 no activation, 4 no row fetch in the loop, 16 no start stagger) on the YOLOv5s stem at batch 32
echo "full:"; python tools/conv_bench.py --min-ms 200 --shape 32,640,640,3,32,6,2,2 --shape 64,224,224,3,64,7,2,3 --shape 64,224,224,3,16,3,2,1 --shape 1,640,640,3,32,6,2,2 2>&1 | grep "k[0-9]s2"
for f in build_variants/libsi_hip_stem*.so; do
  echo "$f:"; SI_HIP_LIB=$f python tools/conv_bench.py --min-ms 200 --shape 32,640,640,3,32,6,2,2 2>&1 | grep k6s2
done
