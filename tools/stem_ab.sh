#!/bin/bash
# tools/stem_ab.sh -- same-box A/B of the two fp32 stem kernels (SI_STEM_ROLL=0: conv_smallc.hip, default: conv_stem_roll.hip),
# sustained timing, the three stems at their benchmark batch sizes.
for roll in 0 1; do
  echo "== SI_STEM_ROLL=$roll"
  SI_STEM_ROLL=$roll python tools/conv_bench.py --min-ms 200 --shape 32,640,640,3,32,6,2,2 --shape 64,224,224,3,64,7,2,3 --shape 64,224,224,3,16,3,2,1 --shape 1,640,640,3,32,6,2,2 2>&1 | grep -v "^in("
done
