#!/bin/bash
# tools/slab_w2_ab.sh (GPU box): the slab kernels with one wave per SIMD (SI_CONV_F16_SLAB_W2=0) against two (512-thread workgroups, default):
# the two 3x3 layers, the fused bottleneck pair, the fp16 network; same box, interleaved.
for r in 1 2; do
  for w in 0 1; do
    echo "== SI_CONV_F16_SLAB_W2=$w (round $r)"
    SI_CONV_F16_SLAB_W2=$w python tools/conv_bench.py --f16 --shape 32,40,40,128,128,3,1,1 --shape 32,20,20,256,256,3,1,1 --min-ms 300 --graph 50 2>&1 | grep k3s
    SI_CONV_F16_SLAB_W2=$w python tools/pw_slab_bench.py --rounds 1
  done
done
bash tools/ab_env.sh SI_CONV_F16_SLAB_W2=0 SI_CONV_F16_SLAB_W2=1 "--fp16 1" 3
