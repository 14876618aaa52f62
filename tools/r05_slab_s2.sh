#!/bin/bash
# tools/r05_slab_s2.sh (GPU box): the row-slab kernel at stride 2: bit-identity, the four YOLOv5s layers standalone (graph-timed, generic
# tiles vs slabs), the fp16 network with SI_CONV_F16_SLAB_S2=0 / 1.
timeout 600 python -m pytest tests/test_gpu_f16.py -q -x -k "slab_kernel" 2>&1 | tail -5
for v in 0 1; do
  echo "== SI_CONV_F16_SLAB_S2=$v"
  SI_CONV_F16_SLAB_S2=$v timeout 300 python tools/conv_bench.py --f16 --shape 32,80,80,128,256,3,2,1 --shape 32,40,40,256,512,3,2,1 --shape 32,80,80,128,128,3,2,1 --shape 32,40,40,256,256,3,2,1 --min-ms 300 --graph 50 2>&1 | grep k3s
done
bash tools/ab_env.sh SI_CONV_F16_SLAB_S2=0 SI_CONV_F16_SLAB_S2=1 "--fp16 1" 3
