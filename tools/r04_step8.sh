#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r04_s2poly; mkdir -p $O
for t in 16 32; do SI_POLY_TILES=$t timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "s2poly" 2>&1 | tail -2; done
SH="--shape 32,80,80,128,256,3,2,1 --shape 32,40,40,256,512,3,2,1 --shape 32,160,160,64,128,3,2,1 --shape 32,80,80,128,128,3,2,1 --shape 32,40,40,256,256,3,2,1 --shape 32,320,320,32,64,3,2,1"
for t in 16 32 16; do
  echo "== SI_POLY_TILES=$t" | tee -a $O/tiles.txt
  SI_POLY_TILES=$t timeout 300 python tools/conv_bench.py --min-ms 40 --algo s2poly $SH 2>&1 | grep -E "k3s2|total" | tee -a $O/tiles.txt
done
