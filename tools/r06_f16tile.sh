#!/bin/bash
# tools/r06_f16tile.sh (GPU box): the 32 x 128 tile of the fp16 lane-order-weights kernel (id 13, retired in round 4) re-measured against today's tiles on the
# small-map layers of YOLOv5s batch 32, standalone, graph-replayed
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
SH=""
for s in 32,40,40,256,256,1,1,0 32,40,40,256,128,1,1,0 32,20,20,512,512,1,1,0 32,20,20,512,256,1,1,0 32,20,20,256,256,1,1,0 32,20,20,1024,512,1,1,0 32,40,40,512,256,1,1,0 32,80,80,128,256,3,2,1 32,40,40,256,512,3,2,1 32,80,80,128,128,3,2,1 32,40,40,256,256,3,2,1; do SH="$SH --shape $s"; done
for v in -1 10 13 9; do echo "== f16_tile $v"; python3 tools/conv_bench.py --f16 $SH --graph 20 --min-ms 20 --f16-tile $v 2>&1 | grep -E "k[13]s|total"; done
