#!/bin/bash
# tools/r06_upsbd.sh (GPU box): the dual-source form in the 64 x 128 lane-order tile (YOLOv5s conv_34 under fp16): tests, the network against the previous
# commit's library, the layer's line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -q -x -k "upsampled or upsample or fp16_graph or full_size_properties_yolov5s" 2>&1 | tail -3
bash tools/ab_prev.sh "--fp16 1" 5
python3 bench.py --no-cpu-baseline --no-aux --no-secondary --min-time 2 --fp16 1 --layers 2>&1 >/dev/null | grep -E "^conv_34 |^conv_40 " | cut -c1-160
