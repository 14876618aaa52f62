#!/bin/bash
# tools/r06_detsplit.sh (GPU box): Detect levels under f32_split as 64-pixel runs of the output (detect_split_tile_kernel, from two tiles per CU on): tests,
# the network against the previous commit's library (batch 32, 8, 4)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -q -x -k "detect or yolo or f32_split" 2>&1 | tail -3
bash tools/ab_prev.sh "--engine-opt f32_split=1" 3
bash tools/ab_prev.sh "--engine-opt f32_split=1 --batch 8" 2
bash tools/ab_prev.sh "--engine-opt f32_split=1 --batch 4" 2
