#!/bin/bash
# tools/r05_split3_cov.sh (GPU box): split3 with the 2 x 2-wave (<= 64 output channels) and 32-channel K-tile forms: error and time per
# launch on the new shapes; then the fused bottleneck pair at small batch in fp16 (fuse_pw on / off, VERDICT r04 item 3's regime).
python tools/split3_check.py
for b in 4 8; do
  echo "== fp16 batch $b: A = fuse_pw=0, B = fuse_pw=1 (img/s, ms per step)"
  for i in 1 2 3; do
    for o in 0 1; do
      python bench.py --no-cpu-baseline --no-aux --no-secondary --min-time 1.5 --fp16 1 --batch $b --engine-opt fuse_pw=$o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fuse_pw=$o', d['value'], d['ms_per_step'])"
    done
  done
done
