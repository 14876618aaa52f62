#!/bin/bash
# tools/r05_c64_pair.sh (GPU box): the 64- and 32-channel bottleneck pairs in one launch (conv_pw_patch_f16.hip) and the Detect tile kernel's
# lean decode: bit-identity tests, standalone timings, the fp16 network with fuse_pw on / off (batch 32, 8, 4).
python -m pytest tests/test_gpu_f16.py -q -x -k "bottleneck_pair or detect_tile" 2>&1 | tail -3
python tools/pw_slab_bench.py --shape 32,80,80,64 --shape 32,160,160,32 --shape 32,40,40,128
python tools/detect_bench_f16.py
python -m pytest tests/test_gpu_engine.py -q -x -k "fp16 or bottleneck" 2>&1 | tail -3
for b in 32 8 4; do
  echo "== fp16 batch $b"
  for i in 1 2 3; do
    for o in 0 1; do
      python bench.py --no-cpu-baseline --no-aux --no-secondary --min-time 1.5 --fp16 1 --batch $b --engine-opt fuse_pw=$o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fuse_pw=$o', d['value'], d['ms_per_step'])"
    done
  done
done
