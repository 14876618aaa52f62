#!/bin/bash
# tools/r06_cv3.sh (GPU box): C3 tail fusion (pair + concat + cv3) -- tests, then same-box interleaved A/B of the fp16 network with fuse_pw = 1 / 2
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06_cv3
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python3 -m pytest tests -m gpu -q -x -k "c3_tail or bottleneck or stem_pair or fp16_graph or full_size_properties_yolov5s" > $O/tests.txt 2>&1
tail -6 $O/tests.txt
B="--fp16 1 --no-cpu-baseline --no-aux --no-secondary --min-time 3"
for rep in 1 2 3; do
  for fp in 1 2; do
    timeout 300 python3 bench.py $B --engine-opt fuse_pw=$fp > $O/fp16_pw${fp}_$rep.json 2>> $O/err.txt
    python3 -c "import json,sys; d=json.loads([l for l in open('$O/fp16_pw${fp}_$rep.json') if l.startswith('{')][-1]); print('fuse_pw=$fp rep $rep', d['value'], d['ms_per_step'])"
  done
done
timeout 300 python3 bench.py $B --layers > $O/fp16_layers.json 2> $O/fp16_layers.txt
grep -E "conv_12 |conv_14 |conv_42 |conv_44 |sum of" $O/fp16_layers.txt
for b in 8 4; do
  for fp in 1 2; do
    timeout 300 python3 bench.py $B --batch $b --engine-opt fuse_pw=$fp > $O/fp16_b${b}_pw${fp}.json 2>> $O/err.txt
    python3 -c "import json,sys; d=json.loads([l for l in open('$O/fp16_b${b}_pw${fp}.json') if l.startswith('{')][-1]); print('batch $b fuse_pw=$fp', d['value'], d['ms_per_step'])"
  done
done
