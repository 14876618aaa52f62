#!/bin/bash
# tools/r06_triple.sh (GPU box): the stem + 3x3 s2 + 1x1 launch -- tests, then same-box interleaved A/B of the fp16 network with fuse_stem = 1 / 2
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06_triple
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python3 -m pytest tests -m gpu -q -x -k "stem or winograd43 or range_guard or without_an_fp16_kernel or split" > $O/tests.txt 2>&1
tail -5 $O/tests.txt
B="--fp16 1 --no-cpu-baseline --no-aux --no-secondary --min-time 3"
for rep in 1 2 3; do
  for fs in 1 2; do
    timeout 300 python3 bench.py $B --engine-opt fuse_stem=$fs > $O/fp16_fs${fs}_$rep.json 2>> $O/err.txt
    python3 -c "import json,sys; d=json.loads([l for l in open('$O/fp16_fs${fs}_$rep.json') if l.startswith('{')][-1]); print('fuse_stem=$fs rep $rep', d['value'], d['ms_per_step'])"
  done
done
timeout 300 python3 bench.py $B --layers > $O/fp16_layers.json 2> $O/fp16_layers.txt
grep -E "conv_2 |conv_1 |sum of" $O/fp16_layers.txt
