#!/bin/bash
# tools/r06_split.sh (GPU box): f32_split -- sibling / wide-1x1 policy and launch-size tiles: tests, then same-box interleaved A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06_split
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python3 -m pytest tests -m gpu -q -x -k "split or guard" > $O/tests.txt 2>&1
tail -4 $O/tests.txt
B="--no-cpu-baseline --no-aux --no-secondary --min-time 3"
for rep in 1 2; do
  for pol in 1 2; do
    for b in 32 4; do
      timeout 300 python3 bench.py $B --batch $b --engine-opt f32_split=1 --engine-opt f32_split_policy=$pol > $O/split_p${pol}_b${b}_$rep.json 2>> $O/err.txt
      python3 -c "import json,sys; d=json.loads([l for l in open('$O/split_p${pol}_b${b}_$rep.json') if l.startswith('{')][-1]); print('policy $pol batch $b rep $rep', d['value'], d['ms_per_step'])"
    done
  done
done
for bm in 64 0; do
  timeout 300 python3 bench.py $B --batch 4 --engine-opt f32_split=1 --engine-opt split3_bm=$bm > $O/split_bm${bm}_b4.json 2>> $O/err.txt
  python3 -c "import json,sys; d=json.loads([l for l in open('$O/split_bm${bm}_b4.json') if l.startswith('{')][-1]); print('split3_bm $bm batch 4', d['value'], d['ms_per_step'])"
done
timeout 300 python3 bench.py $B --batch 4 > $O/fp32_b4.json 2>> $O/err.txt
python3 -c "import json,sys; d=json.loads([l for l in open('$O/fp32_b4.json') if l.startswith('{')][-1]); print('true fp32 batch 4', d['value'], d['ms_per_step'])"
timeout 300 python3 bench.py $B --engine-opt f32_split=1 --layers > $O/split_layers.json 2> $O/split_layers.txt
timeout 300 python3 bench.py $B --batch 4 --engine-opt f32_split=1 --layers > $O/split_layers_b4.json 2> $O/split_layers_b4.txt
