#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_pol4; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-aux --no-secondary --min-time 3 --engine-opt f32_split=1"
for rep in 1 2 3; do for pol in 3 4; do
  python3 bench.py $B --engine-opt f32_split_policy=$pol > $O/p${pol}_$rep.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/p${pol}_$rep.json') if l.startswith('{')][-1]); print('f32_split_policy=$pol rep $rep', d['value'], d['ms_per_step'])"
done; done
for pol in 3 4; do python3 bench.py $B --batch 4 --engine-opt f32_split_policy=$pol > $O/b4_p${pol}.json 2>/dev/null; python3 -c "import json; d=json.loads([l for l in open('$O/b4_p${pol}.json') if l.startswith('{')][-1]); print('batch 4 f32_split_policy=$pol', d['value'])"; done
python3 bench.py $B --layers --engine-opt f32_split_policy=4 > $O/l4.json 2> $O/l4.txt; grep -E "^conv_17 |^conv_35 |^conv_8 |^conv_14 |^conv_39 " $O/l4.txt | cut -c1-150
python3 bench.py $B --layers --engine-opt f32_split_policy=3 > $O/l3.json 2> $O/l3.txt; grep -E "^conv_17 |^conv_35 |^conv_8 |^conv_14 |^conv_39 " $O/l3.txt | cut -c1-150
