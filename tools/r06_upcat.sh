#!/bin/bash
# tools/r06_upcat.sh (GPU box): f32_split with / without the dual-source layers on the split kernel (engine option f32_split_policy = 3 / 2), tests first, then
# three interleaved pairs and the two layers' own times (profiles/r06_f32_split.txt)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_upcat; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests -m gpu -q -x -k "split or upsampled or guard or full_size_properties_yolov5s" 2>&1 | tail -3
B="--no-cpu-baseline --no-aux --no-secondary --min-time 3 --engine-opt f32_split=1"
for rep in 1 2 3; do for pol in 2 3; do
  python3 bench.py $B --engine-opt f32_split_policy=$pol > $O/p${pol}_$rep.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/p${pol}_$rep.json') if l.startswith('{')][-1]); print('f32_split_policy=$pol rep $rep', d['value'], d['ms_per_step'])"
done; done
python3 bench.py $B --layers > $O/l3.json 2> $O/l3.txt; grep -E "^conv_34 |^conv_40 " $O/l3.txt | cut -c1-150
python3 bench.py $B --layers --engine-opt f32_split_policy=2 > $O/l2.json 2> $O/l2.txt; grep -E "^conv_34 |^conv_40 " $O/l2.txt | cut -c1-150
