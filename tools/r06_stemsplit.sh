#!/bin/bash
# tools/r06_stemsplit.sh (GPU box): f32_split with / without the RGB stem on the split kernel (engine option f32_split_policy = 4 / 3): tests first, then
# three interleaved pairs at batch 32, one at batch 4, and the stem's own time (profiles/r06_f32_split.txt)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_stemsplit; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests -m gpu -q -x -k "stem_split3 or split_stem or f32_split_stem" 2>&1 | tail -8
B="--no-cpu-baseline --no-aux --no-secondary --min-time 3 --engine-opt f32_split=1"
for rep in 1 2 3; do for pol in 3 4; do
  python3 bench.py $B --engine-opt f32_split_policy=$pol > $O/p${pol}_$rep.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/p${pol}_$rep.json') if l.startswith('{')][-1]); print('f32_split_policy=$pol rep $rep', d['value'], d['ms_per_step'])"
done; done
for pol in 3 4; do
  python3 bench.py $B --batch 4 --engine-opt f32_split_policy=$pol > $O/b4_p${pol}.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/b4_p${pol}.json') if l.startswith('{')][-1]); print('batch 4 f32_split_policy=$pol', d['value'], d['ms_per_step'])"
done
python3 bench.py $B --layers --engine-opt f32_split_policy=4 > $O/l4.json 2> $O/l4.txt; grep -E "^conv_0 " $O/l4.txt | cut -c1-150
python3 bench.py $B --layers --engine-opt f32_split_policy=3 > $O/l3.json 2> $O/l3.txt; grep -E "^conv_0 " $O/l3.txt | cut -c1-150
