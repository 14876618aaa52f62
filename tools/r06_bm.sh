#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_bm2; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests -m gpu -q -x -k "split" 2>&1 | tail -3
B="--no-cpu-baseline --no-aux --no-secondary --min-time 2 --engine-opt f32_split=1"
for rep in 1 2; do for b in 32 4; do for bm in 0 64 32; do
  python3 bench.py $B --batch $b --engine-opt split3_bm=$bm > $O/b${b}_bm${bm}_$rep.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/b${b}_bm${bm}_$rep.json') if l.startswith('{')][-1]); print('batch $b split3_bm=$bm rep $rep', d['value'])"
done; done; done
python3 bench.py $B --layers --engine-opt split3_bm=64 > $O/l64.json 2> $O/l64.txt
python3 bench.py $B --layers > $O/l0.json 2> $O/l0.txt
python3 bench.py $B --layers --engine-opt split3_bm=32 > $O/l32.json 2> $O/l32.txt
for l in conv_1 detect_0; do grep "^$l " $O/l64.txt $O/l0.txt $O/l32.txt | cut -c1-160; done
