#!/bin/bash
# tools/r06_wino43.sh (GPU box): the headline with the Winograd layers on F(4,3) (engine option winograd=2) against F(2,3) (default), interleaved
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="--no-cpu-baseline --no-aux --no-secondary --min-time 3"
for rep in 1 2; do for w in 1 2; do
  python3 bench.py $B --winograd $w 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('winograd=$w rep $rep', d['value'], d['ms_per_step'])"
done; done
python3 bench.py $B --winograd 2 --layers 2>&1 >/dev/null | grep -E "wino" | cut -c1-150 | head -12
