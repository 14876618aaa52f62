#!/bin/bash
# tools/ab_options.sh "<opts A>" "<opts B>" [rounds] -- same-box, interleaved A/B of two bench.py option sets (e.g. "" vs
# "--engine-opt arena=0"); prints img/s per round and the medians.
A="$1"; B="$2"; R=${3:-3}
for i in $(seq 1 $R); do
  for tag in A B; do
    if [ $tag = A ]; then O="$A"; else O="$B"; fi
    python bench.py --no-cpu-baseline --no-aux --min-time 1.5 $O 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])"
  done
done
