#!/bin/bash
# ab_lib.sh <libA or ""> <libB or ""> "<bench args>" [rounds]
A="$1"; B="$2"; O="$3"; R=${4:-2}
for i in $(seq 1 $R); do
  for tag in A B; do
    if [ $tag = A ]; then L="$A"; else L="$B"; fi
    SI_HIP_LIB=$L python bench.py --no-cpu-baseline --no-aux --min-time 1.5 $O 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])"
  done
done
