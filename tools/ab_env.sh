#!/bin/bash
# tools/ab_env.sh "<ENV=.. A>" "<ENV=.. B>" "<bench.py options>" [rounds] -- same-box, interleaved A/B of two ENVIRONMENTS (kernel switches
# such as SI_CONV_F16_SLAB=0) under the same bench.py options; prints img/s and ms per step per round.
A="$1"; B="$2"; O="$3"; R=${4:-3}
for i in $(seq 1 $R); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --no-cpu-baseline --no-aux --no-secondary --min-time 1.5 $O 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '$E', d['value'], d['ms_per_step'])"
  done
done
