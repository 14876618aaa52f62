#!/bin/bash
# tools/ab_policy.sh "<bench args>" ROUNDS POLICY...  -- same-box interleaved A/B of implicit-GEMM tile policies IN THE NETWORK
# (SI_CONV_POLICY="oc32,bigG,bigP,midG,midP,smallG,small", conv_igemm.hip conv_variant).  GPU box only.
O="$1"; R=$2; shift 2
for i in $(seq 1 $R); do
  for P in "$@"; do
    SI_CONV_POLICY=$P python bench.py --no-cpu-baseline --no-aux --min-time 3 $O 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$P', d['value'], d['ms_per_step'])"
  done
done
