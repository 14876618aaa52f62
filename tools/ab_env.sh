#!/bin/bash
# NOTE (round 6): the product libsi_hip.so reads NO environment variable any more.  The SI_CONV_* / SI_WINO_* / SI_DETECT_* switches this script
# flips exist only in the experiment build: `python -m simpleinfer_amd.build --experiment` -> build_variants/libsi_hip_exp.so, selected with
# SI_HIP_LIB=build_variants/libsi_hip_exp.so.  For A/Bs of the shipped library use engine options instead (tools/ab_options.sh:
# bench.py --engine-opt f16_slab=0 / f32_tile=4 / ...: SiConvPlan fields, include/si_hip.h).
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
