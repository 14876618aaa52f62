#!/bin/bash
# tools/ab_prev.sh (GPU box): same-box A/B of the in-tree kernel library against build_variants/prev/ (the previous commit's, built by hand: git stash,
# simpleinfer_amd.build.build_hip(out=build_variants/prev/libsi_hip.so), copy of libsimpleinfer_amd.so beside it, git stash pop).
# usage: bash tools/ab_prev.sh "<bench.py args>" [pairs]
cd "$(dirname "$0")/.."
O=gpurun_out/ab_prev; mkdir -p $O
export TMPDIR=/tmp
ARGS=${1:---fp16 1}
N=${2:-3}
B="--no-cpu-baseline --no-aux --no-secondary --min-time 3 $ARGS"
for rep in $(seq 1 $N); do for v in prev new; do
  if [ $v = prev ]; then export SI_HIP_LIB=$PWD/build_variants/prev/libsi_hip.so SI_HOST_LIB=$PWD/build_variants/prev/libsimpleinfer_amd.so; else unset SI_HIP_LIB SI_HOST_LIB; fi
  python3 bench.py $B > $O/${v}_$rep.json 2>$O/err_${v}_$rep.txt
  python3 -c "import json; d=json.loads([l for l in open('$O/${v}_$rep.json') if l.startswith('{')][-1]); print('$v rep $rep', d['value'], d['ms_per_step'])"
done; done
unset SI_HIP_LIB SI_HOST_LIB
