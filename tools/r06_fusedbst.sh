#!/bin/bash
# tools/r06_fusedbst.sh (GPU box): the fp16 stem pair / triple kernel with countable buffer stores (BST) against the pointer stores under `if` it had
# (experiment build, SI_FUSED_BST=0), kernel level; the same-bits tests; then the fp16 network line three times
cd "$(dirname "$0")/.."
O=gpurun_out/r06_fusedbst; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests -m gpu -q -x -k "stem_s2c32 or stem_pair or fp16_storage" 2>&1 | tail -2
python3 tools/stem_fused_bench.py 2>&1 | tail -4
export SI_FUSED_ONLY=1
for rep in 1 2; do for b in 0 1; do SI_HIP_LIB=$PWD/build_variants/libsi_hip_exp.so SI_FUSED_BST=$b python3 tools/stem_fused_bench.py | tail -1; done; done
unset SI_FUSED_ONLY
B="--no-cpu-baseline --no-aux --no-secondary --min-time 3 --fp16 1"
for rep in 1 2 3; do
  python3 bench.py $B > $O/fp16_$rep.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/fp16_$rep.json') if l.startswith('{')][-1]); print('fp16 rep $rep', d['value'], d['ms_per_step'])"
done
python3 bench.py $B --layers > $O/l.json 2> $O/l.txt; grep -E "^conv_2 |^conv_0 |^conv_1 " $O/l.txt | cut -c1-150
