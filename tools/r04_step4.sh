#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r04_step4
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
for p in 0 1 0 1; do
  SI_CONV_F16_POLICY=$p timeout 300 python bench.py --fp16 1 --no-cpu-baseline --no-aux --min-time 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp16 policy=$p', d['value'], d['ms_per_step'])" | tee -a $O/fp16_ab.txt
done
