#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r04_s2poly
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "s2poly or resize" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -25 $O/pytest.txt
SH="--shape 32,80,80,128,256,3,2,1 --shape 32,40,40,256,512,3,2,1 --shape 32,160,160,64,128,3,2,1 --shape 32,80,80,128,128,3,2,1 --shape 32,40,40,256,256,3,2,1 --shape 32,320,320,32,64,3,2,1"
for algo in direct s2poly direct s2poly; do
  echo "== $algo" | tee -a $O/bench.txt
  timeout 300 python tools/conv_bench.py --min-ms 40 --algo $algo $SH 2>&1 | grep -E "k3s2|total" | tee -a $O/bench.txt
done
