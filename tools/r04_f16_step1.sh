#!/bin/bash
# round 4, step 1: fp16 "bd" kernels (weights straight from L2 in lane order) -- bit-identity tests, then the standalone sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r04_f16_bd
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f16.py -x -q > $O/pytest_f16.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_f16.txt
tail -5 $O/pytest_f16.txt
timeout 900 bash tools/f16_sweep.sh r04_f16_bd "0 3 4 5 6 7 8" > $O/sweep.txt 2>&1
cat $O/sweep.txt
