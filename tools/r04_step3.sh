#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r04_f16_bd3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f16.py -x -q -k "tile_variant" > $O/pytest_f16.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_f16.txt
tail -5 $O/pytest_f16.txt
timeout 900 bash tools/f16_sweep.sh r04_f16_bd3 "0 9 10 11 12 13 14" > $O/sweep.txt 2>&1
cat $O/sweep.txt
