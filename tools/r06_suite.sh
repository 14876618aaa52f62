#!/bin/bash
# tools/r06_suite.sh (GPU box): the -m gpu suite with per-test durations, then the default bench line
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06_suite
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python3 -m pytest tests -m gpu -q --durations=60 ) > $O/gpu_suite.txt 2>&1
cp gpurun_out/mixed_metric.txt $O/ 2>/dev/null
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -80 $O/gpu_suite.txt
