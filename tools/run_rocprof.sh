#!/bin/bash
# tools/run_rocprof.sh <tag> [extra bench.py args] -- kernel-trace + stats of the default bench command (default --steps / --warmup /
# --min-time; only the CPU baseline and the side measurements are skipped) (plus the extra
# arguments, e.g. --fp16 1), summary into gpurun_out/<tag>/
# (copy what should be judged into profiles/).  Run on the GPU box from the repo root.
set -e
TAG=${1:-prof}
shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-aux --no-secondary "$@" > "$OUT/bench.json" 2> "$OUT/stderr.txt" || true
cd "$REPO"
find "$OUT/raw" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/raw" -name "*kernel_trace.csv" -exec sh -c 'head -1 {} > '"$OUT"'/kernel_trace_head.csv' \;
rm -rf "$OUT/raw"
head -30 "$OUT/kernel_stats.csv"
