#!/bin/bash
# tools/run_gaps.sh <tag> [bench.py args] -- rocprofv3 kernel trace of a short bench run, summarised by tools/gap_analysis.py
TAG=${1:-gaps}
shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-aux --min-time 1 "$@" > "$OUT/bench.json" 2> "$OUT/stderr.txt" || true
cd "$REPO"
T=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 tools/gap_analysis.py "$T" > "$OUT/gaps.txt" 2>&1
rm -rf "$OUT/raw"
cat "$OUT/gaps.txt"
