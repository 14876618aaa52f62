#!/bin/bash
# tools/run_pmc.sh <tag> <counters...> -- <program args>  : rocprofv3 --pmc pass (own run, kernel-trace only)
# usage: bash tools/run_pmc.sh tag "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" python3 tools/conv_bench.py --only 3x3 --reps 3
set -e
TAG=$1; shift
CTRS=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/raw" -- "$@" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt" || true
cd "$REPO"
find "$OUT/raw" -name "*counter_collection.csv" -exec cp {} "$OUT/counters.csv" \;
find "$OUT/raw" -name "*kernel_trace.csv" -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/raw"
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/counters.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    key = (r["Kernel_Name"][:90], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
seen = set()
for r in rows:
    key = (r["Kernel_Name"][:90], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    if (key, r["Dispatch_Id"]) not in seen:
        seen.add((key, r["Dispatch_Id"])); cnt[key] += 1
with open(out + "/summary.txt", "w") as f:
    for key, ctrs in agg.items():
        line = "%s grid=%s n=%d " % (key[0], key[1], cnt[key]) + " ".join("%s=%.4g" % (k, v / cnt[key]) for k, v in sorted(ctrs.items()))
        print(line); f.write(line + "\n")
PY
