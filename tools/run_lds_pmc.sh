#!/bin/bash
# tools/run_lds_pmc.sh <tag> [bench.py args] -- LDS bank-conflict share of every kernel of a bench run (rocprofv3 --pmc, one short pass)
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/raw" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --min-time 0 --max-windows 1 --no-cpu-baseline --no-aux --profile-passes 1 "$@" > "$OUT/bench.json" 2> "$OUT/stderr.txt" || true
cd "$REPO"
find "$OUT/raw" -name "*counter_collection.csv" -exec cp {} "$OUT/counters.csv" \;
rm -rf "$OUT/raw"
python3 - "$OUT" <<'PY'
import csv, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(out + "/counters.csv")):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if (name, r["Dispatch_Id"]) not in seen:
        seen.add((name, r["Dispatch_Id"])); cnt[name] += 1
with open(out + "/lds.txt", "w") as f:
    for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
        act = c.get("SQ_LDS_IDX_ACTIVE", 0)
        line = "%-100s launches %4d  LDS active %10.3g  conflict share %.2f" % (name[:100], cnt[name], act / cnt[name], c.get("SQ_LDS_BANK_CONFLICT", 0) / act if act else 0)
        print(line); f.write(line + "\n")
PY
