#!/bin/bash
# tools/run_traffic.sh <tag> [extra bench.py args, e.g. --fp16 1] -- HBM traffic of the bench kernels from the TCC PMC counters, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), kernel-trace only.
# Writes gpurun_out/<tag>/traffic.json: per kernel name, average bytes per launch, with the gfx950 correction
# (FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads -> doubled; WRITE_SIZE is exact), units KiB.
set -e
TAG=${1:-traffic}
shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/raw_$C" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --min-time 0 --max-windows 1 --no-cpu-baseline --no-aux --no-secondary --profile-passes 1 "$@" > "$OUT/bench_$C.json" 2> "$OUT/stderr_$C.txt" || true
  cd "$REPO"
  find "$OUT/raw_$C" -name "*counter_collection.csv" -exec cp {} "$OUT/counters_$C.csv" \;
  rm -rf "$OUT/raw_$C"
done
python3 - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    seen = set()
    for r in csv.DictReader(open("%s/counters_%s.csv" % (out, c))):
        if r["Counter_Name"] != c:
            continue
        name = r["Kernel_Name"]
        name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        name = name.split("(")[0]
        agg[name] += float(r["Counter_Value"])
        if (name, r["Dispatch_Id"]) not in seen:
            seen.add((name, r["Dispatch_Id"])); cnt[name] += 1
    for k in agg:
        res[k][c] = agg[k] / cnt[k]
        res[k]["launches"] = cnt[k]
final = {}
for k, v in res.items():
    f = v.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0   # KiB -> bytes, gfx950 wide-read correction
    w = v.get("WRITE_SIZE", 0.0) * 1024.0
    final[k] = {"hbm_bytes_per_launch": round(f + w), "fetch_bytes": round(f), "write_bytes": round(w), "launches": v.get("launches", 0),
                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950)"}
import os
final["_recorded_at"] = os.environ.get("SI_COMMIT", "unknown commit")
# the workload these per-launch figures belong to (bench.py attaches them only to a run of the same workload)
try:
    final["_workload"] = json.loads(open(out + "/bench_FETCH_SIZE.json").read().strip().splitlines()[-1])["config"]["workload_key"]
except Exception:
    final["_workload"] = "unknown"
json.dump(final, open(out + "/traffic.json", "w"), indent=1)
for k, v in sorted(((k, v) for k, v in final.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
    print(k, v["hbm_bytes_per_launch"] / 1e6, "MB/launch", v["launches"])
PY
