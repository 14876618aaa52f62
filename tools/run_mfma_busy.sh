#!/bin/bash
# tools/run_mfma_busy.sh <tag> [extra bench.py args, e.g. --fp16 1 | --engine-opt f32_split=1] -- the matrix pipe's BUSY fraction of
# every kernel of a bench.py workload from the SQ counters (BASELINE.json's second metric: "conv MFMA util %").
# One rocprofv3 --pmc pass of its own (kernel-trace only, the program directly behind `--`), per dispatch:
#   mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32)   [MI355X_MICROARCH.md: MFMA_BUSY counts cycles per SIMD, summed over
#                    the chip; SQ_BUSY_CYCLES is per shader engine (32 of them x ... -- the normalisation rounds 2-3 used, 57-78 % then)]
#   mfma_busy_gui  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)      (the other normalisation of round 2)
#   valu_per_mfma  = SQ_INSTS_VALU / SQ_INSTS_MFMA  (SQ_INSTS_VALU includes the MFMAs)
#   lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
# Writes gpurun_out/<tag>/mfma_busy.json keyed like profiles/traffic*.json (kernel name as rocprofv3 prints it minus the namespace,
# `_workload`, `_recorded_at`); bench.py attaches the figure of the dominant kernel to `roofline.mfma_busy` when the workload matches.
set -e
TAG=${1:-mfma_busy}
shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
cd /tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/raw" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --min-time 0 --max-windows 1 --no-cpu-baseline --no-aux --no-secondary --profile-passes 1 "$@" > "$OUT/bench.json" 2> "$OUT/stderr.txt" || true
cd "$REPO"
find "$OUT/raw" -name "*counter_collection.csv" -exec cp {} "$OUT/counters.csv" \;
rm -rf "$OUT/raw"
python3 - "$OUT" <<'PY'
import csv, json, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(out + "/counters.csv")):
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if (name, r["Dispatch_Id"]) not in seen:
        seen.add((name, r["Dispatch_Id"])); cnt[name] += 1
final = {}
for k, c in agg.items():
    n = cnt[k]
    g = lambda x: c.get(x, 0.0) / n
    rec = {"launches": n, "SQ_VALU_MFMA_BUSY_CYCLES": round(g("SQ_VALU_MFMA_BUSY_CYCLES")), "SQ_BUSY_CYCLES": round(g("SQ_BUSY_CYCLES")),
           "GRBM_GUI_ACTIVE": round(g("GRBM_GUI_ACTIVE")), "SQ_INSTS_VALU": round(g("SQ_INSTS_VALU")), "SQ_INSTS_MFMA": round(g("SQ_INSTS_MFMA"))}
    if g("SQ_BUSY_CYCLES") > 0:
        rec["mfma_busy"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CYCLES") * 32.0), 4)
    if g("GRBM_GUI_ACTIVE") > 0:
        rec["mfma_busy_gui"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0), 4)
    if g("SQ_INSTS_MFMA") > 0:
        rec["valu_per_mfma"] = round(g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA"), 2)
    if g("SQ_LDS_IDX_ACTIVE") > 0:
        rec["lds_conflict"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
    rec["note"] = "rocprofv3 --pmc (own pass, kernel-trace only): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32)"
    final[k] = rec
final["_recorded_at"] = os.environ.get("SI_COMMIT", "unknown commit")
try:
    final["_workload"] = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])["config"]["workload_key"]
    eo = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])["config"].get("engine_options") or {}
    if eo:
        final["_workload"] += " " + ",".join("%s=%s" % kv for kv in sorted(eo.items()))
except Exception:
    final["_workload"] = "unknown"
json.dump(final, open(out + "/mfma_busy.json", "w"), indent=1)
for k, v in sorted(((k, v) for k, v in final.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"] * kv[1]["launches"])[:14]:
    print("%-90s n=%-4d busy %.3f (gui %.3f) valu/mfma %s lds_conflict %s" % (k[:90], v["launches"], v.get("mfma_busy", 0), v.get("mfma_busy_gui", 0), v.get("valu_per_mfma"), v.get("lds_conflict")))
PY
