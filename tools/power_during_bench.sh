#!/bin/bash
# tools/power_during_bench.sh [bench args] -- samples rocm-smi (socket power, sclk) while bench.py runs a sustained window.
OUT=${GRAFT_REPO_ROOT:-.}/gpurun_out
TAG=${TAG:-bench}
rm -f /tmp/stop
( while [ ! -f /tmp/stop ]; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz)/sclk \1/' -e 's/.*Power (W): /W /' | tr '\n' ' '
    echo
    sleep 0.2
  done ) > $OUT/power_$TAG.txt &
SPID=$!
python bench.py --no-cpu-baseline --no-aux "$@" > $OUT/power_$TAG.json 2>/dev/null
touch /tmp/stop; wait $SPID 2>/dev/null
python - <<PY
import json
d=json.load(open("$OUT/power_$TAG.json"))
print("$TAG", d["value"], d["ms_per_step"], d["roofline"]["achieved"] if d.get("roofline") else None)
rows=[l.split() for l in open("$OUT/power_$TAG.txt") if l.strip()]
vals=[(int(r[1]), float(r[3])) for r in rows if len(r)>=4 and r[0]=="sclk"]
busy=[v for v in vals if v[1]>600]
print("samples", len(vals), "busy", len(busy))
if busy:
    print("busy sclk MHz: min %d median %d max %d; power W: min %.0f median %.0f max %.0f" % (
        min(v[0] for v in busy), sorted(v[0] for v in busy)[len(busy)//2], max(v[0] for v in busy),
        min(v[1] for v in busy), sorted(v[1] for v in busy)[len(busy)//2], max(v[1] for v in busy)))
PY
