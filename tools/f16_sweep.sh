#!/bin/bash
# tools/f16_sweep.sh <outdir> "<variants>" [conv_bench args] -- per-shape times of the fp16 conv tile variants (YOLOv5s shapes by default)
O=gpurun_out/$1; V=$2; shift; shift
mkdir -p $O
for v in $V; do SI_CONV_F16_VARIANT=$v python tools/conv_bench.py --f16 --min-ms 30 "$@" > $O/v$v.txt 2>&1; done
python3 - $O $V <<'PY'
import sys, re
o = sys.argv[1]; vs = sys.argv[2:]
tab = {}
order = []
for v in vs:
    for ln in open("%s/v%s.txt" % (o, v)):
        m = re.match(r"(\S+->\S+ k\ds\d)\s+(\d+)\s+f16 v\S+\s+([\d.]+)\s+([\d.]+)", ln)
        if m:
            k = m.group(1)
            if k not in tab:
                tab[k] = {"cnt": int(m.group(2))}
                order.append(k)
            tab[k][v] = (float(m.group(3)) * 1e3, float(m.group(4)))
print("%-36s %3s " % ("shape", "cnt") + " ".join("%11s" % ("v" + v) for v in vs) + "  best")
tot = {v: 0.0 for v in vs}; best_tot = 0.0
for k in order:
    r = tab[k]
    cells = []
    bv, bt = None, 1e9
    for v in vs:
        if v in r:
            cells.append("%6.1f/%4.0f" % (r[v][0], r[v][1])); tot[v] += r[v][0] * r["cnt"]
            if r[v][0] < bt: bt, bv = r[v][0], v
        else:
            cells.append("%11s" % "-")
    best_tot += bt * r["cnt"]
    print("%-36s %3d " % (k, r["cnt"]) + " ".join(cells) + "  v" + str(bv))
print("total us/forward: " + " ".join("v%s %.0f" % (v, tot[v]) for v in vs) + "  best-per-shape %.0f" % best_tot)
PY
