#!/usr/bin/env python3
"""tools/gap_analysis.py <kernel_trace.csv> -- where a small-batch step's time goes: kernel durations vs the gaps between launches.

Reads a rocprofv3 --kernel-trace CSV, orders the dispatches by start time, takes the steady-state second half and prints
per step: sum of kernel durations, sum of gaps (next start - previous end, clipped at 0 when kernels of two streams overlap),
the gap distribution and, per kernel name, count / mean duration / mean gap BEFORE the launch.
"""
import csv
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[len(rows) // 2:]
    # a step starts at every launch of the stem kernel
    starts = [i for i, r in enumerate(rows) if "stem" in r[2]]
    if len(starts) < 3:
        print("no steps found")
        return
    rows = rows[starts[0]:starts[-1]]
    nsteps = len(starts) - 1
    span = rows[-1][1] - rows[0][0]
    dur = sum(e - s for s, e, _ in rows)
    gaps = []
    per = defaultdict(lambda: [0, 0, 0])
    busy_end = rows[0][0]
    idle = 0
    for i, (s, e, n) in enumerate(rows):
        g = s - busy_end if i else 0
        if g > 0:
            idle += g
        gaps.append(g)
        p = per[n]
        p[0] += 1
        p[1] += e - s
        p[2] += max(g, 0)
        busy_end = max(busy_end, e)
    print("steps %d  launches/step %.1f  span/step %.1f us  kernel time/step %.1f us  idle (no kernel running)/step %.1f us" %
          (nsteps, len(rows) / nsteps, span / nsteps / 1e3, dur / nsteps / 1e3, idle / nsteps / 1e3))
    pos = sorted(g for g in gaps if g > 0)
    if pos:
        print("gaps > 0: n/step %.1f  median %.2f us  p10 %.2f  p90 %.2f" %
              (len(pos) / nsteps, pos[len(pos) // 2] / 1e3, pos[len(pos) // 10] / 1e3, pos[len(pos) * 9 // 10] / 1e3))
    print("%-110s %6s %9s %9s" % ("kernel", "n/step", "dur us", "gap us"))
    for n, (c, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-110s %6.1f %9.2f %9.2f" % (n[:110], c / nsteps, d / c / 1e3, g / c / 1e3))


if __name__ == "__main__":
    main()
