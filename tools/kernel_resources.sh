#!/bin/bash
# tools/kernel_resources.sh <file.hip> [filter] -- registers / spills / LDS / occupancy of every kernel in one HIP source
# (hipcc -Rpass-analysis=kernel-resource-usage, device code only), one line per kernel
F=$1; PAT=${2:-.}
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I simpleinfer_amd/csrc/hip $SI_DEFS --cuda-device-only -c "$F" -o $T/x.co \
  -Rpass-analysis=kernel-resource-usage 2> $T/remarks.txt; [ -n "$SI_KEEP_REMARKS" ] && cp $T/remarks.txt $SI_KEEP_REMARKS
python3 - $T/remarks.txt "$PAT" <<'PY'
import re, subprocess, sys
cur = None
rows = []
for ln in open(sys.argv[1]):
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    try:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if re.search(sys.argv[2], name):
        print("%-70s vgpr %3s agpr %3s spill %s scratch %s lds %6s occ %s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"),
              r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
PY
rm -rf $T
