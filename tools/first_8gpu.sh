#!/bin/bash
# tools/first_8gpu.sh [outdir] -- FIRST CONTACT with an 8-GPU MI355X node (committed in round 4, never run by the builder: a
# gpurun lease has one GPU).  Everything DESIGN.md section 7 predicts about the sharded path and could not be observed on one
# device, in one pass, ~6 minutes.  Run from the repo root on the node:  bash tools/first_8gpu.sh gpurun_out/first8
#
# What each block confirms or refutes (fields of bench.py's JSON line):
#   A  strong scaling of the headline metric -- batch 32 over 1 / 2 / 4 / 8 GPUs -- against the predicted 7.6 / 13.8 / 23.6 /
#      37.1 k img/s: `value`, `ms_per_step`, `step_bound` ("compute" expected at every N).
#   B  weak scaling, 32 images per GPU = "batch 256 over 8" (BASELINE.json configs[4]): predicted >= 0.95 of 8 x the 1-GPU rate;
#      `gather.gather_gbps_per_peer` is the per-link rate of a 274 MB slab (xGMI link peak ~153 GB/s; the prediction assumes
#      >= 100), `gather.gather_wait_ms` ~ 0 means the fan-out hid behind the next step, `step_bound`.
#   C  the two transports in ONE run (`--gather both`): direct IPC fan-out vs RCCL's ncclAllGather through the C-ABI
#      (`gather_ab.p2p` / `gather_ab.rccl`): is RCCL's schedule on the xGMI mesh a ring (~12.5 ms per 274 MB slab predicted) or
#      direct (~2-3 ms)?
#   D  copy engines: HSA_ENABLE_SDMA unset vs 0 -- do 7 concurrent hipMemcpyDtoDAsync per GPU run on the SDMA engines or as blit
#      kernels that take CUs from the convs?  (E: `__amd_rocclr_copyBuffer` in a kernel trace says blit.)
#   E  one rocprofv3 kernel trace of a single-GPU run with the program directly after `--` (never through env / bash -c: the
#      profiler initialises the GPU before the program starts and an exec after that takes the node down).
#   F  does the direct gather come up when every rank has its own HIP_VISIBLE_DEVICES (peers named by PCI bus id)?
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/first8}
mkdir -p "$O"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="--steps 20 --warmup 3 --min-time 4 --no-cpu-baseline --no-aux --no-secondary"
# Preflight (round 6): what the batch-256 configuration allocates PER RANK -- [B/G, 25200, 85] fp32 = 274.2 MB per slab, a gathered buffer of
# G = 8 slabs, 4 slots of those, all IPC-exported: 8 x 4 x 274.2 MB = 8.8 GB of shared buffers per rank beside the engine's ~3 GB arena.  Allocated
# and run once on ONE device shared by eight ranks in round 6 (profiles/r06_bench_y256_shared_device.json: handle exchange, 21 peer copies per
# rank, checksums of every slab on every rank -- 70 GB of the 288); on eight devices each holds one eighth of that.
python3 - <<'PY' | tee -a "$O/log.txt"
slab = 32 * 25200 * 85 * 4
print("preflight: slab %.1f MB, gathered buffer %.2f GB, 4 slots %.2f GB per rank" % (slab / 1e6, 8 * slab / 1e9, 4 * 8 * slab / 1e9))
PY
rocm-smi --showmeminfo vram 2>/dev/null | grep -E "Total Memory|Used" | tee -a "$O/log.txt"
run() { name=$1; shift; echo "== $name: $*" | tee -a "$O/log.txt"; timeout 600 "$@" > "$O/$name.json" 2>> "$O/log.txt"; echo "rc=$?" >> "$O/log.txt"; }

# A: strong scaling, batch 32 total
for n in 1 2 4 8; do run strong_$n python3 bench.py --gpus $n --global-batch 32 $B; done
# B: weak scaling, 32 per GPU
for n in 1 8; do run weak_$n python3 bench.py --gpus $n $B; done
# C: both transports, same run
run both_8 python3 bench.py --gpus 8 --gather both $B
run both_8_strong python3 bench.py --gpus 8 --global-batch 32 --gather both $B
# D: SDMA off
HSA_ENABLE_SDMA=0 run weak_8_sdma0 python3 bench.py --gpus 8 $B
# E: kernel trace, one GPU, program directly after --
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/prof1" -o r04 -- python3 "$OLDPWD/bench.py" --gpus 1 --steps 20 --warmup 3 --min-time 2 --no-cpu-baseline --no-aux --no-secondary > "$OLDPWD/$O/prof1.json" 2>> "$OLDPWD/$O/log.txt" )
# F: per-rank HIP_VISIBLE_DEVICES (launch.py gives every rank LOCAL_RANK; SI_LAUNCH_PIN_VISIBLE=1 makes it export
#    HIP_VISIBLE_DEVICES=<local rank> so that every rank calls its GPU "device 0")
SI_LAUNCH_PIN_VISIBLE=1 run weak_8_pinned python3 bench.py --gpus 8 --gather p2p $B

python3 - "$O" <<'PY'
import json, sys, glob, os
o = sys.argv[1]
base = None
print("%-18s %10s %9s %8s %-10s %12s %10s  %s" % ("run", "img/s", "ms/step", "eff", "gather", "GB/s/peer", "wait ms", "step_bound"))
for f in sorted(glob.glob(os.path.join(o, "*.json"))):
    try:
        lines = [l for l in open(f) if l.startswith("{")]
        d = json.loads(lines[-1])
    except Exception as ex:
        print("%-18s unreadable (%s)" % (os.path.basename(f), ex)); continue
    n = d["n_gpus"]
    if os.path.basename(f).startswith("strong_1"): base = d["value"]
    g = d.get("gather") or {}
    eff = d["value"] / (n * base) if base else float("nan")
    print("%-18s %10.0f %9.3f %8.3f %-10s %12s %10s  %s" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"], eff, d["config"].get("gather"),
          g.get("gather_gbps_per_peer"), g.get("gather_wait_ms"), d.get("step_bound")))
    ab = d.get("gather_ab")
    if ab:
        for k, v in ab.items():
            print("    %-6s %s" % (k, {kk: vv for kk, vv in v.items() if kk != "gather"}))
tr = glob.glob(os.path.join(o, "prof1", "**", "*kernel_stats.csv"), recursive=True)
for t in tr:
    blit = [l for l in open(t) if "copyBuffer" in l]
    print("kernel trace %s: %s" % (t, "blit copy kernels present: " + blit[0][:80] if blit else "no __amd_rocclr_copyBuffer rows"))
PY
