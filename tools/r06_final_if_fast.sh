#!/bin/bash
# tools/r06_final_if_fast.sh (GPU box): the pool's boxes differ by up to 7 % under fp16 MFMA load (clock behaviour); the committed evidence set names its box
# class.  Probe the box with one short headline run and one fp16 run; run the full evidence script only on a box of the fast class, else report and leave.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P=$(python3 bench.py --no-cpu-baseline --no-aux --no-secondary --min-time 2 2>/dev/null | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])")
Q=$(python3 bench.py --no-cpu-baseline --no-aux --no-secondary --min-time 2 --fp16 1 2>/dev/null | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['value'])")
echo "probe: fp32 $P img/s, fp16 $Q img/s"
mkdir -p gpurun_out; echo "probe: fp32 $P fp16 $Q" >> gpurun_out/r06_probe.txt
if python3 -c "import sys; sys.exit(0 if float('$Q') >= 25850 else 1)"; then
  bash tools/r06_profile_all.sh
else
  echo "slow-class box: evidence run skipped"
fi
