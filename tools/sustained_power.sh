#!/bin/bash
# tools/sustained_power.sh -- is the fp32 matrix pipe clock- or power-limited under sustained load?  Runs (a) the bare MFMA
# loop and (b) the GEMM K-loop for a few seconds each while sampling rocm-smi (power, sclk) next to it.  GPU box only.
set -u
OUT=${GRAFT_REPO_ROOT:-.}/gpurun_out
mkdir -p $OUT
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mlb $GRAFT_REPO_ROOT/tools/mfma_loop_bench.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/glb $GRAFT_REPO_ROOT/tools/gemm_loop_bench.hip || exit 1
sample() {  # $1 = tag; samples until the file /tmp/stop exists
  rm -f /tmp/stop
  ( while [ ! -f /tmp/stop ]; do
      /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '
      echo
      sleep 0.25
    done ) > $OUT/power_$1.txt &
  SPID=$!
}
stop() { touch /tmp/stop; wait $SPID 2>/dev/null; }
/opt/rocm/bin/rocm-smi --showmaxpower --showpower --showclocks --showperflevel 2>&1 | head -40 > $OUT/power_idle.txt
echo "== bare MFMA chain, 8 workgroups / CU, ~3 s"; sample mfma;  timeout 60 /tmp/mlb 800000 0 8; stop
echo "== MFMA + LDS reads + writes + barriers (no global memory), ~3 s"; sample mfma_lds; timeout 60 /tmp/mlb 800000 2 8; stop
echo "== GEMM v0 (round-1 structure) M 51200 N 256 K 1152, ~3 s"; sample gemm_v0; timeout 60 /tmp/glb 51200 256 1152 10000 0; stop
echo "== GEMM v3 (128x64 LDS-DMA) same shape, ~3 s"; sample gemm_v3; timeout 60 /tmp/glb 51200 256 1152 10000 3; stop
echo "== GEMM v0 M 819200 N 64 K 288 (HBM heavy), ~3 s"; sample gemm_hbm; timeout 60 /tmp/glb 819200 64 288 8000 0; stop
for f in idle mfma mfma_lds gemm_v0 gemm_v3 gemm_hbm; do echo "--- $f"; tail -4 $OUT/power_$f.txt; done
