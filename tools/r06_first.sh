#!/bin/bash
# tools/r06_first.sh (GPU box) -- round 6, first call: evidence that needs no new product code
#  1. MFMA-busy counters of the three workloads
#  2. the small-batch floor: batch 4 / 8 per-layer tables with the K loop compiled out of conv_igemm_f32_fast_kernel and with the kernel
#     returning at entry
#  3. the batch-256 configuration allocated once on what exists: 8 ranks x 32 images on the shared device, 274 MB slabs, 4 slots (item 6)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06_first
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export SI_COMMIT=$(cat .commit_id 2>/dev/null || echo unknown)
rocm-smi --showclocks 2>/dev/null | head -20 > $O/clocks.txt
# 1
timeout 400 bash tools/run_mfma_busy.sh r06_first/busy_fp32 > $O/busy_fp32.txt 2>&1
timeout 400 bash tools/run_mfma_busy.sh r06_first/busy_fp16 --fp16 1 > $O/busy_fp16.txt 2>&1
timeout 400 bash tools/run_mfma_busy.sh r06_first/busy_f32split --engine-opt f32_split=1 > $O/busy_f32split.txt 2>&1
# 2
B="--no-cpu-baseline --no-aux --no-secondary --min-time 2 --layers"
for b in 4 8; do
  for v in base nokloop empty; do
    lib=""; [ $v != base ] && lib=$PWD/build_variants/libsi_hip_$v.so
    SI_HIP_LIB=$lib timeout 300 python3 bench.py --batch $b $B > $O/floor_b${b}_$v.json 2> $O/floor_b${b}_$v.txt
    SI_HIP_LIB=$lib timeout 300 python3 bench.py --batch $b --graph 1 $B > $O/floor_b${b}_${v}_graph.json 2> $O/floor_b${b}_${v}_graph.txt
  done
done
# 3
SI_BENCH_SHARE_DEVICE=1 timeout 900 python3 bench.py --gpus 8 --gather both --steps 3 --warmup 1 --min-time 0 --max-windows 1 --no-cpu-baseline --no-aux --no-secondary > $O/bench_y256_shared_device.json 2> $O/bench_y256_shared_device.err
SI_BENCH_FORCE_DIST=1 timeout 300 python3 bench.py --gpus 1 --gather both --steps 5 --warmup 2 --min-time 1 --no-cpu-baseline --no-aux --no-secondary > $O/bench_world1_both_b32.json 2> $O/bench_world1_both_b32.err
rocm-smi --showmeminfo vram 2>/dev/null | head > $O/vram_after.txt
tail -3 $O/busy_fp32.txt; tail -c 600 $O/bench_y256_shared_device.json; tail -5 $O/bench_y256_shared_device.err
