#!/bin/bash
# tools/r06_profile_all.sh -- the round-6 evidence set in one GPU-box call (everything lands under gpurun_out/r06_final/; what is judged is
# copied into profiles/ by tools/r06_collect.sh): the -m gpu suite with durations, PMC traffic tables and MFMA-busy tables (fp32, fp16, f32_split),
# the default bench line, rocprofv3 kernel stats of the same command (fp32, fp16, f32_split, ResNet18), per-layer tables, small batches, the
# shared-device multi-rank runs incl. the batch-256 configuration and RCCL through the C-ABI at world 1.
export SI_COMMIT=${SI_COMMIT:-$(cat .commit_id 2>/dev/null || git rev-parse --short HEAD 2>/dev/null || echo unknown)}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r06_final
mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -q --durations=25 ) > $O/gpu_suite.txt 2>&1
cp gpurun_out/mixed_metric.txt $O/ 2>/dev/null
bash tools/run_traffic.sh r06_final/traffic > $O/traffic_stdout.txt 2>&1
bash tools/run_traffic.sh r06_final/traffic_fp16 --fp16 1 > $O/traffic_fp16_stdout.txt 2>&1
bash tools/run_traffic.sh r06_final/traffic_resnet18 --model resnet18 --batch 64 --size 224 > $O/traffic_resnet18_stdout.txt 2>&1
bash tools/run_traffic.sh r06_final/traffic_b8 --batch 8 > $O/traffic_b8_stdout.txt 2>&1
bash tools/run_traffic.sh r06_final/traffic_b4 --batch 4 > $O/traffic_b4_stdout.txt 2>&1
bash tools/run_traffic.sh r06_final/traffic_f32split --engine-opt f32_split=1 > $O/traffic_f32split_stdout.txt 2>&1
bash tools/run_mfma_busy.sh r06_final/busy_fp32 > $O/busy_fp32.txt 2>&1
bash tools/run_mfma_busy.sh r06_final/busy_fp16 --fp16 1 > $O/busy_fp16.txt 2>&1
bash tools/run_mfma_busy.sh r06_final/busy_f32split --engine-opt f32_split=1 > $O/busy_f32split.txt 2>&1
for t in traffic traffic_fp16 traffic_resnet18 traffic_b8 traffic_b4 traffic_f32split; do cp $O/$t/traffic.json profiles/$t.json; done
cp $O/busy_fp32/mfma_busy.json profiles/mfma_busy.json; cp $O/busy_fp16/mfma_busy.json profiles/mfma_busy_fp16.json; cp $O/busy_f32split/mfma_busy.json profiles/mfma_busy_f32split.json
( time python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err
bash tools/run_rocprof.sh r06_final/prof --min-time 3 > /dev/null 2>&1
python3 bench.py --no-cpu-baseline --no-aux --no-secondary --layers > $O/bench_layers.json 2> $O/layers.txt
for b in 16 8 4 2 1; do python3 bench.py --batch $b --no-cpu-baseline --no-aux --min-time 3 --layers > $O/bench_batch$b.json 2> $O/layers_batch$b.txt; done
python3 bench.py --no-cpu-baseline --no-aux --fp16 1 --layers > $O/bench_fp16.json 2> $O/layers_fp16.txt
bash tools/run_rocprof.sh r06_final/prof_fp16 --fp16 1 --min-time 3 > /dev/null 2>&1
python3 bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 --layers > $O/bench_resnet18_b64.json 2> $O/layers_resnet18.txt
bash tools/run_rocprof.sh r06_final/prof_resnet18 --model resnet18 --batch 64 --size 224 --min-time 3 > /dev/null 2>&1
python3 bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 --fp16 1 > $O/bench_resnet18_b64_fp16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-aux --model mobilenetv3 --batch 64 --size 224 > $O/bench_mobilenetv3_b64.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-aux --model mobilenetv3 --batch 64 --size 224 --fp16 1 > $O/bench_mobilenetv3_b64_fp16.json 2>/dev/null
SI_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus 2 --gather p2p --no-cpu-baseline --no-aux --global-batch 32 --min-time 3 > $O/bench_2ranks_shared_device.json 2> $O/bench_2ranks.err
SI_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus 8 --gather p2p --no-cpu-baseline --no-aux --global-batch 32 --min-time 3 > $O/bench_8ranks_shared_device.json 2> $O/bench_8ranks.err
SI_BENCH_SHARE_DEVICE=1 timeout 900 python3 bench.py --gpus 8 --gather both --steps 3 --warmup 1 --min-time 0 --max-windows 1 --no-cpu-baseline --no-aux --no-secondary > $O/bench_y256_shared_device.json 2> $O/bench_y256.err
SI_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --gather both --no-cpu-baseline --no-aux --no-secondary --min-time 3 > $O/bench_world1_gather_both.json 2> $O/bench_world1_both.err
# opt-in f32_split (fp32 tensors, three fp16 MFMA products per fp32 product): bench line, per-layer table, kernel stats
python3 bench.py --no-cpu-baseline --no-aux --no-secondary --engine-opt f32_split=1 --layers > $O/bench_f32_split.json 2> $O/layers_f32_split.txt
bash tools/run_rocprof.sh r06_final/prof_f32_split --engine-opt f32_split=1 --min-time 3 > /dev/null 2>&1
python3 bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 --engine-opt f32_split=1 > $O/bench_resnet18_b64_f32_split.json 2>/dev/null
ls $O; tail -4 $O/gpu_suite.txt
