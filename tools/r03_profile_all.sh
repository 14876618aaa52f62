#!/bin/bash
# tools/r03_profile_all.sh -- the round-3 evidence set, one GPU-box call: rocprofv3 kernel stats + PMC traffic of the default
# bench command, the bench records of every configuration, the small-batch / shared-device multi-rank runs.  Everything lands
# under gpurun_out/r03_final/; what should be judged is copied into profiles/ by hand afterwards.
# provenance stamped into traffic*.json: the GPU box has no .git, so pass SI_COMMIT in (tools/r02_profile_all.sh did the same)
export SI_COMMIT=${SI_COMMIT:-$(git rev-parse --short HEAD 2>/dev/null || echo unknown)}
O=gpurun_out/r03_final
mkdir -p $O
# the PMC traffic tables first: bench.py attaches them (by workload) to the records taken below
bash tools/run_traffic.sh r03_final/traffic > $O/traffic_stdout.txt 2>&1
bash tools/run_traffic.sh r03_final/traffic_fp16 --fp16 1 > $O/traffic_fp16_stdout.txt 2>&1
bash tools/run_traffic.sh r03_final/traffic_resnet18 --model resnet18 --batch 64 --size 224 > $O/traffic_resnet18_stdout.txt 2>&1
cp $O/traffic/traffic.json profiles/traffic.json; cp $O/traffic_fp16/traffic.json profiles/traffic_fp16.json; cp $O/traffic_resnet18/traffic.json profiles/traffic_resnet18.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/run_rocprof.sh r03_final/prof --min-time 3 > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-aux --layers > $O/bench_layers.json 2> $O/layers.txt
for b in 16 8 4 2 1; do python bench.py --batch $b --no-cpu-baseline --no-aux --min-time 3 --layers > $O/bench_batch$b.json 2> $O/layers_batch$b.txt; done
python bench.py --no-cpu-baseline --no-aux --fp16 1 > $O/bench_fp16.json 2>/dev/null
bash tools/run_rocprof.sh r03_final/prof_fp16 --fp16 1 --min-time 3 > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 > $O/bench_resnet18_b64.json 2>/dev/null
bash tools/run_rocprof.sh r03_final/prof_resnet18 --model resnet18 --batch 64 --size 224 --min-time 3 > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 --fp16 1 > $O/bench_resnet18_b64_fp16.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aux --model mobilenetv3 --batch 64 --size 224 > $O/bench_mobilenetv3_b64.json 2>/dev/null
SI_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --gather p2p --no-cpu-baseline --no-aux --global-batch 32 --min-time 3 > $O/bench_2ranks_shared_device.json 2> $O/bench_2ranks.err
SI_BENCH_SHARE_DEVICE=1 python bench.py --gpus 8 --gather p2p --no-cpu-baseline --no-aux --global-batch 32 --min-time 3 > $O/bench_8ranks_shared_device.json 2> $O/bench_8ranks.err
ls -la $O
