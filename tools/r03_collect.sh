#!/bin/bash
# tools/r03_collect.sh -- copy what tools/r03_profile_all.sh left under gpurun_out/r03_final/ into profiles/ (the tracked evidence)
F=gpurun_out/r03_final
for n in default fp16 resnet18_b64 resnet18_b64_fp16 mobilenetv3_b64 batch16 batch8 batch4 batch2 batch1 2ranks_shared_device 8ranks_shared_device; do cp $F/bench_$n.json profiles/r03_bench_$n.json; done
cp $F/prof/kernel_stats.csv profiles/r03_kernel_stats.csv; cp $F/prof/bench.json profiles/r03_bench_under_rocprof.json
cp $F/prof_fp16/kernel_stats.csv profiles/r03_fp16_kernel_stats.csv; cp $F/prof_resnet18/kernel_stats.csv profiles/r03_resnet18_kernel_stats.csv
cp $F/layers.txt profiles/r03_layers.txt
for b in 8 4 1; do cp $F/layers_batch$b.txt profiles/r03_layers_batch$b.txt; done
cp $F/traffic/traffic.json profiles/traffic.json; cp $F/traffic_fp16/traffic.json profiles/traffic_fp16.json; cp $F/traffic_resnet18/traffic.json profiles/traffic_resnet18.json
