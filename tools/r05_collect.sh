#!/bin/bash
# tools/r05_collect.sh -- copy what tools/r05_profile_all.sh left under gpurun_out/r05_final/ into profiles/ (the tracked evidence)
F=gpurun_out/r05_final
for n in default fp16 fp16_slab0 f32_split resnet18_b64 resnet18_b64_fp16 mobilenetv3_b64 mobilenetv3_b64_fp16 batch16 batch8 batch4 batch2 batch1 2ranks_shared_device 8ranks_shared_device world1_gather_both; do cp $F/bench_$n.json profiles/r05_bench_$n.json; done
cp $F/prof/kernel_stats.csv profiles/r05_kernel_stats.csv; cp $F/prof/bench.json profiles/r05_bench_under_rocprof.json
cp $F/prof_fp16/kernel_stats.csv profiles/r05_fp16_kernel_stats.csv; cp $F/prof_resnet18/kernel_stats.csv profiles/r05_resnet18_kernel_stats.csv
cp $F/prof_f32_split/kernel_stats.csv profiles/r05_f32_split_kernel_stats.csv
cp $F/layers.txt profiles/r05_layers.txt; cp $F/layers_fp16.txt profiles/r05_layers_fp16.txt; cp $F/layers_resnet18.txt profiles/r05_layers_resnet18.txt; cp $F/layers_f32_split.txt profiles/r05_layers_f32_split.txt
for b in 8 4 1; do cp $F/layers_batch$b.txt profiles/r05_layers_batch$b.txt; done
for t in traffic traffic_fp16 traffic_resnet18 traffic_b8 traffic_b4; do cp $F/$t/traffic.json profiles/$t.json; done
cp $F/bench_default.err profiles/r05_bench_default_time.txt
