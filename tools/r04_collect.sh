#!/bin/bash
# tools/r04_collect.sh -- copy what tools/r04_profile_all.sh left under gpurun_out/r04_final/ into profiles/ (the tracked evidence)
F=gpurun_out/r04_final
for n in default fp16 fp16_policy0 resnet18_b64 resnet18_b64_fp16 mobilenetv3_b64 mobilenetv3_b64_fp16 batch16 batch8 batch4 batch2 batch1 2ranks_shared_device 8ranks_shared_device world1_gather_both; do cp $F/bench_$n.json profiles/r04_bench_$n.json; done
cp $F/prof/kernel_stats.csv profiles/r04_kernel_stats.csv; cp $F/prof/bench.json profiles/r04_bench_under_rocprof.json
cp $F/prof_fp16/kernel_stats.csv profiles/r04_fp16_kernel_stats.csv; cp $F/prof_resnet18/kernel_stats.csv profiles/r04_resnet18_kernel_stats.csv
cp $F/layers.txt profiles/r04_layers.txt; cp $F/layers_fp16.txt profiles/r04_layers_fp16.txt; cp $F/layers_resnet18.txt profiles/r04_layers_resnet18.txt
for b in 8 4 1; do cp $F/layers_batch$b.txt profiles/r04_layers_batch$b.txt; done
cp $F/pmc_f16.txt profiles/r04_f16_pmc.txt
for t in traffic traffic_fp16 traffic_resnet18 traffic_b8 traffic_b4; do cp $F/$t/traffic.json profiles/$t.json; done
