#!/bin/bash
# tools/r06_collect.sh -- copy what tools/r06_profile_all.sh left under gpurun_out/r06_final/ into profiles/ (the tracked evidence)
F=gpurun_out/r06_final
for n in default fp16 f32_split resnet18_b64 resnet18_b64_fp16 resnet18_b64_f32_split mobilenetv3_b64 mobilenetv3_b64_fp16 batch16 batch8 batch4 batch2 batch1 2ranks_shared_device 8ranks_shared_device y256_shared_device world1_gather_both; do cp $F/bench_$n.json profiles/r06_bench_$n.json; done
cp $F/prof/kernel_stats.csv profiles/r06_kernel_stats.csv; cp $F/prof/bench.json profiles/r06_bench_under_rocprof.json
cp $F/prof_fp16/kernel_stats.csv profiles/r06_fp16_kernel_stats.csv; cp $F/prof_resnet18/kernel_stats.csv profiles/r06_resnet18_kernel_stats.csv
cp $F/prof_f32_split/kernel_stats.csv profiles/r06_f32_split_kernel_stats.csv
cp $F/layers.txt profiles/r06_layers.txt; cp $F/layers_fp16.txt profiles/r06_layers_fp16.txt; cp $F/layers_resnet18.txt profiles/r06_layers_resnet18.txt; cp $F/layers_f32_split.txt profiles/r06_layers_f32_split.txt
for b in 8 4 1; do cp $F/layers_batch$b.txt profiles/r06_layers_batch$b.txt; done
for t in traffic traffic_fp16 traffic_resnet18 traffic_b8 traffic_b4 traffic_f32split; do cp $F/$t/traffic.json profiles/$t.json; done
cp $F/busy_fp32/mfma_busy.json profiles/mfma_busy.json; cp $F/busy_fp16/mfma_busy.json profiles/mfma_busy_fp16.json; cp $F/busy_f32split/mfma_busy.json profiles/mfma_busy_f32split.json
cp $F/bench_default.err profiles/r06_bench_default_time.txt; cp $F/gpu_suite.txt profiles/r06_gpu_suite.txt; cp $F/mixed_metric.txt profiles/r06_mixed_metric.txt
