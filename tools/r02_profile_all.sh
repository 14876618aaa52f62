# provenance stamped into profiles/traffic*.json: the commit this tree is at (the GPU box has no .git: pass SI_COMMIT in)
export SI_COMMIT=${SI_COMMIT:-$(git rev-parse --short HEAD 2>/dev/null || echo unknown)}
bash tools/run_rocprof.sh r02c_prof > /dev/null 2>&1
bash tools/run_traffic.sh r02c_traffic > gpurun_out/r02c_traffic_stdout.txt 2>&1
python bench.py > gpurun_out/r02c_bench_default.json 2> gpurun_out/r02c_bench_default.err
python bench.py --no-cpu-baseline --no-aux --fp16 1 > gpurun_out/r02c_bench_fp16.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 > gpurun_out/r02c_bench_resnet18_b64.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 --fp16 1 > gpurun_out/r02c_bench_resnet18_b64_fp16.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aux --model mobilenetv3 --batch 64 --size 224 > gpurun_out/r02c_bench_mobilenetv3_b64.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aux --batch 1 --steps 200 --warmup 20 > gpurun_out/r02c_bench_batch1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aux --layers > /dev/null 2> gpurun_out/r02c_layers.txt
SI_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --gather p2p --no-cpu-baseline --no-aux --global-batch 32 > gpurun_out/r02c_bench_2ranks_shared_device.json 2> gpurun_out/r02c_bench_2ranks.err
C="python3 $GRAFT_REPO_ROOT/tools/conv_bench.py --reps 5 --shape 32,40,40,256,512,3,2,1 --shape 32,80,80,128,256,3,2,1 --shape 32,640,640,3,32,6,2,2"
bash tools/run_pmc.sh r02c_pmc_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" $C > /dev/null 2>&1
bash tools/run_pmc.sh r02c_pmc_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" $C > /dev/null 2>&1
head -12 gpurun_out/r02c_prof/kernel_stats.csv
cat gpurun_out/r02c_traffic_stdout.txt | tail -9
cat gpurun_out/r02c_bench_default.json | cut -c1-900
