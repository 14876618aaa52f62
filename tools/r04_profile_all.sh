#!/bin/bash
# tools/r04_profile_all.sh -- the round-4 evidence set in one GPU-box call (everything lands under gpurun_out/r04_final/; what is
# judged is copied into profiles/ by tools/r04_collect.sh): PMC traffic tables (fp32 batch 32 / 8 / 4, fp16, ResNet18), the default
# bench line, rocprofv3 kernel stats of the same command (fp32, fp16, ResNet18), per-layer tables with the bound each layer sits
# under, SQ counters of the fp16 K-heavy 3x3 layers (one-stage 64x64 vs the lane-order-weights tiles), the shared-device
# multi-rank runs incl. RCCL through the C-ABI at world 1.
export SI_COMMIT=${SI_COMMIT:-$(git rev-parse --short HEAD 2>/dev/null || echo unknown)}
export TMPDIR=/tmp
O=gpurun_out/r04_final
mkdir -p $O
bash tools/run_traffic.sh r04_final/traffic > $O/traffic_stdout.txt 2>&1
bash tools/run_traffic.sh r04_final/traffic_fp16 --fp16 1 > $O/traffic_fp16_stdout.txt 2>&1
bash tools/run_traffic.sh r04_final/traffic_resnet18 --model resnet18 --batch 64 --size 224 > $O/traffic_resnet18_stdout.txt 2>&1
bash tools/run_traffic.sh r04_final/traffic_b8 --batch 8 > $O/traffic_b8_stdout.txt 2>&1
bash tools/run_traffic.sh r04_final/traffic_b4 --batch 4 > $O/traffic_b4_stdout.txt 2>&1
for t in traffic traffic_fp16 traffic_resnet18 traffic_b8 traffic_b4; do cp $O/$t/traffic.json profiles/$t.json; done
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
bash tools/run_rocprof.sh r04_final/prof --min-time 3 > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-aux --no-secondary --layers > $O/bench_layers.json 2> $O/layers.txt
for b in 16 8 4 2 1; do python bench.py --batch $b --no-cpu-baseline --no-aux --min-time 3 --layers > $O/bench_batch$b.json 2> $O/layers_batch$b.txt; done
python bench.py --no-cpu-baseline --no-aux --fp16 1 --layers > $O/bench_fp16.json 2> $O/layers_fp16.txt
SI_CONV_F16_POLICY=0 python bench.py --no-cpu-baseline --no-aux --fp16 1 --min-time 3 > $O/bench_fp16_policy0.json 2>/dev/null
bash tools/run_rocprof.sh r04_final/prof_fp16 --fp16 1 --min-time 3 > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 --layers > $O/bench_resnet18_b64.json 2> $O/layers_resnet18.txt
bash tools/run_rocprof.sh r04_final/prof_resnet18 --model resnet18 --batch 64 --size 224 --min-time 3 > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-aux --model resnet18 --batch 64 --size 224 --fp16 1 > $O/bench_resnet18_b64_fp16.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aux --model mobilenetv3 --batch 64 --size 224 > $O/bench_mobilenetv3_b64.json 2>/dev/null
python bench.py --no-cpu-baseline --no-aux --model mobilenetv3 --batch 64 --size 224 --fp16 1 > $O/bench_mobilenetv3_b64_fp16.json 2>/dev/null
SI_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --gather p2p --no-cpu-baseline --no-aux --global-batch 32 --min-time 3 > $O/bench_2ranks_shared_device.json 2> $O/bench_2ranks.err
SI_BENCH_SHARE_DEVICE=1 python bench.py --gpus 8 --gather p2p --no-cpu-baseline --no-aux --global-batch 32 --min-time 3 > $O/bench_8ranks_shared_device.json 2> $O/bench_8ranks.err
SI_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --gather both --no-cpu-baseline --no-aux --no-secondary --min-time 3 > $O/bench_world1_gather_both.json 2> $O/bench_world1_both.err
# fp16 SQ / LDS counters on the K = 2304 3x3 stride-2 layer: one-stage 64x64 (variant 0) vs lane-order weights 128x128 as 1x4 waves (9) and 64x128 (10)
for v in 0 9 10; do
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    ( cd /tmp && SI_CONV_F16_VARIANT=$v rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OLDPWD/$O/pmc_f16_v$v -- python3 $OLDPWD/tools/conv_bench.py --f16 --reps 20 --shape 32,40,40,256,512,3,2,1 > /dev/null 2>&1 )
  done
done
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
out = open(o + "/pmc_f16.txt", "w")
for v in (0, 9, 10):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob("%s/pmc_f16_v%d/**/*counter_collection.csv" % (o, v), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "conv_igemm_f16" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Counter_Name"], r["Dispatch_Id"]) not in seen:
                seen.add((r["Counter_Name"], r["Dispatch_Id"])); cnt[r["Counter_Name"]] += 1
    per = {k: agg[k] / max(cnt[k], 1) for k in agg}
    print("variant %d (fp16 3x3 s2 40x40x256 -> 20x20x512, K = 2304, batch 32), per dispatch:" % v, file=out)
    for k in sorted(per):
        print("   %-32s %14.0f" % (k, per[k]), file=out)
    if per.get("SQ_WAIT_INST_ANY"):
        print("   SQ_WAIT_INST_LDS / SQ_WAIT_INST_ANY = %.3f" % (per.get("SQ_WAIT_INST_LDS", 0) / per["SQ_WAIT_INST_ANY"]), file=out)
    if per.get("SQ_BUSY_CYCLES") and per.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("   matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES / ...)): raw ratio %.4f" % (per["SQ_VALU_MFMA_BUSY_CYCLES"] / per["SQ_BUSY_CYCLES"]), file=out)
    if per.get("SQ_LDS_IDX_ACTIVE"):
        print("   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f" % (per.get("SQ_LDS_BANK_CONFLICT", 0) / per["SQ_LDS_IDX_ACTIVE"]), file=out)
out.close()
print(open(o + "/pmc_f16.txt").read())
PY
ls $O
