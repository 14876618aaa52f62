#!/bin/bash
# tools/slab_pmc.sh (GPU box): SQ / LDS counters of the slab kernel on the two stride-1 target layers (two rocprofv3 --pmc passes)
SH="--f16 --shape 32,40,40,128,128,3,1,1 --shape 32,20,20,256,256,3,1,1 --reps 50"
bash tools/run_pmc.sh r05_slab_pmc_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" python3 $PWD/tools/conv_bench.py $SH | grep slab
bash tools/run_pmc.sh r05_slab_pmc_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" python3 $PWD/tools/conv_bench.py $SH | grep slab
