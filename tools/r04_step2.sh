#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r04_f16_abl
mkdir -p $O
SH="--shape 32,40,40,256,512,3,2,1 --shape 32,40,40,128,128,3,1,1 --shape 32,20,20,256,256,3,1,1 --shape 32,80,80,128,256,3,2,1 --shape 32,20,20,1024,512,1,1,0"
timeout 1200 bash tools/f16_ablate.sh run "0 1 2 3 7 8 19 23 27" 3 $SH > $O/abl_v3.txt 2>&1
cat $O/abl_v3.txt
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -k "host_tensor or borrowed or arena or lanes" > $O/pytest_engine.txt 2>&1; echo "rc=$?" >> $O/pytest_engine.txt
tail -5 $O/pytest_engine.txt
