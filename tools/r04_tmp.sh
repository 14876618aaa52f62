#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 1200 python -m pytest tests/test_gpu_f16.py tests/test_gpu_tiles.py tests/test_gpu_ops.py -x -q -k "conv or tile or residual or f16" 2>&1 | tail -3
ab() { for r in 1 2; do for lib in build_variants/libsi_hip_head.so simpleinfer_amd/libsi_hip.so; do
SI_HIP_LIB=$lib python bench.py "$@" --no-cpu-baseline --no-aux --no-secondary --min-time 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$* $lib', d['value'], d['ms_per_step'])"
done; done; }
ab --fp16 1
ab --fp16 1 --model resnet18 --batch 64 --size 224
ab --model resnet18 --batch 64 --size 224
ab --fp16 1 --model mobilenetv3 --batch 64 --size 224
ab
