#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for v in 0 1 0 1; do SI_CONV_F16_S2C32=$v python tools/conv_bench.py --f16 --min-ms 50 --shape 32,320,320,32,64,3,2,1 2>&1 | grep -E "k3s2"; done
for r in 1 2; do for p in 0 1; do
SI_CONV_F16_S2C32=$p python bench.py --fp16 1 --no-cpu-baseline --no-aux --no-secondary --min-time 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp16 s2c32=$p', d['value'], d['ms_per_step'])"
done; done
