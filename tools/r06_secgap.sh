#!/bin/bash
# tools/r06_secgap.sh (GPU box): the `secondary` leg's fp16 figure against a run of its own, same box -- warm-up / window length
cd "$(dirname "$0")/.."
O=gpurun_out/r06_secgap; mkdir -p $O
export TMPDIR=/tmp
p() { python3 -c "import json,sys; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); s=d.get('secondary') or {}; print('$2', 'value', d['value'], 'secondary fp16', (s.get('yolov5s_fp16_b32') or {}).get('value'), 'f32split', (s.get('yolov5s_f32split_b32') or {}).get('value'))"; }
python3 bench.py --fp16 1 --no-cpu-baseline --no-aux --no-secondary --min-time 3 > $O/own1.json 2>/dev/null; p $O/own1.json "own run (fp16)          "
python3 bench.py --no-cpu-baseline > $O/def1.json 2>/dev/null; p $O/def1.json "default secondary       "
python3 bench.py --no-cpu-baseline --secondary-warmup 1.5 --secondary-time 2 > $O/warm.json 2>/dev/null; p $O/warm.json "warmup 1.5 s, 2 s timed "
python3 bench.py --no-cpu-baseline --no-aux > $O/noaux.json 2>/dev/null; p $O/noaux.json "no aux leg              "
python3 bench.py --no-cpu-baseline --min-time 1 > $O/short.json 2>/dev/null; p $O/short.json "headline 1 s only       "
python3 bench.py --fp16 1 --no-cpu-baseline --no-aux --no-secondary --min-time 3 > $O/own2.json 2>/dev/null; p $O/own2.json "own run (fp16) again    "
