#!/bin/bash
# tools/variant_sweep.sh -- SI_CONV_VARIANT sweep with sustained timing (tools/conv_bench.py --min-ms): ms per launch for
# every tile variant on every distinct YOLOv5s conv shape at batch 32.  GPU box only.
for v in 4 0 1 2 5 6 7 8 10 3; do
  echo "== variant $v"
  SI_CONV_VARIANT=$v python tools/conv_bench.py --min-ms 150 --reps 20 2>&1 | grep -v "^in(" 
done
