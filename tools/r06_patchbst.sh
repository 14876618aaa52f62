cd "$(dirname "$0")/.."
timeout 900 python3 -m pytest tests -m gpu -q -x tests/test_gpu_f16.py 2>&1 | tail -2
timeout 600 python3 -m pytest tests -m gpu -q -x -k "fp16 or f16 or half" tests/test_gpu_engine.py 2>&1 | tail -2
bash tools/ab_prev.sh "--fp16 1" 3
bash tools/ab_prev.sh "--fp16 1 --batch 8" 2
