cd "$(dirname "$0")/.."
timeout 900 python3 -m pytest tests -m gpu -q -x -k "full_size_same_bits or full_size_against" --durations=5 2>&1 | tail -12
