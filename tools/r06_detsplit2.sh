cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash tools/ab_prev.sh "--engine-opt f32_split=1" 3
bash tools/ab_prev.sh "--engine-opt f32_split=1 --batch 16" 2
bash tools/ab_prev.sh "--engine-opt f32_split=1 --batch 8" 2
