#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r04_step5
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -40 $O/pytest_gpu.txt
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_step5/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "bound-aware", d["roofline"].get("frac_bound_aware"))
print(json.dumps(d.get("secondary"), indent=1))
print(json.dumps(d["roofline"].get("largest_instantiation")))
PY
