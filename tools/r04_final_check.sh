#!/bin/bash
# tools/r04_final_check.sh -- what the driver runs at round end, in one GPU-box call: the GPU test suite, smoke(), the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r04_final_check
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_final_check/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"].get("frac_bound_aware"))
print({k:(v.get("value"),v.get("frac")) for k,v in d["secondary"].items()})
PY
