#!/bin/bash
# tools/r06_soak.sh (GPU box): the -m gpu suite three times in a row (fresh processes), then smoke() and the default bench line: flakiness check before hand-over
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do timeout 900 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2; done
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
python3 bench.py 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], {k: v.get('value') for k, v in d['secondary'].items() if isinstance(v, dict)})"
