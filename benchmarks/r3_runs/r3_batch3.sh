#!/usr/bin/env bash
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out
python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/r3b3_bench.json 2> $O/r3b3_bench.err
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > $O/r3_pytest_gpu.log 2>&1
tail -25 $O/r3_pytest_gpu.log | cut -c1-400
python -c "
import json
r = json.loads([l for l in open('$O/r3b3_bench.json') if l.startswith('{')][-1]); c = r['config']
print(r['value'], r['ms_per_step'], {k: c[k] for k in c if k.startswith('host')})
print(json.dumps(r.get('roofline'))[:1500]); print(json.dumps(r.get('hbm')))"
