#!/usr/bin/env bash
# Final validation + measurement of the round on one box: smoke(), the whole GPU suite, the four rocprofv3 passes, the bench line with
# the CPU baseline, the other configurations per GPU.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > $O/r3_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r3_smoke.log
timeout 1700 python -m pytest tests -q -m gpu --durations=12 > $O/r3_pytest_gpu.log 2>&1
PASS_TIMEOUT=240 bash benchmarks/run_profile.sh r3final 4 > $O/r3final_profile.log 2>&1
python bench.py > $O/r3final_bench.json 2> $O/r3final_bench.err
for c in 3 4 5; do
  timeout 300 python bench.py --config $c --steps 60 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1
done > $O/r3final_other_configs.jsonl
tail -3 $O/r3_smoke.log; tail -16 $O/r3_pytest_gpu.log | cut -c1-300; tail -1 $O/r3final_bench.json | cut -c1-300
python -c "
import sys, json
for l in open('$O/r3final_other_configs.jsonl'):
    r = json.loads(l); print(r['metric'][-20:], r['value'], r['ms_per_step'])"
