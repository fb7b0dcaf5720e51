#!/usr/bin/env bash
# Final measurement pass of the round (one GPU call): the four rocprofv3 passes of bench.py (kernel trace + FETCH_SIZE + WRITE_SIZE + MFMA
# busy), the per-step trace table, the bench line with the CPU baseline, the other configurations per GPU.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
PASS_TIMEOUT=240 bash benchmarks/run_profile.sh r3final 4 > $O/r3final_profile.log 2>&1
python benchmarks/trace_step.py "$(find $O/prof_r3final/trace -name "*kernel_trace.csv" | head -1)" 70 > $O/r3final_trace_step.txt 2>&1 || true
cp $O/prof_r3final/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null || true
python bench.py > $O/r3final_bench.json 2> $O/r3final_bench.err
for c in 3 4 5; do
  timeout 300 python bench.py --config $c --steps 60 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1
done > $O/r3final_other_configs.jsonl
tail -1 $O/r3final_bench.json | cut -c1-1200; cat $O/r3final_other_configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['metric'][-20:], r['value'], r['ms_per_step'])"
head -30 $O/prof_r3final/summary.txt
