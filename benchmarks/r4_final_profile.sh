#!/usr/bin/env bash
# Final measurement pass of round 4 (one GPU call): the rocprofv3 passes of bench.py (kernel trace + FETCH_SIZE + WRITE_SIZE + MFMA busy), the
# per-step trace table, the bench line with the CPU baseline, the exact-fp32 line, the other configurations per GPU, the config-1 drift report.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
python bench.py > $O/r4final_bench.json 2> $O/r4final_bench.err
PASS_TIMEOUT=240 bash benchmarks/run_profile.sh r4final 4 > $O/r4final_profile.log 2>&1
python benchmarks/trace_step.py "$(find $O/prof_r4final/trace -name "*kernel_trace.csv" | head -1)" 70 > $O/r4final_trace_step.txt 2>&1 || true
python bench.py --precision f32 --steps 60 --no-cpu-baseline > $O/r4final_bench_f32.json 2>/dev/null
for c in 3 4 5; do
  timeout 300 python bench.py --config $c --steps 60 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1
done > $O/r4final_other_configs.jsonl
timeout 600 python benchmarks/drift_report.py > $O/r4final_drift_config1.txt 2>&1 || true
tail -1 $O/r4final_bench.json | cut -c1-1500; cat $O/r4final_other_configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['metric'][-20:], r['value'], r['ms_per_step'])"
head -40 $O/prof_r4final/summary.txt; tail -12 $O/r4final_drift_config1.txt
