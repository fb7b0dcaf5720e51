#!/usr/bin/env bash
# Final measurement pass of round 6 (one GPU call): the default bench.py line (CPU baseline, precision_modes, box_calibration, clock / power), the
# rocprofv3 passes of bench.py (kernel trace + stats, FETCH_SIZE, WRITE_SIZE, MFMA busy), the per-step trace table, the exact-fp32 line with its
# roofline leg, the other configurations per GPU.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
python bench.py > $O/r6final_bench.json 2> $O/r6final_bench.err
PASS_TIMEOUT=240 bash benchmarks/run_profile.sh r6final 4 > $O/r6final_profile.log 2>&1
python benchmarks/trace_step.py "$(find $O/prof_r6final/trace -name "*kernel_trace.csv" | head -1)" 80 > $O/r6final_trace_step.txt 2>&1 || true
find $O/prof_r6final -name '*.csv' -size +8M -delete
python bench.py --precision f32 --steps 60 --no-cpu-baseline > $O/r6final_bench_f32.json 2>/dev/null
for c in 3 4 5; do
  timeout 300 python bench.py --config $c --steps 60 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1
done > $O/r6final_other_configs.jsonl
tail -1 $O/r6final_bench.json | cut -c1-3000; cat $O/r6final_other_configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['metric'][-20:], r['value'], r['ms_per_step'])"
head -45 $O/prof_r6final/summary.txt; head -12 $O/r6final_trace_step.txt
