#!/usr/bin/env bash
# Final measurement pass of round 5 (one GPU call): the default bench.py line (CPU baseline, precision_modes, clock / power), the rocprofv3 passes of
# bench.py (kernel trace + stats, FETCH_SIZE, WRITE_SIZE, MFMA busy), the per-step trace table, the exact-fp32 line, the other configurations per GPU,
# the attention micro-benchmark.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
python bench.py > $O/r5final_bench.json 2> $O/r5final_bench.err
PASS_TIMEOUT=240 bash benchmarks/run_profile.sh r5final 4 > $O/r5final_profile.log 2>&1
python benchmarks/trace_step.py "$(find $O/prof_r5final/trace -name "*kernel_trace.csv" | head -1)" 80 > $O/r5final_trace_step.txt 2>&1 || true
find $O/prof_r5final -name '*.csv' -size +8M -delete
python bench.py --precision f32 --steps 60 --no-cpu-baseline > $O/r5final_bench_f32.json 2>/dev/null
for c in 3 4 5; do
  timeout 300 python bench.py --config $c --steps 60 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1
done > $O/r5final_other_configs.jsonl
timeout 200 python benchmarks/probe_attn.py 50 > $O/r5final_attention_microbench.txt 2>/dev/null
tail -1 $O/r5final_bench.json | cut -c1-2500; cat $O/r5final_other_configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['metric'][-20:], r['value'], r['ms_per_step'])"
head -45 $O/prof_r5final/summary.txt; head -12 $O/r5final_trace_step.txt
