#!/usr/bin/env bash
# Final GPU call of round 4: full suite + smoke, then the measurement pass (bench line with CPU baseline, rocprofv3 trace + PMC passes, exact-fp32
# line, other configurations)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
mkdir -p $O/r4val
python -m pytest tests -m gpu -q > $O/r4val/pytest_gpu.log 2>&1
tail -4 $O/r4val/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r4val/smoke.log 2>&1; echo "smoke rc $?"
python bench.py > $O/r4final_bench.json 2> $O/r4final_bench.err
PASS_TIMEOUT=240 bash benchmarks/run_profile.sh r4final 4 > $O/r4final_profile.log 2>&1
python benchmarks/trace_step.py "$(find $O/prof_r4final/trace -name "*kernel_trace.csv" | head -1)" 70 > $O/r4final_trace_step.txt 2>&1 || true
python bench.py --precision f32 --steps 60 --no-cpu-baseline > $O/r4final_bench_f32.json 2>/dev/null
for c in 3 4 5; do
  timeout 300 python bench.py --config $c --steps 60 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1
done > $O/r4final_other_configs.jsonl
python -c "
import json;r=json.load(open('$O/r4final_bench.json'));print(r['value'],r['ms_per_step'],r['roofline']['frac'],r['roofline'].get('launch_classes'),r['hbm']['frac'])"
