#!/usr/bin/env bash
# r4 GPU call 6: A/B of the conv-epilogue GroupNorm statistics with the batched merge kernel + a kernel trace of the step
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b6
mkdir -p $O
for v in 0 1 0 1; do
  CGD_GN_EPI=$v python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/bench_epi${v}.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench_epi${v}.json'));print('GN_EPI $v', r['value'],r['ms_per_step'],r['config']['launches_per_step'], r['hbm']['ms_per_step'], r['hbm']['frac'], r['roofline']['frac'], r['roofline']['avg_launch_us'])"
done
(cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/trace" -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > "$ROOT/$O/trace.log" 2>&1)
python benchmarks/trace_step.py "$(find $O/trace -name "*kernel_trace.csv" | head -1)" 80 > $O/trace_step.txt 2>&1
find $O -name '*kernel_trace.csv' -size +20M -delete
head -60 $O/trace_step.txt
