#!/usr/bin/env bash
# r5 GPU call 16: the default bench.py line + kernel trace on the final defaults (kgemm mode 2)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c16
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-600
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/trace.log 2>&1)
T=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python benchmarks/trace_step.py "$T" 80 > $O/trace_step.txt 2>&1; head -50 $O/trace_step.txt
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
find $O/trace -name '*.csv' -size +5M -delete
