#!/usr/bin/env bash
# r5 GPU call 22 (the round's last seconds of GPU budget): rocprofv3 --kernel-trace --stats of bench.py on the final defaults + the per-step trace table
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$ROOT/gpurun_out/r5c22
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/trace.log 2>&1
echo "rocprof rc $?"
cd $ROOT
python benchmarks/trace_step.py "$(find $O/trace -name '*kernel_trace.csv' | head -1)" 80 > $O/trace_step.txt 2>&1 || true
find $O -name '*kernel_trace.csv' -delete
head -30 $O/trace_step.txt
