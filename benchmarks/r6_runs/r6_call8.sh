#!/usr/bin/env bash
# r6 GPU call 8: does the DEFAULT (null) stream cost anything per launch?  The step loop on a created stream against the default stream.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c8
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', 'host enqueue', r['config']['host_isolated_enqueue_ms_per_rank'])")"; }
for i in 1 2 3; do
  run "default stream  " "A=1"
  run "created stream  " "CGD_BENCH_STREAM=1"
done | tee $O/ab_stream.txt
