#!/usr/bin/env bash
# r5 GPU call 12: hgemm2 tile height on the large-M, small-K 1x1 convs (CGD_HGEMM_VAR=3: 64-row tiles everywhere)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c12
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2 3; do
  run "default                 " "A=1"
  run "64-row tiles everywhere " "CGD_HGEMM_VAR=3"
done | tee $O/ab.txt
