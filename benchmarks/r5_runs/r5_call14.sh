#!/usr/bin/env bash
# r5 GPU call 14: with CGD_DEFER=2 the reduces of the deferred GEMMs are absorbed by their consumers: is kgemm still worth it for those?
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c14
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2 3; do
  run "kgemm everywhere (default)       " "A=1"
  run "kgemm off                        " "CGD_KGEMM=0"
  run "kgemm only where nothing absorbs " "CGD_KGEMM=2"
done | tee $O/ab.txt
