#!/usr/bin/env bash
# r6 GPU call 33: split-K of the weight GEMMs (hgemm2) again on the final kernels: chunks per slice >= 4 (default) / 6 / 12 (K = 768 GEMMs in one slice)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c33
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2 3; do
  run "default (>= 4 chunks per slice)" "A=1"
  run ">= 6 chunks per slice          " "CGD_HGEMM=1,64,6"
  run ">= 12 chunks per slice         " "CGD_HGEMM=1,64,12"
done | tee $O/ab_hgemm_split_final.txt
