#!/usr/bin/env bash
# r4 GPU call 15: step-level A/B of the automatic two-K-group hgemm2 selection (CGD_HGEMM_KG=0 auto vs 1 never)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b15
mkdir -p $O
for v in 1 0 1 0 1 0; do
  CGD_HGEMM_KG=$v python bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench.json'));print('HGEMM_KG $v', r['value'],r['ms_per_step'],r['roofline']['other_mfma_kernel']['ms_per_step'], r['roofline']['other_mfma_kernel']['achieved'])"
done
