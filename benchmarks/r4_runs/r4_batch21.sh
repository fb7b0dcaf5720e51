#!/usr/bin/env bash
# r4 GPU call 21: hgemm2 weight ring depth 8 / 12 / 16 IN PLACE (cold weight streams), step-level A/B (the warm micro-benchmark sweep said < 10 %)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b21
mkdir -p $O
for v in 8 12 16 8 12 16; do
  CGD_HGEMM_RING=$v python bench.py --steps 120 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench.json'));print('HGEMM_RING $v', r['value'],r['ms_per_step'],r['roofline']['other_mfma_kernel']['ms_per_step'])"
done
