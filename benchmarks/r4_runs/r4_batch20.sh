#!/usr/bin/env bash
# r4 GPU call 20: weight prefetch modes: 0 off, 2 events only (no toucher), 3 one event + four touchers per ViT layer, 1 one event + toucher per GEMM
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b20
mkdir -p $O
for v in 0 2 3 1 0 3; do
  CGD_PREFETCH=$v python bench.py --steps 120 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench.json'));print('PREFETCH $v', r['value'],r['ms_per_step'],r['config']['launches_per_step'],r['roofline']['other_mfma_kernel']['ms_per_step'])"
done
