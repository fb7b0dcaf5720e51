#!/usr/bin/env bash
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or clip_vit or unet_small or unet_256 or modified_resnet" 2>&1 | tail -5 | cut -c1-600
timeout 300 python -m pytest tests/test_gpu_step.py -q -x -k "headline_shape_single or p_sample_trajectory_bf16x3 or dual_clip" 2>&1 | tail -3 | cut -c1-600
run() {
  env "$@" timeout 100 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null |
    python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'])"
}
for _ in 1 2; do
  run CGD_GEMV=0
  run CGD_GEMV=1
done
