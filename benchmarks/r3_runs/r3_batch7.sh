#!/usr/bin/env bash
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -k "thin_input or unet_small or unet_64 or unet_256 or lpips or test_conv" 2>&1 | tail -6 | cut -c1-400
run() {
  env "$@" timeout 100 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null |
    python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'])"
}
for _ in 1 2; do
  run CGD_THIN=0
  run CGD_THIN=1
done
