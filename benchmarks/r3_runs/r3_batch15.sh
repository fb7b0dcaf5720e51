#!/usr/bin/env bash
# hgemm2_kernel's coalesced epilogue (CGD_HGEMM_EPI=1, default) against the per-lane one: bit-identity + parity tests, whole-step A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm or clip_vit_b32 or unet_64 or unet_128 or attention" 2>&1 | tail -4
run() {
  env "$@" timeout 100 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null |
    python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'])"
}
for _ in 1 2; do
  run CGD_HGEMM_EPI=0
  run CGD_HGEMM_EPI=1
done
