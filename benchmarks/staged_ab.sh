#!/usr/bin/env bash
# First GPU call of the next round: grade the variants that were staged without a GPU (knob-gated, off by default) and A/B them on
# the step, all on ONE box.  Usage (on the GPU box, from the repo root): bash benchmarks/staged_ab.sh [steps]
#   CGD_ATTN_X3=1    fused attention kernels on bf16x3 MFMA products (attn.hip, X3 instantiations)
#   CGD_WINO_OCC2=1  wconv_kernel<GN, 2, 2>: the 8-row Winograd tile at two workgroups per CU
set -uo pipefail
STEPS=${1:-100}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
echo "== parity of the staged variants"
CGD_ATTN_X3=1 timeout 120 python -m pytest tests/test_gpu_parity.py -k "attention" -x -q 2>&1 | tail -3
CGD_WINO_OCC2=1 timeout 120 python -m pytest tests/test_gpu_parity.py -k "winograd or unet_256" -x -q 2>&1 | tail -3
echo "== same-box A/B (steps/s, ms/step)"
run() {
  env "$@" timeout 100 python bench.py --steps "$STEPS" --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null |
    python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'])"
}
for _ in 1 2; do
  run CGD_NOP=1
  run CGD_ATTN_X3=1
  run CGD_WINO_OCC2=1
  run CGD_ATTN_X3=1 CGD_WINO_OCC2=1
done
