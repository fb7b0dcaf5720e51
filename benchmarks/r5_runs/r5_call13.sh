#!/usr/bin/env bash
# r5 GPU call 13: non-temporal weight loads (CGD_NT bits: 1 kconv on single-tile maps, 2 kgemm at M <= 64, 4 gemv): parity + same-box A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c13
mkdir -p $O
CGD_NT=7 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_gemm or small_map_variant or unet_small" > $O/pytest_nt.log 2>&1
echo "pytest nt rc $?"; tail -4 $O/pytest_nt.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3; do
  run "default (no nt)   " "A=1"
  run "nt: kconv 8x8     " "CGD_NT=1"
  run "nt: kgemm M<=64   " "CGD_NT=2"
  run "nt: gemv          " "CGD_NT=4"
  run "nt: all three     " "CGD_NT=7"
done | tee $O/ab.txt
