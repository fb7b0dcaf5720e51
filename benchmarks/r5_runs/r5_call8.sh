#!/usr/bin/env bash
# r5 GPU call 8: kconv_kernel with its weight fragments two chunks ahead (three register sets): parity + same-box A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c8
mkdir -p $O
CGD_KCONV=1,1024,4,1,0,3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_map_variant or unet_small" > $O/pytest_ring3.log 2>&1
echo "pytest ring3 rc $?"; tail -4 $O/pytest_ring3.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2 3; do
  run "default (ring 2)" "A=1"
  run "ring 3          " "CGD_KCONV=1,1024,4,1,0,3"
done | tee $O/ab.txt
