#!/usr/bin/env bash
# r5 GPU call 2: ADVICE fixes + split-load unroll + op-level GroupNorm-record tests; CGD_DEFER 1 vs 2 same-box A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r5c2
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "groupnorm or knob_change or unet_small or batch2 or winograd or clip_vit_b32 or unet_256" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -8 $O/pytest.log
CGD_DEFER=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet_small or unet_64 or clip_vit_b32" > $O/pytest_defer2.log 2>&1
echo "pytest defer2 rc $?"; tail -4 $O/pytest_defer2.log
for i in 1 2; do
  for f in 1 2; do
    echo "defer=$f: $(CGD_DEFER=$f timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"
  done
done | tee $O/ab_defer.txt
