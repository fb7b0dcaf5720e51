#!/usr/bin/env bash
# r5 GPU call 11: record invalidation by 2-D overlap in every activation writer: the record tests, UNets, and the per-step merge count (must stay 48)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c11
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "groupnorm or knob_change or batch2 or unet_small or unet_256" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces', r['config']['groupnorm_record_merges_per_step'], 'record merges')" | tee $O/bench.txt
